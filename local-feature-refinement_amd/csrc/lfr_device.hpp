// Device-side arithmetic of the solver path (gfx950).  Included by lfr_solve.hip only.
//   interpolate()      cost.cc:13-48   BiquadraticInterpolator::Evaluate
//   eval_edge()        cost.cc:78-90   residual + (Ceres) ScaledLoss/Corrector, solve.cc:111,120
//   ls_next_step()     Ceres line_search.cc / polynomial.cc (Armijo, cubic interpolation)
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>

namespace lfrdev {

// numerical contract: solve.cc:89,111,120,147-154 + Ceres Solver::Options defaults
constexpr double kBound = 1.0;
constexpr double kCauchyB = 0.25 * 0.25;
constexpr double kCauchyC = 1.0 / (0.25 * 0.25);
constexpr double kTukeyA2 = 0.0625 * 0.0625;
constexpr int kMaxIterations = 100;
constexpr int kMaxInvalid = 10;
constexpr double kFunctionTol = 1e-4, kGradientTol = 1e-8, kParameterTol = 1e-4;
constexpr double kInitialRadius = 1e4, kMaxRadius = 1e16, kMinRadius = 1e-32;
constexpr double kMinRelDecrease = 1e-3, kMinLmDiag = 1e-6, kMaxLmDiag = 1e32;
constexpr double kLsSufficientDecrease = 1e-4, kLsMaxContraction = 1e-3, kLsMinContraction = 0.6;
constexpr int kLsMaxIterations = 20;
constexpr double kLsMinStep = 1e-9;

__device__ __forceinline__ double clampb(double v) { return fmin(fmax(v, -kBound), kBound); }

// 1/x for finite x > 0: v_rcp_f64 + two Newton steps (error <= 1 ulp; no denormal/inf handling)
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
// sqrt(x) for finite x >= 0 in the normal range (here: rho' * sim in [0, ~1]): v_rsq_f64 + Newton/Goldschmidt
__device__ __forceinline__ double fast_sqrt(double x) {
    if (!(x > 1e-290)) return sqrt(x);                 // zero / tiny: take the exact path (rare: Tukey outliers give 0)
    const double r = __builtin_amdgcn_rsq(x);
    double g = x * r, h = 0.5 * r;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    return fma(fma(-g, g, x), h, g);                   // final residual correction
}
// A literal pinned into an SGPR pair at the point of use: without it the compiler hoists the ten
// series coefficients below out of the solve loop and parks them in 20 VGPRs for the whole kernel
// (plus a v_mov_b64 in front of every v_fmac).  SALU moves issue beside the other wave's VALU work.
__device__ __forceinline__ double sgpr_const(double c) {
    asm volatile("" : "+s"(c));
    return c;
}
// f32 -> f64 that can be neither hoisted nor CSE'd (see eval_edge) and needs no copy of its source
__device__ __forceinline__ double cvt_pinned(float x) {
    double d;
    asm volatile("v_cvt_f64_f32_e32 %0, %1" : "=v"(d) : "v"(x));
    return d;
}
// log(x) for finite x >= 1 (the Cauchy loss evaluates log(1 + s/b)): x = m * 2^e with m in
// [sqrt(1/2), sqrt(2)), log m = 2 atanh((m-1)/(m+1)) by its odd series (|z| <= 0.1716, 11 terms),
// e*ln2 added in two pieces.  ~35 instructions instead of ~90 of the generic libm log; error ~1 ulp.
__device__ __forceinline__ double log_ge1(double x) {
    int e = __builtin_amdgcn_frexp_exp(x);             // x = f * 2^e, f in [0.5, 1)
    double m = __builtin_amdgcn_frexp_mant(x);
    if (m < 0.70710678118654752440) { m *= 2.0; --e; }
    const double z = (m - 1.0) * fast_rcp(m + 1.0);
    const double w = z * z;
    double p = sgpr_const(1.0 / 23.0);
    p = fma(p, w, sgpr_const(1.0 / 21.0)); p = fma(p, w, sgpr_const(1.0 / 19.0)); p = fma(p, w, sgpr_const(1.0 / 17.0));
    p = fma(p, w, sgpr_const(1.0 / 15.0)); p = fma(p, w, sgpr_const(1.0 / 13.0)); p = fma(p, w, sgpr_const(1.0 / 11.0));
    p = fma(p, w, sgpr_const(1.0 / 9.0)); p = fma(p, w, sgpr_const(1.0 / 7.0)); p = fma(p, w, sgpr_const(1.0 / 5.0));
    p = fma(p, w, sgpr_const(1.0 / 3.0));
    const double lm = fma(2.0 * z * w, p, 2.0 * z);    // log(m)
    const double ed = (double)e;
    return fma(ed, 6.93147180369123816490e-01, fma(ed, 1.90821492927058770002e-10, lm));
}

struct EdgeOut {
    double cost;            // 0.5 * rho(s)
    double r0, r1;          // corrected residual
    double j00, j01, j10, j11;   // corrected d r / d x_src
    double sq;              // sqrt(rho'): d r / d x_dst = sq * I
};

// flow: 18 floats, index 2*(3*i+j)+k.  WANT_JAC=false skips the derivative sums.
// The biquadratic form f_k = sum_ij Lr_i Lc_j d_ijk (cost.cc:32-35) is evaluated separably,
// f_k = sum_i Lr_i (sum_j Lc_j d_ijk): same 54 FMAs for value + both partials but no table of 27
// weight products, so the live set stays ~45 VGPRs (the association differs from the reference's
// (Lr_i*Lc_j)*d by rounding only, ~1e-16 relative).
// WANT_JAC = false evaluates the cost only: no derivative sums, no rho', no corrected residual/jacobian
// (o.cost is the only valid output).
template <bool WANT_JAC>
__device__ __forceinline__ void eval_edge(const float (&flow)[18], float simf, int kind, int tukey_variant,
                                          double x1r, double x1c, double x2r, double x2c, EdgeOut &o) {
    const double row = fmax(fmin(x1r, 0.5), -0.5), col = fmax(fmin(x1c, 0.5), -0.5);
    const bool row_in = (row == x1r), col_in = (col == x1c);     // cost.cc:38,41
    const double lc[3] = {2. * col * (col - .5), (-4.) * (col - .5) * (col + .5), 2. * col * (col + .5)};
    const double dlc[3] = {2. * col + 2. * (col - .5), (-4.) * (col - .5) + (-4.) * (col + .5), 2. * col + 2. * (col + .5)};
    double f0 = 0., f1 = 0., dr0 = 0., dr1 = 0., dc0 = 0., dc1 = 0.;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        // The flows never change during a solve, so the compiler would hoist the f32->f64
        // conversions out of the iteration loop and keep 36 extra VGPRs alive per edge slot.
        // The conversions are volatile asm: they stay next to their use (and read the resident
        // float registers directly).
        const double a0 = cvt_pinned(flow[6 * i]), a1 = cvt_pinned(flow[6 * i + 1]), b0 = cvt_pinned(flow[6 * i + 2]),
                     b1 = cvt_pinned(flow[6 * i + 3]), c0 = cvt_pinned(flow[6 * i + 4]), c1 = cvt_pinned(flow[6 * i + 5]);
        const double t0 = lc[0] * a0 + lc[1] * b0 + lc[2] * c0;
        const double t1 = lc[0] * a1 + lc[1] * b1 + lc[2] * c1;
        const double lri = (i == 0) ? 2. * row * (row - .5) : (i == 1) ? (-4.) * (row - .5) * (row + .5) : 2. * row * (row + .5);
        f0 += lri * t0; f1 += lri * t1;
        if (WANT_JAC) {
            const double u0 = dlc[0] * a0 + dlc[1] * b0 + dlc[2] * c0;
            const double u1 = dlc[0] * a1 + dlc[1] * b1 + dlc[2] * c1;
            const double dlri = (i == 0) ? 2. * row + 2. * (row - .5) : (i == 1) ? (-4.) * (row - .5) + (-4.) * (row + .5)
                                                                                 : 2. * row + 2. * (row + .5);
            dr0 += dlri * t0; dr1 += dlri * t1;
            dc0 += lri * u0; dc1 += lri * u1;
        }
    }
    if (WANT_JAC) {
        if (!row_in) { dr0 = 0.; dr1 = 0.; }
        if (!col_in) { dc0 = 0.; dc1 = 0.; }
    }
    const double r0 = x2r - x1r - f0, r1 = x2c - x1c - f1;       // cost.cc:87
    const double s = r0 * r0 + r1 * r1;
    const double w = cvt_pinned(simf);
    double rho0, rho1;
    if (kind == 0) {                                              // CauchyLoss(0.25)
        const double sum = 1.0 + s * kCauchyC;
        rho0 = kCauchyB * log_ge1(sum);
        rho1 = WANT_JAC ? fmax(DBL_MIN, fast_rcp(sum)) : 1.0;
    } else {                                                      // TukeyLoss(0.0625)
        const double k0 = (tukey_variant == 1) ? kTukeyA2 / 6.0 : kTukeyA2 / 3.0;
        const double k1 = (tukey_variant == 1) ? 0.5 : 1.0;
        if (s <= kTukeyA2) {
            const double v = 1.0 - s / kTukeyA2, v2 = v * v;
            rho0 = k0 * (1.0 - v2 * v);
            rho1 = k1 * v2;
        } else { rho0 = k0; rho1 = 0.0; }
    }
    rho0 *= w;
    o.cost = 0.5 * rho0;
    if (!WANT_JAC) return;
    rho1 *= w;
    const double sq = fast_sqrt(rho1);  // Corrector, rho'' <= 0 branch
    o.r0 = r0 * sq; o.r1 = r1 * sq; o.sq = sq;
    if (WANT_JAC) {
        o.j00 = (-1.0 - dr0) * sq; o.j01 = (-dc0) * sq;
        o.j10 = (-dr1) * sq;       o.j11 = (-1.0 - dc1) * sq;
    }
}

// ------------------------------------------------------------------------------------------
// Armijo line search helpers (rare path: only when the full LM step fails sufficient decrease)
// ------------------------------------------------------------------------------------------
struct LsSample { double x, value, gradient; bool value_valid, gradient_valid; };

__device__ inline double ipow(double x, int n) {       // x^n, n >= 0 (Ceres uses pow())
    double v = 1.0;
    for (int i = 0; i < n; ++i) v *= x;
    return v;
}

__device__ inline double polyval(const double *p, int n, double x) {
    double v = 0.0;
    for (int i = 0; i < n; ++i) v = v * x + p[i];
    return v;
}

__device__ inline int poly_root_real_parts(const double *pin, int n, double *out) {
    while (n > 0 && pin[0] == 0.0) { ++pin; --n; }
    const int deg = n - 1;
    if (deg <= 0) return 0;
    if (deg == 1) { out[0] = -pin[1] / pin[0]; return 1; }
    if (deg == 2) {
        const double a = pin[0], b = pin[1], c = pin[2];
        const double D = b * b - 4 * a * c, sD = sqrt(fabs(D));
        if (D >= 0) {
            if (b >= 0) { out[0] = (-b - sD) / (2.0 * a); out[1] = (2.0 * c) / (-b - sD); }
            else        { out[0] = (2.0 * c) / (-b + sD); out[1] = (-b + sD) / (2.0 * a); }
        } else { out[0] = -b / (2.0 * a); out[1] = -b / (2.0 * a); }
        return 2;
    }
    double a[5], zr[4], zi[4];
    double R = 0.0;
    for (int i = 0; i <= deg; ++i) a[i] = pin[i] / pin[0];
    for (int i = 1; i <= deg; ++i) R = fmax(R, fabs(a[i]));
    R = 1.0 + R;
    // Durand-Kerner style start points R * (0.4 + 0.9i)^k
    const double sr0[4] = {1.0, 0.4, -0.65, -0.908}, si0[4] = {0.0, 0.9, 0.72, -0.297};
    for (int k = 0; k < deg; ++k) { zr[k] = R * sr0[k]; zi[k] = R * si0[k]; }
    for (int it = 0; it < 200; ++it) {          // Aberth-Ehrlich
        double maxw = 0.0;
        for (int k = 0; k < deg; ++k) {
            double pr = 1.0, pi = 0.0, dr = 0.0, di = 0.0;
            for (int i = 1; i <= deg; ++i) {
                const double ndr = dr * zr[k] - di * zi[k] + pr, ndi = dr * zi[k] + di * zr[k] + pi;
                const double npr = pr * zr[k] - pi * zi[k] + a[i], npi = pr * zi[k] + pi * zr[k];
                dr = ndr; di = ndi; pr = npr; pi = npi;
            }
            double den = dr * dr + di * di;
            if (den == 0.0) continue;
            const double wr = (pr * dr + pi * di) / den, wi = (pi * dr - pr * di) / den;
            double sr = 0.0, si = 0.0;
            for (int j = 0; j < deg; ++j) {
                if (j == k) continue;
                const double er = zr[k] - zr[j], ei = zi[k] - zi[j], d2 = er * er + ei * ei;
                if (d2 == 0.0) continue;
                sr += er / d2; si += -ei / d2;
            }
            const double qr = 1.0 - (wr * sr - wi * si), qi = -(wr * si + wi * sr);
            den = qr * qr + qi * qi;
            if (den == 0.0) continue;
            const double cr = (wr * qr + wi * qi) / den, ci = (wi * qr - wr * qi) / den;
            zr[k] -= cr; zi[k] -= ci;
            maxw = fmax(maxw, fabs(cr) + fabs(ci));
        }
        if (maxw < 1e-15 * R) break;
    }
    for (int k = 0; k < deg; ++k) out[k] = zr[k];
    return deg;
}

__device__ inline int solve_dense(double *A, double *b, int n) {
    for (int k = 0; k < n; ++k) {
        int piv = k;
        for (int i = k + 1; i < n; ++i) if (fabs(A[i * n + k]) > fabs(A[piv * n + k])) piv = i;
        if (A[piv * n + k] == 0.0) return -1;
        if (piv != k) {
            for (int j = 0; j < n; ++j) { const double t = A[k * n + j]; A[k * n + j] = A[piv * n + j]; A[piv * n + j] = t; }
            const double t = b[k]; b[k] = b[piv]; b[piv] = t;
        }
        for (int i = k + 1; i < n; ++i) {
            const double f = A[i * n + k] / A[k * n + k];
            for (int j = k; j < n; ++j) A[i * n + j] -= f * A[k * n + j];
            b[i] -= f * b[k];
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        double s = b[k];
        for (int j = k + 1; j < n; ++j) s -= A[k * n + j] * b[j];
        b[k] = s / A[k * n + k];
    }
    return 0;
}

// Ceres MinimizeInterpolatingPolynomial over [x_min, x_max]
__device__ inline double minimize_interpolating_polynomial(const LsSample *s, int ns, double x_min, double x_max) {
    int ncons = 0;
    for (int i = 0; i < ns; ++i) ncons += (int)s[i].value_valid + (int)s[i].gradient_valid;
    const int deg = ncons - 1;
    double lhs[36], poly[6];
    for (int i = 0; i < 36; ++i) lhs[i] = 0.0;
    int row = 0;
    for (int i = 0; i < ns; ++i) {
        if (s[i].value_valid) {
            for (int j = 0; j <= deg; ++j) lhs[row * ncons + j] = ipow(s[i].x, deg - j);
            poly[row++] = s[i].value;
        }
        if (s[i].gradient_valid) {
            for (int j = 0; j < deg; ++j) lhs[row * ncons + j] = (deg - j) * ipow(s[i].x, deg - j - 1);
            poly[row++] = s[i].gradient;
        }
    }
    double best_x = (x_min + x_max) / 2.0;
    if (solve_dense(lhs, poly, ncons) != 0) return best_x;
    double best_v = polyval(poly, ncons, best_x), v;
    v = polyval(poly, ncons, x_min); if (v < best_v) { best_v = v; best_x = x_min; }
    v = polyval(poly, ncons, x_max); if (v < best_v) { best_v = v; best_x = x_max; }
    if (ncons > 2) {
        double deriv[5], roots[4];
        for (int i = 0; i < deg; ++i) deriv[i] = poly[i] * (deg - i);
        const int nr = poly_root_real_parts(deriv, deg, roots);
        for (int i = 0; i < nr; ++i) {
            if (roots[i] < x_min || roots[i] > x_max) continue;
            v = polyval(poly, ncons, roots[i]);
            if (v < best_v) { best_v = v; best_x = roots[i]; }
        }
    }
    for (int i = 0; i < ns; ++i) {
        if (s[i].x < x_min || s[i].x > x_max) continue;
        v = polyval(poly, ncons, s[i].x);
        if (v < best_v) { best_v = v; best_x = s[i].x; }
    }
    return best_x;
}

// One contraction of ArmijoLineSearch::DoSearch: given the failed `current` sample, returns the
// next step size, or a negative value when the search gives up.  `n_iter` is incremented.
__device__ __noinline__ double ls_next_step(const LsSample &initial, const LsSample &previous, const LsSample &current,
                                      double dir_max, int &n_iter) {
    if (++n_iter >= kLsMaxIterations) return -1.0;
    const double lo = kLsMaxContraction * current.x, hi = kLsMinContraction * current.x;
    double step;
    if (!current.value_valid) step = fmin(fmax(current.x * 0.5, lo), hi);
    else {
        LsSample s[3]; int ns = 0;
        s[ns++] = initial; s[ns++] = current;
        if (previous.value_valid) s[ns++] = previous;
        step = minimize_interpolating_polynomial(s, ns, lo, hi);
    }
    if (step * dir_max < kLsMinStep) return -1.0;
    return step;
}

// ------------------------------------------------------------------------------------------
// wave-level helpers (wave64)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int k) {     // k wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
    return v;
}
// LDS traffic between the lanes of ONE wave: program order is execution order, the fences only
// stop the compiler from moving LDS accesses across the hand-off.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------
// sub-group helpers: a wave64 hosts 64/S independent groups of S lanes (S = 8, 16, 32).
// Reductions are butterflies over DPP row operations (VALU, no LDS); every lane of a group ends
// with the bitwise-identical result (the pairing is symmetric), so group-uniform decisions stay
// uniform.  Broadcasts use ds_swizzle in bit-mask mode (LDS crossbar, no LDS memory).
// All of these must run with every lane of the wave active.
// ------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int l = __double2loint(v), h = __double2hiint(v);       // old = src: no zero-init moves needed
    const int lo = __builtin_amdgcn_update_dpp(l, l, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(h, h, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int PATTERN>
__device__ __forceinline__ double swizzle_f64(double v) {
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), PATTERN);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), PATTERN);
    return __hiloint2double(hi, lo);
}
constexpr int kDppRowRor = 0x120;          // row_ror:n = 0x120 + n (rotate within a 16-lane row)
constexpr int kDppHalfMirror = 0x141;      // lane i <-> 7 - i within 8 lanes
constexpr int kDppQuadXor1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int kSwizzleXor16 = (0x10 << 10) | 0x1F;

// xor-32 exchange (the two halves of the wave): ds_bpermute through __shfl_xor
__device__ __forceinline__ double xor32_f64(double v) { return __shfl_xor(v, 32, 64); }

template <int S, typename Op>
__device__ __forceinline__ double group_reduce(double v, Op op) {
    static_assert(S == 8 || S == 16 || S == 32 || S == 64, "group size");
    if (S == 8) {
        v = op(v, dpp_f64<kDppHalfMirror>(v));
        v = op(v, dpp_f64<kDppQuadXor1>(v));
        v = op(v, dpp_f64<kDppQuadXor2>(v));
    } else {
        v = op(v, dpp_f64<kDppRowRor + 8>(v));
        v = op(v, dpp_f64<kDppRowRor + 4>(v));
        v = op(v, dpp_f64<kDppRowRor + 2>(v));
        v = op(v, dpp_f64<kDppRowRor + 1>(v));
        if (S >= 32) v = op(v, swizzle_f64<kSwizzleXor16>(v));
        if (S == 64) v = op(v, xor32_f64(v));
    }
    return v;
}
template <int S>
__device__ __forceinline__ double group_sum(double v) { return group_reduce<S>(v, [](double a, double b) { return a + b; }); }
template <int S>
__device__ __forceinline__ double group_max(double v) { return group_reduce<S>(v, [](double a, double b) { return fmax(a, b); }); }

// value of lane (K mod 32-lane window) selected by a bit-mask swizzle: new = (lane & AND) | K
template <int AND, int K>
__device__ __forceinline__ double swz_bcast(double v) { return swizzle_f64<(K << 5) | AND>(v); }
template <int AND>
__device__ __forceinline__ double swz_bcast_k(double v, int k) {     // k folds after full unrolling
    switch (k) {
        case 0: return swz_bcast<AND, 0>(v);
        case 1: return swz_bcast<AND, 1>(v);
        case 2: return swz_bcast<AND, 2>(v);
        case 3: return swz_bcast<AND, 3>(v);
        case 4: return swz_bcast<AND, 4>(v);
        case 5: return swz_bcast<AND, 5>(v);
        case 6: return swz_bcast<AND, 6>(v);
        case 7: return swz_bcast<AND, 7>(v);
        case 8: return swz_bcast<AND, 8>(v);
        case 9: return swz_bcast<AND, 9>(v);
        case 10: return swz_bcast<AND, 10>(v);
        case 11: return swz_bcast<AND, 11>(v);
        case 12: return swz_bcast<AND, 12>(v);
        case 13: return swz_bcast<AND, 13>(v);
        case 14: return swz_bcast<AND, 14>(v);
        case 15: return swz_bcast<AND, 15>(v);
        case 16: return swz_bcast<AND, 16>(v);
        case 17: return swz_bcast<AND, 17>(v);
        case 18: return swz_bcast<AND, 18>(v);
        case 19: return swz_bcast<AND, 19>(v);
        case 20: return swz_bcast<AND, 20>(v);
        case 21: return swz_bcast<AND, 21>(v);
        case 22: return swz_bcast<AND, 22>(v);
        case 23: return swz_bcast<AND, 23>(v);
        case 24: return swz_bcast<AND, 24>(v);
        case 25: return swz_bcast<AND, 25>(v);
        case 26: return swz_bcast<AND, 26>(v);
        case 27: return swz_bcast<AND, 27>(v);
        case 28: return swz_bcast<AND, 28>(v);
        case 29: return swz_bcast<AND, 29>(v);
        case 30: return swz_bcast<AND, 30>(v);
        case 31: return swz_bcast<AND, 31>(v);
        default: return v;
    }
}
// value held by sub-lane k of the own S-lane group (k wave-uniform / compile-time after unrolling)
template <int S>
__device__ __forceinline__ double group_bcast_k(double v, int k) {
    if (S == 64) return readlane_f64(v, k);
    constexpr int and_mask = (S == 8) ? 0x18 : (S == 16) ? 0x10 : 0x00;
    return swz_bcast_k<and_mask>(v, k);
}

}  // namespace lfrdev

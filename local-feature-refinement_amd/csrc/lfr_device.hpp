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
// 1/sqrt(x) for finite x in the normal range: v_rsq_f64 + two coupled Goldschmidt steps (g -> sqrt x, h -> 1 / (2 sqrt x))
__device__ __forceinline__ double fast_rsqrt(double x) {
    const double r = __builtin_amdgcn_rsq(x);
    double g = x * r, h = 0.5 * r;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    h = fma(h, e, h);
    return h + h;
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
// The series coefficients 1/23 ... 1/3: either literals pinned into SGPR pairs at their use (22 s_mov_b32 per call, plus the moves that
// restore what those SGPRs held), or - LogCoef::regs, for callers with VGPRs to spare - eleven values the caller keeps in registers
// across its whole solve loop (v_fma_f64 takes them as its addend: no instruction per use).
struct LogCoef {
    double c[11];          // 1/23, 1/21, ..., 1/3
    __device__ __forceinline__ void load() {
#pragma unroll
        for (int i = 0; i < 11; ++i) { c[i] = 1.0 / (23.0 - 2.0 * i); asm volatile("" : "+v"(c[i])); }     // opaque: not rematerialised per use
    }
};
__device__ __forceinline__ double log_ge1(double x, const LogCoef *lc = nullptr) {
    int e = __builtin_amdgcn_frexp_exp(x);             // x = f * 2^e, f in [0.5, 1)
    double m = __builtin_amdgcn_frexp_mant(x);
    if (m < 0.70710678118654752440) { m *= 2.0; --e; }
    const double z = (m - 1.0) * fast_rcp(m + 1.0);
    const double w = z * z;
    double p;
    if (lc) {
        // one statement: ten dependent v_fma_f64 (not v_mov + v_fmac each, and no pad state after every statement)
        asm("v_fma_f64 %0, %1, %2, %3\n\tv_fma_f64 %0, %0, %2, %4\n\tv_fma_f64 %0, %0, %2, %5\n\tv_fma_f64 %0, %0, %2, %6\n\t"
            "v_fma_f64 %0, %0, %2, %7\n\tv_fma_f64 %0, %0, %2, %8\n\tv_fma_f64 %0, %0, %2, %9\n\tv_fma_f64 %0, %0, %2, %10\n\t"
            "v_fma_f64 %0, %0, %2, %11\n\tv_fma_f64 %0, %0, %2, %12"
            : "=&v"(p) : "v"(lc->c[0]), "v"(w), "v"(lc->c[1]), "v"(lc->c[2]), "v"(lc->c[3]), "v"(lc->c[4]), "v"(lc->c[5]), "v"(lc->c[6]),
              "v"(lc->c[7]), "v"(lc->c[8]), "v"(lc->c[9]), "v"(lc->c[10]));
    } else {
        p = sgpr_const(1.0 / 23.0);
        p = fma(p, w, sgpr_const(1.0 / 21.0)); p = fma(p, w, sgpr_const(1.0 / 19.0)); p = fma(p, w, sgpr_const(1.0 / 17.0));
        p = fma(p, w, sgpr_const(1.0 / 15.0)); p = fma(p, w, sgpr_const(1.0 / 13.0)); p = fma(p, w, sgpr_const(1.0 / 11.0));
        p = fma(p, w, sgpr_const(1.0 / 9.0)); p = fma(p, w, sgpr_const(1.0 / 7.0)); p = fma(p, w, sgpr_const(1.0 / 5.0));
        p = fma(p, w, sgpr_const(1.0 / 3.0));
    }
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
// AT_ZERO (wave-uniform, run time): both end points are the origin - the first evaluation of every solve (solve.cc:609-612 starts
// from zero displacements).  The Lagrange basis at 0 is {0, 1, 0}, its derivative {-1, 0, 1}: the interpolant is the centre flow and its
// partials are central differences - 10 conversions and 4 subtractions instead of 18 conversions, the basis and 54 multiply-adds,
// with the same bits (the general form multiplies by those exact zeros and ones).
template <bool WANT_JAC>
__device__ __forceinline__ void eval_edge(const float (&flow)[18], float simf, int kind, int tukey_variant,
                                          double x1r, double x1c, double x2r, double x2c, EdgeOut &o, const bool at_zero = false,
                                          const LogCoef *lc = nullptr) {
    double f0 = 0., f1 = 0., dr0 = 0., dr1 = 0., dc0 = 0., dc1 = 0.;
    if (at_zero) {
        f0 = cvt_pinned(flow[8]); f1 = cvt_pinned(flow[9]);
        if (WANT_JAC) {
            dr0 = cvt_pinned(flow[14]) - cvt_pinned(flow[2]); dr1 = cvt_pinned(flow[15]) - cvt_pinned(flow[3]);
            dc0 = cvt_pinned(flow[10]) - cvt_pinned(flow[6]); dc1 = cvt_pinned(flow[11]) - cvt_pinned(flow[7]);
        }
    } else {
        const double row = fmax(fmin(x1r, 0.5), -0.5), col = fmax(fmin(x1c, 0.5), -0.5);
        const bool row_in = (row == x1r), col_in = (col == x1c);     // cost.cc:38,41
        const double lc[3] = {2. * col * (col - .5), (-4.) * (col - .5) * (col + .5), 2. * col * (col + .5)};
        const double dlc[3] = {2. * col + 2. * (col - .5), (-4.) * (col - .5) + (-4.) * (col + .5), 2. * col + 2. * (col + .5)};
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
    }
    const double r0 = x2r - x1r - f0, r1 = x2c - x1c - f1;       // cost.cc:87
    const double s = r0 * r0 + r1 * r1;
    const double w = cvt_pinned(simf);
    double rho0 = 0.0, sq = 0.0;
    // Cauchy edges (intra-track, kind 0) take a path of their own when the whole wave has no other kind: a wave-uniform branch
    // instead of the exec-mask bracket of a divergent if / else around every evaluation
    auto cauchy = [&]() {                                         // CauchyLoss(0.25)
        const double sum = 1.0 + s * kCauchyC;
        rho0 = kCauchyB * log_ge1(sum, lc);
        if (WANT_JAC) {
            // Corrector (rho'' <= 0 branch): sqrt(w rho') = sqrt(w / sum) = w rsqrt(w sum): one reciprocal square root instead of a
            // reciprocal and a square root.  Outside the positive normal range (w <= 0, a non-finite residual) the two-step form
            // decides (class mask 0x2ff: everything but +normal; one v_cmp_class and a skipped branch on the usual path).
            const double p = w * sum;
            sq = w * fast_rsqrt(p);
            if (__builtin_amdgcn_class(p, 0x2ff)) sq = sqrt(w * fmax(DBL_MIN, 1.0 / sum));
        }
    };
    auto tukey = [&]() {                                          // TukeyLoss(0.0625)
        const double k0 = (tukey_variant == 1) ? kTukeyA2 / 6.0 : kTukeyA2 / 3.0;
        const double k1 = (tukey_variant == 1) ? 0.5 : 1.0;
        double rho1;
        if (s <= kTukeyA2) {
            const double v = 1.0 - s / kTukeyA2, v2 = v * v;
            rho0 = k0 * (1.0 - v2 * v);
            rho1 = k1 * v2;
        } else { rho0 = k0; rho1 = 0.0; }
        sq = WANT_JAC ? fast_sqrt(rho1 * w) : 0.0;
    };
    if (__builtin_amdgcn_ballot_w64(kind != 0) == 0ull) cauchy();
    else if (kind == 0) cauchy();
    else tukey();
    rho0 *= w;
    o.cost = 0.5 * rho0;
    if (!WANT_JAC) return;
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

// Real roots of q (degree m = 2..4, coefficients highest first, q[0] != 0) inside [lo, hi]: the critical points of q cut
// the interval into monotone pieces, a sign change brackets one root, a safeguarded Newton iteration polishes it.  Writes
// exactly m ascending values inside [lo, hi]; a piece without a root contributes its left end (x_min or an inflection point of
// the interpolant: a harmless extra candidate for the minimum).
__device__ inline double horner(const double *q, int m, double x) {
    double v = q[0];
    for (int i = 1; i <= m; ++i) v = v * x + q[i];
    return v;
}
__device__ inline void real_roots_in(const double *q, int m, double lo, double hi, double *out) {
    double c[3][5];                                           // q and its derivatives down to the quadratic
    const int nlev = m - 2;
    for (int i = 0; i <= m; ++i) c[0][i] = q[i];
    for (int l = 1; l <= nlev; ++l)
        for (int i = 0; i <= m - l; ++i) c[l][i] = c[l - 1][i] * (m - l + 1 - i);
    double r[4], bp[4];
    {
        const double A = c[nlev][0], B = c[nlev][1], C = c[nlev][2], D = B * B - 4 * A * C;
        double r0 = lo, r1 = lo;
        if (D >= 0) {
            const double sD = sqrt(D), t = B >= 0 ? -B - sD : -B + sD;      // -(B + sign(B) sqrt(D)): no cancellation
            const double u = t / (2.0 * A), v = t != 0.0 ? (2.0 * C) / t : u;
            r0 = fmin(u, v); r1 = fmax(u, v);
            if (!(r0 == r0)) r0 = lo;
            if (!(r1 == r1)) r1 = lo;
        }
        r[0] = fmin(fmax(r0, lo), hi); r[1] = fmin(fmax(r1, lo), hi);
    }
    for (int l = nlev - 1; l >= 0; --l) {
        const int dl = m - l;                                 // degree of c[l]; c[l + 1] is its derivative
        for (int i = 0; i < dl - 1; ++i) bp[i] = r[i];
        for (int i = 0; i < dl; ++i) {
            double a = i == 0 ? lo : bp[i - 1], b = i == dl - 1 ? hi : bp[i];
            const double fa = horner(c[l], dl, a), fb = horner(c[l], dl, b);
            double x = a;
            if (fa != 0.0 && fb == 0.0) x = b;
            else if ((fa < 0.0 && fb > 0.0) || (fa > 0.0 && fb < 0.0)) {
                x = 0.5 * (a + b);
                for (int it = 0; it < 200; ++it) {
                    const double fx = horner(c[l], dl, x);
                    if (fx == 0.0) break;
                    if ((fx < 0.0) == (fa < 0.0)) a = x; else b = x;
                    double xn = x - fx / horner(c[l + 1], dl - 1, x);
                    if (!(xn > a && xn < b)) xn = 0.5 * (a + b);
                    if (!(xn > a && xn < b)) break;           // the bracket is down to neighbouring numbers
                    if (fabs(xn - x) <= 2.220446049250313e-16 * fabs(xn)) { x = xn; break; }
                    x = xn;
                }
            }
            r[i] = x;
        }
    }
    for (int i = 0; i < m; ++i) out[i] = r[i];
}

// Candidate abscissae for the minimum of the interpolant over [lo, hi]: Ceres takes the real parts of ALL roots of the
// derivative (FindPolynomialRoots: companion-matrix eigenvalues) and keeps those inside the interval.  Degree <= 2 in closed
// form exactly as Ceres (a complex pair contributes its real part); degree 3 and 4: the real roots inside the interval - only a
// real critical point can be the minimum (a polynomial without one between two candidates is monotone there).  The first version
// ran a simultaneous complex iteration (Aberth) from a circle of radius 1 + max |a_i / a_0|: up to 200 complex steps on
// scratch arrays (~4e5 cycles per call: one component that contracted its step 40 times set the duration of the whole config-5
// launch), and it lost the small roots when the leading coefficient is tiny (a nearly cubic function fitted by a quartic).
// Loop bounds are run-time values on purpose: the arrays live in scratch and the function keeps a small register footprint, which
// is what the packed kernel needs around its (rare) call - a fully unrolled register version cost it 42 spilled VGPRs and 14 %.
__device__ inline int poly_root_real_parts(const double *pin, int n, double lo, double hi, double *out) {
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
    real_roots_in(pin, deg, lo, hi, out);
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
        const int nr = poly_root_real_parts(deriv, deg, x_min, x_max, roots);
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

// ---- the same contraction in REGISTERS, for the workgroup-per-component kernel ----
// Every loop below has compile-time bounds (the number of constraints N = 3..6 is a template parameter) and every array index is
// static after unrolling.  The loop version above keeps its small arrays in scratch memory and costs ~4e5 cycles per call
// whatever the root finder (measured: a component that contracted its step 40 times lived 10-12 ms and set the duration of the
// whole config-5 launch; 6.4 ms with this version).  The packed kernel keeps the loop version: this one's register footprint
// cost it 42 spilled VGPRs around the call and 14 % of its run time.  Both are pinned against numpy.roots (test_gpu_units.py).
template <int N>
__device__ __forceinline__ double polyval_n(const double (&p)[N], double x) {
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) v = v * x + p[i];
    return v;
}

// Divisions of the register / wave versions below.  -DLFR_LS_FAST_DIV=1 turns them into a multiplication by v_rcp_f64 + two Newton steps
// inside the normal range (a contraction performs ~45 divisions in dependent chains) - measured SLOWER (round 5: 12.9-21 us per call
// against 9.8-17.8): the safeguarded Newton iterations of the root isolation stop when two iterates agree to the last bit, and with a
// reciprocal that is off by an ulp they keep hopping between neighbours until the bracket closes.  IEEE divisions stay.
#ifndef LFR_LS_FAST_DIV
#define LFR_LS_FAST_DIV 0
#endif
__device__ __forceinline__ double ls_div(double a, double b) {
#if LFR_LS_FAST_DIV
    const double ab = fabs(b);
    if (ab > 1e-280 && ab < 1e280) return a * fast_rcp(b);
#endif
    return a / b;
}
// real roots of a cubic / quartic inside [lo, hi] (see real_roots_in above): every level returns exactly M ascending values
template <int M>
__device__ __forceinline__ double horner(const double (&q)[M + 1], double x) {
    double v = q[0];
#pragma unroll
    for (int i = 1; i <= M; ++i) v = v * x + q[i];
    return v;
}
template <int M>
__device__ __forceinline__ double bracketed_root(const double (&q)[M + 1], const double (&dq)[M], double a, double b, double fa) {
    double x = 0.5 * (a + b);
    for (int it = 0; it < 200; ++it) {
        const double fx = horner<M>(q, x);
        if (fx == 0.0) break;
        if ((fx < 0.0) == (fa < 0.0)) a = x; else b = x;
        const double d = horner<M - 1>(dq, x);
        double xn = x - ls_div(fx, d);
        if (!(xn > a && xn < b)) xn = 0.5 * (a + b);
        if (!(xn > a && xn < b)) break;                          // the bracket is down to neighbouring numbers
        if (fabs(xn - x) <= 2.220446049250313e-16 * fabs(xn)) { x = xn; break; }
        x = xn;
    }
    return x;
}
// W = true: called by a whole wave with wave-uniform arguments - the M monotone pieces are independent, so lane i isolates the root of
// piece i and the results are read back with v_readlane: one round of safeguarded Newton iterations per level instead of M (the
// contraction of a line search costs 10-30 k cycles, most of them in these iterations; same arithmetic per piece, same bits).
template <int M, bool W = false>
__device__ __forceinline__ void real_roots_in(const double (&q)[M + 1], double lo, double hi, double (&out)[M]) {
    static_assert(M >= 2 && M <= 4, "quadratic, cubic or quartic");
    if constexpr (M == 2) {
        const double A = q[0], B = q[1], C = q[2];
        const double D = B * B - 4 * A * C;
        double r0 = lo, r1 = lo;
        if (D >= 0) {
            const double sD = sqrt(D), t = B >= 0 ? -B - sD : -B + sD;          // t = -(B + sign(B) sqrt(D)): no cancellation
            const double u = ls_div(t, 2.0 * A), v = t != 0.0 ? ls_div(2.0 * C, t) : u;
            r0 = fmin(u, v); r1 = fmax(u, v);
            if (!(r0 == r0)) r0 = lo;                                            // (A == 0 or overflow: no usable breakpoint)
            if (!(r1 == r1)) r1 = lo;
        }
        out[0] = fmin(fmax(r0, lo), hi); out[1] = fmin(fmax(r1, lo), hi);
    } else {
        double dq[M], bp[M - 1];
#pragma unroll
        for (int i = 0; i < M; ++i) dq[i] = q[i] * (M - i);
        real_roots_in<M - 1, W>(dq, lo, hi, bp);
        if constexpr (W) {
            const int l = (int)(threadIdx.x & 63u);
            double a = lo, b = M == 1 ? hi : bp[0];                  // (lanes beyond M repeat piece 0)
#pragma unroll
            for (int i = 1; i < M; ++i) { if (l == i) { a = bp[i - 1]; b = i == M - 1 ? hi : bp[i]; } }
            const double fa = horner<M>(q, a), fb = horner<M>(q, b);
            double r = a;
            if (fa != 0.0 && fb == 0.0) r = b;
            else if ((fa < 0.0 && fb > 0.0) || (fa > 0.0 && fb < 0.0)) r = bracketed_root<M>(q, dq, a, b, fa);
#pragma unroll
            for (int i = 0; i < M; ++i)
                out[i] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(r), i), __builtin_amdgcn_readlane(__double2loint(r), i));
        } else {
#pragma unroll
            for (int i = 0; i < M; ++i) {
                const double a = i == 0 ? lo : bp[i - 1], b = i == M - 1 ? hi : bp[i];
                const double fa = horner<M>(q, a), fb = horner<M>(q, b);
                double r = a;
                if (fa != 0.0 && fb == 0.0) r = b;
                else if ((fa < 0.0 && fb > 0.0) || (fa > 0.0 && fb < 0.0)) r = bracketed_root<M>(q, dq, a, b, fa);
                out[i] = r;
            }
        }
    }
}

// poly_root_real_parts for d[0] x^(M-1) + ... + d[M-1]
template <int M, bool W = false>
__device__ __forceinline__ int poly_root_real_parts_n(const double (&d)[M], double lo, double hi, double (&out)[4]) {
    static_assert(M >= 2 && M <= 5, "derivative of a polynomial with 3..6 coefficients");
    int lead = 0;
    bool counting = true;
#pragma unroll
    for (int i = 0; i < M; ++i) { counting = counting && d[i] == 0.0; lead += counting ? 1 : 0; }
    double e[5] = {0.0, 0.0, 0.0, 0.0, 0.0};                 // e[i] = d[i + lead]
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int l = 0; l + i < M; ++l) e[i] = (lead == l) ? d[i + l] : e[i];
    const int deg = M - lead - 1;
    if (deg <= 0) return 0;
    if (deg == 1) { out[0] = ls_div(-e[1], e[0]); return 1; }
    if (deg == 2) {
        const double a = e[0], b = e[1], c = e[2];
        const double D = b * b - 4 * a * c, sD = sqrt(fabs(D));
        if (D >= 0) {
            if (b >= 0) { out[0] = ls_div(-b - sD, 2.0 * a); out[1] = ls_div(2.0 * c, -b - sD); }
            else        { out[0] = ls_div(2.0 * c, -b + sD); out[1] = ls_div(-b + sD, 2.0 * a); }
        } else { out[0] = ls_div(-b, 2.0 * a); out[1] = out[0]; }
        return 2;
    }
    if constexpr (M >= 5) {
        if (deg == 4) { const double q[5] = {e[0], e[1], e[2], e[3], e[4]}; real_roots_in<4, W>(q, lo, hi, out); return 4; }
    }
    if constexpr (M >= 4) {
        const double q[4] = {e[0], e[1], e[2], e[3]};
        double r[3];
        real_roots_in<3, W>(q, lo, hi, r);
        out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
        return 3;
    }
    return 0;
}

// Gaussian elimination with partial pivoting on registers: the pivot row is exchanged by selects
template <int N>
__device__ __forceinline__ bool solve_dense_n(double (&A)[N][N], double (&b)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        int piv = k;
        double pv = A[k][k];
#pragma unroll
        for (int i = k + 1; i < N; ++i) if (fabs(A[i][k]) > fabs(pv)) { piv = i; pv = A[i][k]; }
        if (pv == 0.0) return false;
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const bool sw = piv == i;
#pragma unroll
            for (int j = 0; j < N; ++j) { const double t = A[k][j], u = A[i][j]; A[k][j] = sw ? u : t; A[i][j] = sw ? t : u; }
            const double t = b[k], u = b[i]; b[k] = sw ? u : t; b[i] = sw ? t : u;
        }
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const double f = ls_div(A[i][k], A[k][k]);
#pragma unroll
            for (int j = k; j < N; ++j) A[i][j] -= f * A[k][j];
            b[i] -= f * b[k];
        }
    }
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {
        double s = b[k];
#pragma unroll
        for (int j = k + 1; j < N; ++j) s -= A[k][j] * b[j];
        b[k] = ls_div(s, A[k][k]);
    }
    return true;
}

// Ceres MinimizeInterpolatingPolynomial over [x_min, x_max] with N constraints: cx/cv/cg = abscissa, right-hand side and
// "is a gradient constraint" of constraint r, in the order value, gradient per sample; sx = the ns sample abscissae
template <int N, bool W = false>
__device__ __forceinline__ double minimize_interpolating_polynomial_n(const double (&cx)[6], const double (&cv)[6], const bool (&cg)[6],
                                                                      const double (&sx)[3], int ns, double x_min, double x_max) {
    constexpr int deg = N - 1;
    double A[N][N], poly[N];
#pragma unroll
    for (int r = 0; r < N; ++r) {
        double pw[N];                                            // pw[j] = x^j by repeated multiplication from 1 (Ceres: pow())
        pw[0] = 1.0;
#pragma unroll
        for (int j = 1; j < N; ++j) pw[j] = pw[j - 1] * cx[r];
#pragma unroll
        for (int j = 0; j <= deg; ++j) {
            const double value_row = pw[deg - j];
            const double gradient_row = j < deg ? (deg - j) * pw[j < deg ? deg - j - 1 : 0] : 0.0;
            A[r][j] = cg[r] ? gradient_row : value_row;
        }
        poly[r] = cv[r];
    }
    double best_x = (x_min + x_max) / 2.0;
    if (!solve_dense_n<N>(A, poly)) return best_x;
    double best_v = polyval_n<N>(poly, best_x), v;
    v = polyval_n<N>(poly, x_min); if (v < best_v) { best_v = v; best_x = x_min; }
    v = polyval_n<N>(poly, x_max); if (v < best_v) { best_v = v; best_x = x_max; }
    {
        double deriv[N - 1], roots[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < deg; ++i) deriv[i] = poly[i] * (deg - i);
        const int nr = poly_root_real_parts_n<N - 1, W>(deriv, x_min, x_max, roots);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i >= nr || roots[i] < x_min || roots[i] > x_max) continue;
            v = polyval_n<N>(poly, roots[i]);
            if (v < best_v) { best_v = v; best_x = roots[i]; }
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (i >= ns || sx[i] < x_min || sx[i] > x_max) continue;
        v = polyval_n<N>(poly, sx[i]);
        if (v < best_v) { best_v = v; best_x = sx[i]; }
    }
    return best_x;
}

template <bool W>
__device__ __forceinline__ double ls_next_step_impl(const LsSample &initial, const LsSample &previous, const LsSample &current,
                                                   double dir_max, int &n_iter) {
    if (++n_iter >= kLsMaxIterations) return -1.0;
    const double lo = kLsMaxContraction * current.x, hi = kLsMinContraction * current.x;
    double step;
    if (!current.value_valid) step = fmin(fmax(current.x * 0.5, lo), hi);
    else {
        // candidate constraints in Ceres' order (value, gradient per sample: initial, current, previous), compacted to the valid ones
        const bool with_prev = previous.value_valid;
        const int ns = with_prev ? 3 : 2;
        const double sx[3] = {initial.x, current.x, previous.x};
        const double px[6] = {initial.x, initial.x, current.x, current.x, previous.x, previous.x};
        const double pv[6] = {initial.value, initial.gradient, current.value, current.gradient, previous.value, previous.gradient};
        const bool ok[6] = {initial.value_valid, initial.gradient_valid, current.value_valid, current.gradient_valid,
                            with_prev && previous.value_valid, with_prev && previous.gradient_valid};
        double cx[6], cv[6]; bool cg[6];
        int n = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) { cx[r] = 0.0; cv[r] = 0.0; cg[r] = false; }
#pragma unroll
        for (int c = 0; c < 6; ++c) {
#pragma unroll
            for (int r = 0; r <= c; ++r) {
                const bool here = ok[c] && n == r;
                cx[r] = here ? px[c] : cx[r]; cv[r] = here ? pv[c] : cv[r]; cg[r] = here ? (c & 1) != 0 : cg[r];
            }
            n += ok[c] ? 1 : 0;
        }
        switch (n) {
            case 3: step = minimize_interpolating_polynomial_n<3, W>(cx, cv, cg, sx, ns, lo, hi); break;
            case 4: step = minimize_interpolating_polynomial_n<4, W>(cx, cv, cg, sx, ns, lo, hi); break;
            case 5: step = minimize_interpolating_polynomial_n<5, W>(cx, cv, cg, sx, ns, lo, hi); break;
            case 6: step = minimize_interpolating_polynomial_n<6, W>(cx, cv, cg, sx, ns, lo, hi); break;
            default: step = fmin(fmax(current.x * 0.5, lo), hi); break;      // < 3 constraints: not reachable (the initial sample always has both)
        }
    }
    if (step * dir_max < kLsMinStep) return -1.0;
    return step;
}
__device__ __noinline__ double ls_next_step_regs(const LsSample &initial, const LsSample &previous, const LsSample &current,
                                            double dir_max, int &n_iter) {
    return ls_next_step_impl<false>(initial, previous, current, dir_max, n_iter);
}
// the same for a WHOLE wave with wave-uniform arguments (all 64 lanes active): the pieces of the root isolation a lane each
__device__ __noinline__ double ls_next_step_wave(const LsSample &initial, const LsSample &previous, const LsSample &current,
                                            double dir_max, int &n_iter) {
    return ls_next_step_impl<true>(initial, previous, current, dir_max, n_iter);
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

// dpp_f64 for a source that stays live: the destination is a fresh register pair (an empty asm "defines" it), so no copy of v is
// made in front of the two v_mov_b32_dpp (dpp_f64 ties the destination to its source: right when the source dies there)
template <int CTRL>
__device__ __forceinline__ double dpp_f64_keep(double v) {
    int ul, uh;
    asm volatile("" : "=v"(ul), "=v"(uh));
    const int lo = __builtin_amdgcn_update_dpp(ul, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(uh, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// the lanes of the 4-lane banks in BANK take v through the permutation, the others keep `old`
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_f64_merge(double old, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xf, BANK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xf, BANK, false);
    return __hiloint2double(hi, lo);
}
constexpr int kDppIdentity = 0xE4;         // quad_perm [0,1,2,3]

template <int S, typename Op>
__device__ __forceinline__ double group_reduce(double v, Op op) {
    static_assert(S == 8 || S == 16 || S == 32 || S == 64, "group size");
    if (S == 8) {
        v = op(v, dpp_f64_keep<kDppHalfMirror>(v));
        v = op(v, dpp_f64_keep<kDppQuadXor1>(v));
        v = op(v, dpp_f64_keep<kDppQuadXor2>(v));
    } else {
        v = op(v, dpp_f64_keep<kDppRowRor + 8>(v));
        v = op(v, dpp_f64_keep<kDppRowRor + 4>(v));
        v = op(v, dpp_f64_keep<kDppRowRor + 2>(v));
        v = op(v, dpp_f64_keep<kDppRowRor + 1>(v));
        if (S >= 32) v = op(v, swizzle_f64<kSwizzleXor16>(v));
        if (S == 64) v = op(v, xor32_f64(v));
    }
    return v;
}
template <int S>
__device__ __forceinline__ double group_sum(double v) { return group_reduce<S>(v, [](double a, double b) { return a + b; }); }
template <int S>
__device__ __forceinline__ double group_max(double v) { return group_reduce<S>(v, [](double a, double b) { return fmax(a, b); }); }

// ---- 64-bit DPP (gfx90a+ "DP ALU DPP": fp64 VOP1/VOP2 take a DPP source operand, control row_newbcast only) ----
// row_newbcast:K hands every lane of a 16-lane row the value of the row's lane K, inside the consuming instruction: the pivot-row
// broadcasts of the elimination need neither the LDS crossbar (ds_swizzle: two LDS instructions and a round trip per double) nor a
// register.  bank_mask restricts the WRITE to 4-lane banks of the row (the other lanes keep dst), which splits a row into two
// 8-lane groups with their own pivot lanes.
// (asm, not __builtin_amdgcn_update_dpp: the builtin is declared for 32-bit operands in this clang.  hipcc does not see inside an
// asm statement: every statement below opens with s_nop 1, which covers "VALU write -> DPP read of the same VGPR" (2 wait states)
// whatever was scheduled in front of it.)
template <int K>
__device__ __forceinline__ double bcast16_f64(double v) {             // lane K of the own 16-lane row
    double r;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
    return r;
}
template <int K>
__device__ __forceinline__ double bcast8_f64(double v) {              // lane K of the own 8-lane group (two groups per row)
    double r;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0x3\n\t"
        "v_mov_b64_dpp %0, %1 row_newbcast:%3 row_mask:0xf bank_mask:0xc" : "=&v"(r) : "v"(v), "n"(K), "n"(K + 8));
    return r;
}
// h_i += nf * (h_i of lane K of the row), i = 0..N-1, as N v_fmac_f64_dpp
#define LFR_FMAC_DPP(i) "v_fmac_f64_dpp %" #i ", %" #i ", %[nf] row_newbcast:%[k] row_mask:0xf bank_mask:%[bank]\n\t"
template <int K, int BANK>
__device__ __forceinline__ void fmac_bcast(double nf, double &a) {
    asm volatile("s_nop 1\n\t" LFR_FMAC_DPP(0) : "+v"(a) : [nf] "v"(nf), [k] "n"(K), [bank] "n"(BANK));
}
template <int K, int BANK>
__device__ __forceinline__ void fmac_bcast(double nf, double &a, double &b) {
    asm volatile("s_nop 1\n\t" LFR_FMAC_DPP(0) LFR_FMAC_DPP(1) : "+v"(a), "+v"(b) : [nf] "v"(nf), [k] "n"(K), [bank] "n"(BANK));
}
template <int K, int BANK>
__device__ __forceinline__ void fmac_bcast(double nf, double &a, double &b, double &c) {
    asm volatile("s_nop 1\n\t" LFR_FMAC_DPP(0) LFR_FMAC_DPP(1) LFR_FMAC_DPP(2) : "+v"(a), "+v"(b), "+v"(c) : [nf] "v"(nf), [k] "n"(K), [bank] "n"(BANK));
}
template <int K, int BANK>
__device__ __forceinline__ void fmac_bcast(double nf, double &a, double &b, double &c, double &d) {
    asm volatile("s_nop 1\n\t" LFR_FMAC_DPP(0) LFR_FMAC_DPP(1) LFR_FMAC_DPP(2) LFR_FMAC_DPP(3)
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : [nf] "v"(nf), [k] "n"(K), [bank] "n"(BANK));
}
#undef LFR_FMAC_DPP
// columns [C0, CL) of h, then rhs, four to a statement
template <int K, int BANK, int C0, int CL>
__device__ __forceinline__ void fmac_bcast_row(double nf, double (&h)[CL], double &rhs) {
    constexpr int n = CL - C0;                       // live columns; rhs rides as column n
    if constexpr (n >= 4) {
        fmac_bcast<K, BANK>(nf, h[C0], h[C0 + 1], h[C0 + 2], h[C0 + 3]);
        fmac_bcast_row<K, BANK, C0 + 4, CL>(nf, h, rhs);
    } else if constexpr (n == 3) fmac_bcast<K, BANK>(nf, h[C0], h[C0 + 1], h[C0 + 2], rhs);
    else if constexpr (n == 2) fmac_bcast<K, BANK>(nf, h[C0], h[C0 + 1], rhs);
    else if constexpr (n == 1) fmac_bcast<K, BANK>(nf, h[C0], rhs);
    else fmac_bcast<K, BANK>(nf, rhs);
}

// Four sums at once.  S >= 16: a TRANSPOSED butterfly - after the xor-8 step the lower half of a 16-lane row carries the partial sums
// of a (b), the upper half those of c (d); after the mirror step inside 8 lanes each quad carries one of the four; two quad steps, then
// the totals are handed to every lane by row_newbcast.  7 + 7 + 7 + 3 + 3 + 4 = 31 instructions against 4 x 12 for four butterflies
// (4 x 20 with the copies the tied dpp_f64 made until round 6).  Every lane of the group ends with the bit-identical four sums.
#ifndef LFR_SUM4_WIDE
#define LFR_SUM4_WIDE 1
#endif
template <int S>
__device__ __forceinline__ void group_sum4(double &a, double &b, double &c, double &d) {
    if constexpr (S == 8 || (S >= 32 && LFR_SUM4_WIDE == 0)) {
        a = group_sum<S>(a); b = group_sum<S>(b); c = group_sum<S>(c); d = group_sum<S>(d);
    } else {
        double ra = dpp_f64_keep<kDppRowRor + 8>(c);                 // every lane: its partner's c
        ra = dpp_f64_merge<kDppRowRor + 8, 0x3>(ra, a);              // lanes 0-7: the partner's a instead
        const double ka = dpp_f64_merge<kDppIdentity, 0xc>(a, c);    // lanes 0-7: own a, lanes 8-15: own c
        const double P = ka + ra;
        double rb = dpp_f64_keep<kDppRowRor + 8>(d);
        rb = dpp_f64_merge<kDppRowRor + 8, 0x3>(rb, b);
        const double kb = dpp_f64_merge<kDppIdentity, 0xc>(b, d);
        const double Q = kb + rb;
        double rt = dpp_f64_keep<kDppHalfMirror>(Q);
        rt = dpp_f64_merge<kDppHalfMirror, 0x5>(rt, P);              // banks 0, 2 (lanes 0-3, 8-11): the mirror lane's P
        const double kt = dpp_f64_merge<kDppIdentity, 0xa>(P, Q);    // banks 1, 3 keep Q
        double T = kt + rt;
        T += dpp_f64_keep<kDppQuadXor2>(T);
        T += dpp_f64_keep<kDppQuadXor1>(T);                           // quads 0..3 of the row: sums of a, b, c, d over the row
        if (S >= 32) T += swizzle_f64<kSwizzleXor16>(T);
        if (S == 64) T += xor32_f64(T);
        asm("s_nop 1\n\t"
            "v_mov_b64_dpp %0, %4 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b64_dpp %1, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b64_dpp %2, %4 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b64_dpp %3, %4 row_newbcast:12 row_mask:0xf bank_mask:0xf"
            : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(T));
    }
}

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

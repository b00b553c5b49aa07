// Probe (round 3): what the pieces of the workgroup kernel's factorization cost on gfx950.
//   clocks      s_memtime ticks against the 100 MHz wall clock (are the "cycles" of the phase profiles core cycles?)
//   chains      dependent v_fma_f64; v_readlane -> v_fma_f64 -> v_readlane; 8 independent fma chains
//   diag16      LDL^T of a 16x16 tile, lane = row: (0) v_readlane broadcasts, (1) pivot column through LDS, (2) ds_swizzle broadcasts
//   lds         broadcast ds_read_b64 throughput, 1 and 8 waves
//   mfma        v_mfma_f64_16x16x4_f64: 1 / 2 / 4 accumulator chains per wave, 1 and 2 waves per SIMD; the blgp/neg bits
// build: hipcc --offload-arch=gfx950 -O3 -o diag16_probe diag16_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
using f64x4 = __attribute__((ext_vector_type(4))) double;

__device__ __forceinline__ double readlane_f64(double v, int k) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}
template <int PATTERN>
__device__ __forceinline__ double swizzle_f64(double v) {
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), PATTERN);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), PATTERN);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

__global__ void k_clock(unsigned long long *out) {
    const unsigned long long w0 = wall_clock64(), m0 = __builtin_amdgcn_s_memtime(), c0 = __builtin_readcyclecounter();
    double x = threadIdx.x;
    for (int i = 0; i < 200000; ++i) x = fma(x, 1.0000001, 1e-9);
    const unsigned long long w1 = wall_clock64(), m1 = __builtin_amdgcn_s_memtime(), c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = w1 - w0; out[1] = m1 - m0; out[2] = c1 - c0; out[3] = (unsigned long long)x; }
}

// MODE 0: dependent fma chain; 1: readlane -> fma chain; 2: 8 independent fma chains; 3: broadcast LDS reads
template <int MODE>
__global__ __launch_bounds__(512) void k_chain(double *out, unsigned long long *cyc, int iters) {
    __shared__ double lds[1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = 1.0 + 1e-6 * i;
    __syncthreads();
    double x = 1.0 + 1e-9 * lane, acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = x + i;
    const unsigned long long m0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {
        for (int i = 0; i < iters; ++i) x = fma(x, 0.999999, 1e-7);
    } else if (MODE == 1) {
        for (int i = 0; i < iters; ++i) { const double y = readlane_f64(x, i & 63); x = fma(x, 0.999999, y * 1e-9); }
    } else if (MODE == 2) {
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = fma(acc[c], 0.999999, 1e-7);
    } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += lds[(i * 8 + c) & 1023];        // same address in every lane
        }
    }
    const unsigned long long m1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; ++i) x += acc[i];
    if (threadIdx.x == 0) cyc[blockIdx.x] = m1 - m0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

// 16x16 LDL^T, lane = row (all four 16-lane groups compute the same).  tile: packed lower rows in LDS (136 doubles)
template <int VAR>
__global__ __launch_bounds__(512) void k_diag(const double *tile_in, double *tile_out, unsigned long long *cyc, int iters) {
    __shared__ double T[8][136 + 16];
    __shared__ double col[8][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15;
    unsigned long long total = 0;
    double a[16];
    for (int it = 0; it < iters; ++it) {
        for (int i = lane; i < 136; i += 64) T[wave][i] = tile_in[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const unsigned long long m0 = __builtin_amdgcn_s_memtime();
        const int base = r16 * (r16 + 1) / 2;
#pragma unroll
        for (int j = 0; j < 16; ++j) { const double v = T[wave][base + min(j, r16)]; a[j] = j <= r16 ? v : 0.0; }
        if (VAR == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const double dk = readlane_f64(a[k], k);
                const double ik = fast_rcp(dk);
                if (lane == k) T[wave][136 + k] = ik;
                const double lik = a[k] * ik;
#pragma unroll
                for (int j = k + 1; j < 16; ++j) a[j] = fma(-lik, readlane_f64(a[k], j), a[j]);
            }
        } else if (VAR == 1) {
            // pivot column through LDS: every lane writes its entry of column k, everybody reads the column back (broadcast reads).
            // The next pivot column entry is updated and written FIRST, the other updates of the step run under the round trip.
            if (lane < 16) col[wave][r16] = a[0];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                double c[16];
#pragma unroll
                for (int j = k; j < 16; ++j) c[j] = col[wave][j];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const double ik = fast_rcp(c[k]);
                if (lane == k) T[wave][136 + k] = ik;
                const double lik = a[k] * ik;
                if (k + 1 < 16) {
                    a[k + 1] = fma(-lik, c[k + 1], a[k + 1]);
                    if (lane < 16) col[wave][r16] = a[k + 1];
                }
#pragma unroll
                for (int j = k + 2; j < 16; ++j) a[j] = fma(-lik, c[j], a[j]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                // ds_swizzle bit mode: new lane = (lane & 0x10) | j inside each half of the wave: lane j of the own 16-lane group
                auto bc = [&](double v, int j) -> double {
                    switch (j) {
#define C(J) case J: return swizzle_f64<(J << 5) | 0x10>(v);
                        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15)
#undef C
                    }
                    return v;
                };
                const double dk = bc(a[k], k);
                const double ik = fast_rcp(dk);
                if (lane == k) T[wave][136 + k] = ik;
                const double lik = a[k] * ik;
#pragma unroll
                for (int j = k + 1; j < 16; ++j) a[j] = fma(-lik, bc(a[k], j), a[j]);
            }
        }
        if (lane < 16) {
#pragma unroll
            for (int j = 0; j < 16; ++j) if (j <= r16) T[wave][base + j] = a[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        total += __builtin_amdgcn_s_memtime() - m0;
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = total;
    if (blockIdx.x == 0 && wave == 0) for (int i = lane; i < 152; i += 64) tile_out[i] = T[0][i];
}

// CH accumulator chains per wave
template <int CH>
__global__ __launch_bounds__(512) void k_mfma(double *out, unsigned long long *cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f64x4 c[CH];
    for (int i = 0; i < CH; ++i) c[i] = {0.0, 0.0, 0.0, 0.0};
    const double a = 1.0 + lane * 1e-9, b = 0.5;
    const unsigned long long m0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < CH; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    const unsigned long long m1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
    for (int i = 0; i < CH; ++i) s += c[i][0] + c[i][3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = m1 - m0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_neg(double *out) {         // which operand do the blgp bits of the f64 MFMA negate?
    const int lane = threadIdx.x;
    const double a = 1.0 + (lane & 15), b = 2.0 + (lane >> 4);
    f64x4 c = {100.0, 100.0, 100.0, 100.0};
    f64x4 r0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    f64x4 r1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 1);
    f64x4 r2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 2);
    f64x4 r4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 4);
    if (lane == 5) { out[0] = r0[1]; out[1] = r1[1]; out[2] = r2[1]; out[3] = r4[1]; }
}

static double ticks(const std::vector<unsigned long long> &h, double per) { double s = 0; for (auto v : h) s += (double)v; return s / h.size() / per; }

template <typename F>
void timed(const char *name, int blocks, int threads, double units, F launch) {
    unsigned long long *cyc; double *out;
    hipMalloc(&cyc, blocks * 8); hipMalloc(&out, (size_t)blocks * threads * 8);
    launch(out, cyc);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); launch(out, cyc); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    printf("%-58s blocks %3d x %3d thr: %9.1f ticks/unit  %9.2f ns/unit wall  (%s)\n", name, blocks, threads, ticks(h, units), ms * 1e6 / units, hipGetErrorString(hipGetLastError()));
    hipFree(cyc); hipFree(out);
}

int main() {
    {
        unsigned long long *d, h[4]; hipMalloc(&d, 32);
        k_clock<<<1, 64>>>(d); hipDeviceSynchronize(); hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        printf("clocks: wall_clock64 %llu ticks (100 MHz => %.1f us), s_memtime %llu ticks => %.1f MHz, readcyclecounter %llu => %.1f MHz; 200000 dependent fma: %.2f s_memtime ticks each\n",
               h[0], h[0] / 100.0, h[1], h[1] / (h[0] / 100.0), h[2], h[2] / (h[0] / 100.0), (double)h[1] / 200000);
    }
    const int it = 20000;
    for (int blocks : {1, 256}) for (int thr : {64, 512}) {
        timed("dependent v_fma_f64 chain (per fma)", blocks, thr, it, [&](double *o, unsigned long long *c) { k_chain<0><<<blocks, thr>>>(o, c, it); });
        timed("readlane -> fma chain (per link)", blocks, thr, it, [&](double *o, unsigned long long *c) { k_chain<1><<<blocks, thr>>>(o, c, it); });
        timed("8 independent fma chains (per fma)", blocks, thr, it * 8.0, [&](double *o, unsigned long long *c) { k_chain<2><<<blocks, thr>>>(o, c, it); });
        timed("broadcast ds_read_b64 + add (per read)", blocks, thr, it * 8.0, [&](double *o, unsigned long long *c) { k_chain<3><<<blocks, thr>>>(o, c, it); });
    }
    {   // diag16
        std::vector<double> A(256), tile(136);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) A[i * 16 + j] = (i == j ? 20.0 : 0.0) + std::cos(0.37 * (i + 1) * (j + 1)) + std::cos(0.37 * (j + 1) * (i + 1));
        for (int i = 0; i < 16; ++i) for (int j = 0; j <= i; ++j) tile[i * (i + 1) / 2 + j] = 0.5 * (A[i * 16 + j] + A[j * 16 + i]);
        // reference LDL^T (unscaled columns, d on the diagonal)
        std::vector<double> R(tile);
        auto at = [&](std::vector<double> &M, int i, int j) -> double & { return M[i * (i + 1) / 2 + j]; };
        for (int k = 0; k < 16; ++k) for (int i = k + 1; i < 16; ++i) { const double l = at(R, i, k) / at(R, k, k); for (int j = k + 1; j <= i; ++j) at(R, i, j) -= l * at(R, j, k); }
        double *din, *dout; hipMalloc(&din, 136 * 8); hipMalloc(&dout, 152 * 8); hipMemcpy(din, tile.data(), 136 * 8, hipMemcpyHostToDevice);
        auto check = [&](const char *nm) {
            std::vector<double> o(152); hipMemcpy(o.data(), dout, 152 * 8, hipMemcpyDeviceToHost);
            double e = 0, ei = 0; for (int i = 0; i < 136; ++i) e = std::fmax(e, std::fabs(o[i] - R[i]));
            for (int k = 0; k < 16; ++k) ei = std::fmax(ei, std::fabs(o[136 + k] - 1.0 / at(R, k, k)));
            printf("   %s: max |tile - reference| %.2e, max |1/d - reference| %.2e\n", nm, e, ei);
        };
        const int di = 2000;
        for (int blocks : {1, 256}) for (int thr : {64, 512}) {
            timed("diag16 LDL^T, v_readlane broadcasts (per tile)", blocks, thr, di, [&](double *, unsigned long long *c) { k_diag<0><<<blocks, thr>>>(din, dout, c, di); });
            if (blocks == 1 && thr == 64) check("readlane");
            timed("diag16 LDL^T, pivot column through LDS (per tile)", blocks, thr, di, [&](double *, unsigned long long *c) { k_diag<1><<<blocks, thr>>>(din, dout, c, di); });
            if (blocks == 1 && thr == 64) check("lds column");
            timed("diag16 LDL^T, ds_swizzle broadcasts (per tile)", blocks, thr, di, [&](double *, unsigned long long *c) { k_diag<2><<<blocks, thr>>>(din, dout, c, di); });
            if (blocks == 1 && thr == 64) check("swizzle");
        }
    }
    for (int blocks : {1, 256}) for (int thr : {256, 512}) {
        timed("mfma_f64_16x16x4: 1 chain per wave (per mfma)", blocks, thr, it, [&](double *o, unsigned long long *c) { k_mfma<1><<<blocks, thr>>>(o, c, it); });
        timed("mfma_f64_16x16x4: 2 chains per wave (per mfma)", blocks, thr, it * 2.0, [&](double *o, unsigned long long *c) { k_mfma<2><<<blocks, thr>>>(o, c, it); });
        timed("mfma_f64_16x16x4: 4 chains per wave (per mfma)", blocks, thr, it * 4.0, [&](double *o, unsigned long long *c) { k_mfma<4><<<blocks, thr>>>(o, c, it); });
    }
    {
        double *d, h[4]; hipMalloc(&d, 32); k_neg<<<1, 64>>>(d); hipDeviceSynchronize(); hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        printf("mfma f64 blgp bits (lane 5, reg 1): plain %.1f, blgp=1 %.1f, blgp=2 %.1f, blgp=4 %.1f   (a=6.., b=2..: a.b summed over 4 k)\n", h[0], h[1], h[2], h[3]);
    }
    return 0;
}

// Probe: cost of v_mfma_f64_16x16x4_f64 and of the factorization's tile step on gfx950, one workgroup per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f64x4 = __attribute__((ext_vector_type(4))) double;
__device__ __forceinline__ uint32_t tri(int i, int j) { return (__umul24((unsigned)i, (unsigned)i + 1u) >> 1) + (unsigned)j; }

// mode 0: dependent MFMA chain; 1: two independent chains; 2: tile step (loads + 2 MFMA + stores) per wave; 3: tile step without MFMA
template <int MODE>
__global__ __launch_bounds__(256) void probe(double *out, unsigned long long *cyc, int iters, int n) {
    extern __shared__ double Mat[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
    for (int i = tid; i < n * (n + 1) / 2; i += 256) Mat[i] = 1e-3 * (i % 97);
    __syncthreads();
    f64x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    double a = 1.0 + lane * 1e-9, b = 0.5;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long m0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    } else if (MODE == 1) {
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
        }
    } else {
        const int kb = 0, ke = 8, mt = (n - ke) >> 4;
        for (int it = 0; it < iters; ++it) {
            int t = 0;
            for (int I = 0; I < mt; ++I)
                for (int J = 0; J < I; ++J, ++t) {
                    if ((t & 3) != wave) continue;
                    const int ia = ke + 16 * I + r16, jb = ke + 16 * J + r16, row0 = ke + 16 * I + kq;
                    const uint32_t oa = tri(ia, kb + kq), ob = tri(jb, kb + kq);
                    const double a0 = Mat[oa], a1 = Mat[oa + 4], b0 = Mat[ob] * 1e-6, b1 = Mat[ob + 4] * 1e-6;
                    f64x4 c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) c[r] = Mat[tri(row0 + 4 * r, jb)];
                    if (MODE == 2) {
                        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c, 0, 0, 0);
                    } else {
                        c[0] += a0 * b0; c[1] += a1 * b1; c[2] += a0 * b1; c[3] += a1 * b0;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) Mat[tri(row0 + 4 * r, jb)] = c[r];
                }
            __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long m1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = m1 - m0; }
    out[blockIdx.x * 256 + tid] = c0[0] + c1[1] + Mat[tid];
}

template <int MODE>
void run(const char *name, int blocks, int iters, int n, double units_per_iter) {
    double *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * 256 * sizeof(double)); hipMalloc(&cyc, blocks * 2 * sizeof(unsigned long long));
    const size_t lds = (size_t)n * (n + 1) / 2 * 8;
    hipFuncSetAttribute((const void *)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 256, lds>>>(out, cyc, iters, n);
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256, lds>>>(out, cyc, iters, n);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks); hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    printf("%-34s blocks %4d: %8.3f ms; per unit: %9.1f ns wall, %9.1f readcyclecounter ticks, %9.1f s_memtime ticks (%s)\n", name, blocks, ms,
           ms * 1e6 / (iters * units_per_iter), (double)h[0] / (iters * units_per_iter), (double)h[1] / (iters * units_per_iter), hipGetErrorString(hipGetLastError()));
    hipFree(out); hipFree(cyc);
}

int main() {
    const int n = 176;                       // mt = 10 full block rows below an 8-column panel -> 45 interior tiles
    for (int blocks : {1, 256}) {
        run<0>("dependent mfma_f64 chain", blocks, 20000, n, 1);
        run<1>("two independent chains (per mfma)", blocks, 20000, n, 2);
        run<2>("tile step x45 tiles / 4 waves", blocks, 2000, n, 45.0 / 4);
        run<3>("tile step, VALU instead of MFMA", blocks, 2000, n, 45.0 / 4);
    }
    return 0;
}

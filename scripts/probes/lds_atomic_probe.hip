// Probe (round 6): what an LDS fp64 atomic costs on gfx950, as the packed kernel's assembly issues them (no-return ds_add_f64 from
// one-wave workgroups, eight waves per CU), against plain stores and against VALU work running beside them.
//   pattern 0  64 distinct addresses, conflict-free (stride 8 B)
//   pattern 1  the packed kernel's shape: four 16-lane groups with their own 2.5-KB region, inside a group the lanes hit 6 diagonal
//              addresses (row r of a 17-column matrix: r * 18 * 8 bytes), i.e. 2-3 lanes per address
//   pattern 2  every lane of a 16-lane group the same address (16-way same-address)
//   pattern 3  all 64 lanes one address
// Reports cycles per wave-instruction (s_memtime over a loop of N instructions, one s_waitcnt at the end) with 1 and 8 one-wave
// workgroups per CU, for ds_add_f64, ds_write_b64, and ds_add_f64 interleaved with 8 fp64 FMAs per atomic.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o lds_atomic_probe lds_atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>   // 0: ds_add_f64, 1: ds_write_b64, 2: ds_add_f64 + 8 FMAs, 3: 8 FMAs only, 4: ds_add_rtn_f64 (value used)
__global__ __launch_bounds__(64) void k_probe(int pattern, int iters, unsigned long long *cyc, double *sink) {
    __shared__ double lds[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) lds[i] = 0.0;
    __syncthreads();
    int idx;
    if (pattern == 0) idx = lane;
    else if (pattern == 1) idx = (lane >> 4) * 313 + ((lane * 7) % 6) * 2 * 18;
    else if (pattern == 2) idx = (lane >> 4) * 313;
    else idx = 0;
    double *p = &lds[idx];
    double v = 1.0 + 1e-9 * lane, acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = v + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0 || OP == 2) __hip_atomic_fetch_add(p + (u & 1), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (OP == 1) { asm volatile("ds_write_b64 %0, %1" :: "v"((unsigned)(size_t)(p + (u & 1))), "v"(v) : "memory"); }
            if (OP == 4) acc[u] += __hip_atomic_fetch_add(p + (u & 1), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (OP == 2 || OP == 3) {
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = fma(acc[c], 0.999999, 1e-7);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    if (lane == 0) { atomicAdd(&cyc[0], t1 - t0); atomicAdd(&cyc[1], 1ull); }
    if (s == 123.456) sink[0] = s + lds[lane];
}

template <int OP>
static void run(const char *name, int pattern, int blocks_per_cu) {
    unsigned long long *d_c; double *d_s;
    hipMalloc(&d_c, 16); hipMalloc(&d_s, 8);
    hipMemset(d_c, 0, 16);
    const int iters = 2000;
    hipLaunchKernelGGL(k_probe<OP>, dim3(256 * blocks_per_cu), dim3(64), 0, 0, pattern, iters, d_c, d_s);
    hipDeviceSynchronize();
    unsigned long long h[2];
    hipMemcpy(h, d_c, 16, hipMemcpyDeviceToHost);
    printf("%-28s pattern %d  %d wave(s)/CU: %7.1f cycles per wave-instruction (%.1f per CU-wide instruction slot)\n", name, pattern, blocks_per_cu,
           (double)h[0] / h[1] / (iters * 8.0), (double)h[0] / h[1] / (iters * 8.0) / blocks_per_cu);
    hipFree(d_c); hipFree(d_s);
}

int main() {
    for (int bpc : {1, 8}) {
        for (int pat = 0; pat < 4; ++pat) {
            run<0>("ds_add_f64", pat, bpc);
            run<1>("ds_write_b64", pat, bpc);
            run<4>("ds_add_rtn_f64", pat, bpc);
        }
        run<3>("8 fma only", 0, bpc);
        for (int pat = 0; pat < 3; ++pat) run<2>("ds_add_f64 + 8 fma", pat, bpc);
    }
    return 0;
}

// Microbenchmark behind DESIGN.md §6: what the "Total" span pays outside the kernels on this box.
//   hipMalloc/hipFree cost by size, hipHostMalloc cost, H2D bandwidth pageable vs pinned, D2H pinned,
//   kernel-driven zero-copy reads of pinned host memory (72-byte rows gathered in random order).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/pcie_probe.hip -o gpurun_out/pcie_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_gather(int64_t rows, const uint32_t *perm, const uint4 *src, uint4 *dst) {      // 80-byte rows, 5 x 16 B
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = t / 5; const int c = (int)(t - 5 * r);
    if (r >= rows) return;
    dst[5 * r + c] = src[5 * (int64_t)perm[r] + c];
}
__global__ void k_touch(int64_t n, uint4 *p) { const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = make_uint4(1, 2, 3, 4); }

int main() {
    CK(hipSetDevice(0)); CK(hipFree(nullptr));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (size_t mb : {1, 16, 64, 256, 512, 1024, 2048}) {
        void *p = nullptr;
        double t0 = now_ms(); CK(hipMalloc(&p, mb << 20)); double t1 = now_ms();
        hipLaunchKernelGGL(k_touch, dim3((unsigned)(((mb << 20) / 16 + 255) / 256)), dim3(256), 0, st, (int64_t)((mb << 20) / 16), (uint4 *)p);
        CK(hipStreamSynchronize(st)); double t2 = now_ms();
        CK(hipFree(p)); double t3 = now_ms();
        void *q = nullptr; CK(hipMalloc(&q, mb << 20)); double t4 = now_ms(); CK(hipFree(q));
        printf("hipMalloc %5zu MB: %.3f ms  first touch %.3f ms  hipFree %.3f ms  second hipMalloc %.3f ms\n", mb, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
    }
    const size_t bytes = (size_t)360 << 20;
    void *dev = nullptr; CK(hipMalloc(&dev, bytes));
    {   // pageable
        char *h = (char *)malloc(bytes); memset(h, 1, bytes);
        for (int rep = 0; rep < 3; ++rep) { double t0 = now_ms(); CK(hipMemcpyAsync(dev, h, bytes, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); double t1 = now_ms();
            printf("H2D pageable 360 MB: %.2f ms = %.1f GB/s\n", t1 - t0, bytes / (t1 - t0) / 1e6); }
        double t0 = now_ms(); CK(hipHostRegister(h, bytes, hipHostRegisterDefault)); double t1 = now_ms();
        printf("hipHostRegister 360 MB: %.2f ms\n", t1 - t0);
        for (int rep = 0; rep < 2; ++rep) { t0 = now_ms(); CK(hipMemcpyAsync(dev, h, bytes, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); t1 = now_ms();
            printf("H2D registered 360 MB: %.2f ms = %.1f GB/s\n", t1 - t0, bytes / (t1 - t0) / 1e6); }
        CK(hipHostUnregister(h)); free(h);
    }
    char *pin = nullptr;
    { double t0 = now_ms(); CK(hipHostMalloc((void **)&pin, bytes, hipHostMallocDefault)); double t1 = now_ms(); memset(pin, 2, bytes); double t2 = now_ms();
      printf("hipHostMalloc 360 MB: %.2f ms, memset %.2f ms\n", t1 - t0, t2 - t1); }
    for (int rep = 0; rep < 3; ++rep) { double t0 = now_ms(); CK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); double t1 = now_ms();
        printf("H2D pinned 360 MB: %.2f ms = %.1f GB/s\n", t1 - t0, bytes / (t1 - t0) / 1e6); }
    {   // two streams, two halves
        hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        for (int rep = 0; rep < 2; ++rep) { double t0 = now_ms(); CK(hipMemcpyAsync(dev, pin, bytes / 2, hipMemcpyHostToDevice, st));
            CK(hipMemcpyAsync((char *)dev + bytes / 2, pin + bytes / 2, bytes / 2, hipMemcpyHostToDevice, s2)); CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(s2)); double t1 = now_ms();
            printf("H2D pinned 2 streams 360 MB: %.2f ms = %.1f GB/s\n", t1 - t0, bytes / (t1 - t0) / 1e6); }
    }
    for (size_t mb : {1, 14, 64}) for (int rep = 0; rep < 2; ++rep) { double t0 = now_ms(); CK(hipMemcpyAsync(pin, dev, mb << 20, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); double t1 = now_ms();
        printf("D2H pinned %zu MB: %.3f ms = %.1f GB/s\n", mb, t1 - t0, (double)(mb << 20) / (t1 - t0) / 1e6); }
    {   // small sync latency
        int *flag; CK(hipHostMalloc((void **)&flag, 64, hipHostMallocDefault));
        for (int rep = 0; rep < 3; ++rep) { double t0 = now_ms(); CK(hipMemcpyAsync(flag, dev, 64, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); double t1 = now_ms(); printf("64-B D2H + sync: %.4f ms\n", t1 - t0); }
    }
    {   // zero-copy gather: kernel reads 80-byte rows of pinned host memory in random order, writes HBM
        const int64_t rows = (int64_t)(bytes / 80);
        std::vector<uint32_t> perm(rows);
        for (int64_t i = 0; i < rows; ++i) perm[i] = (uint32_t)i;
        uint64_t s = 88172645463325252ull;
        for (int64_t i = rows - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; std::swap(perm[i], perm[s % (uint64_t)(i + 1)]); }
        uint32_t *dperm; CK(hipMalloc(&dperm, 4 * rows)); CK(hipMemcpy(dperm, perm.data(), 4 * rows, hipMemcpyHostToDevice));
        void *dst; CK(hipMalloc(&dst, bytes));
        for (int64_t frac : {1, 8}) for (int rep = 0; rep < 2; ++rep) {
            const int64_t r = rows / frac;
            double t0 = now_ms();
            hipLaunchKernelGGL(k_gather, dim3((unsigned)((5 * r + 255) / 256)), dim3(256), 0, st, r, dperm, (const uint4 *)pin, (uint4 *)dst);
            CK(hipStreamSynchronize(st)); double t1 = now_ms();
            printf("zero-copy random gather of %lld 80-B rows from pinned host: %.2f ms = %.1f GB/s\n", (long long)r, t1 - t0, 80.0 * r / (t1 - t0) / 1e6);
        }
        for (int rep = 0; rep < 2; ++rep) { double t0 = now_ms();
            hipLaunchKernelGGL(k_gather, dim3((unsigned)((5 * rows + 255) / 256)), dim3(256), 0, st, rows, dperm, (const uint4 *)dev, (uint4 *)dst);
            CK(hipStreamSynchronize(st)); double t1 = now_ms();
            printf("same gather from HBM: %.3f ms = %.1f GB/s (read+write %.1f GB/s)\n", t1 - t0, 80.0 * rows / (t1 - t0) / 1e6, 160.0 * rows / (t1 - t0) / 1e6); }
    }
    {   // host memcpy speed (download into a caller buffer)
        char *dsth = (char *)malloc((size_t)14 << 20); memset(dsth, 0, (size_t)14 << 20);
        for (int rep = 0; rep < 3; ++rep) { double t0 = now_ms(); memcpy(dsth, pin, (size_t)14 << 20); double t1 = now_ms(); printf("host memcpy 14 MB from pinned: %.3f ms\n", t1 - t0); }
    }
    return 0;
}

// Tile shapes of the one-sweep radix passes at the pipeline's sizes (config 4): build with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I local-feature-refinement_amd/csrc [-DLFR_SORT_ONESWEEP_CONFIG='rocprim::radix_sort_onesweep_config<...>'] \
//         scripts/probes/sort_probe.hip -o sort_probe
// prints microseconds per sort (median of 20) for: 2.5 M (u64 key, u32 value) 52 bits; 2.5 M (u32, u32) 18 bits; 0.88 M (u32, u32) 19 bits.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

#include "lfr_sort.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class K>
static int run(const char *what, size_t n, int bits) {
    std::mt19937_64 rng(7);
    std::vector<K> hk(n);
    std::vector<uint32_t> hv(n);
    const K mask = bits >= (int)(8 * sizeof(K)) ? ~(K)0 : (((K)1 << bits) - 1);
    for (size_t i = 0; i < n; ++i) { hk[i] = (K)rng() & mask; hv[i] = (uint32_t)i; }
    K *kin, *kout; uint32_t *vin, *vout; void *tmp; size_t bytes = 0;
    CK(hipMalloc(&kin, sizeof(K) * n)); CK(hipMalloc(&kout, sizeof(K) * n)); CK(hipMalloc(&vin, 4 * n)); CK(hipMalloc(&vout, 4 * n));
    CK(hipMemcpy(kin, hk.data(), sizeof(K) * n, hipMemcpyHostToDevice)); CK(hipMemcpy(vin, hv.data(), 4 * n, hipMemcpyHostToDevice));
    CK(lfr::sort_pairs_raw(nullptr, bytes, kin, kout, vin, vout, (int64_t)n, 0, bits, nullptr));
    CK(hipMalloc(&tmp, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> t;
    for (int r = 0; r < 23; ++r) {
        CK(hipEventRecord(a, nullptr));
        CK(lfr::sort_pairs_raw(tmp, bytes, kin, kout, vin, vout, (int64_t)n, 0, bits, nullptr));
        CK(hipEventRecord(b, nullptr)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (r >= 3) t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    std::vector<K> out(n);
    CK(hipMemcpy(out.data(), kout, sizeof(K) * n, hipMemcpyDeviceToHost));
    const bool ok = std::is_sorted(out.begin(), out.end());
    printf("%-34s %8.1f us (min %.1f)  %s\n", what, t[t.size() / 2], t[0], ok ? "sorted" : "NOT SORTED");
    (void)hipFree(kin); (void)hipFree(kout); (void)hipFree(vin); (void)hipFree(vout); (void)hipFree(tmp);
    return ok ? 0 : 1;
}

int main() {
    int rc = 0;
    rc |= run<unsigned long long>("2.5 M u64 keys, 52 bits", 2500535, 52);
    rc |= run<uint32_t>("2.5 M u32 keys, 18 bits", 2500535, 18);
    rc |= run<uint32_t>("0.88 M u32 keys, 19 bits", 882435, 19);
    return rc;
}

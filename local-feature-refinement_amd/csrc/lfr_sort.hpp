// Device primitives of the pipeline (graph stage, batch assembly): the stable key/value sort, and a one-launch exclusive prefix sum (below).
//
// rocPRIM's one-sweep radix sort enqueues, on top of its kernels, one hipMemsetAsync for the digit histograms and TWO per digit place
// (the decoupled-look-back states of that pass and its ordered block counter): a 52-bit sort is 7 places = 15 fills of ~4-5 us each, and
// the config-4 pipeline carried ~45 of them (0.22 ms of a 2.5 ms chain).  Nothing makes them per pass: the states of different places
// are different words.  onesweep_pairs() below drives rocPRIM's own kernels (rocprim::detail::onesweep_histograms /
// onesweep_scan_histograms / onesweep_iteration, unchanged) but gives every place its own look-back states and block counter, laid out
// behind the histograms in ONE region that ONE fill clears: 1 fill + 2 + places kernels per sort.  A radix sort is stable, so the result
// is the same permutation whichever driver produced it.
//
// Below the merge-sort limit of LfrRadixSortConfig (256 K items), for aliasing buffers and for >= 2^30 items the call goes to
// rocprim::radix_sort_pairs as before.  Only .hip translation units see this header.
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>                // (rocPRIM's texture_cache_iterator.hpp calls memset without it)

#include <rocprim/rocprim.hpp>

#include <cstdint>
#include <cstdlib>
#include <variant>

namespace lfr {

using LfrRadixSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, (size_t)256 * 1024>;
constexpr size_t kSortMergeLimit = (size_t)256 * 1024;
constexpr int64_t kOnesweepSmallFrom = 32 * 1024;      // short keys (<= 20 bits) take the one-sweep driver from this many items on

// The driver below calls rocPRIM's detail kernels with the signatures of rocPRIM 4.2.0 (ROCm 7.2).  Another version takes the library's
// public entry point for every sort (the results are the same; the fills come back) until the driver has been checked against it.
#if !defined(LFR_SORT_OWN_DRIVER)
#if ROCPRIM_VERSION == 400200
#define LFR_SORT_OWN_DRIVER 1
#else
#define LFR_SORT_OWN_DRIVER 0
#endif
#endif

namespace sortdetail {
inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
template <class K, class V>
inline bool aliasing(const K *kin, const K *kout, const V *vin, const V *vout, int64_t n) {
    auto overlap = [n](const void *a, size_t sa, const void *b, size_t sb) {
        const char *x = static_cast<const char *>(a), *y = static_cast<const char *>(b);
        return x < y + sb * (size_t)n && y < x + sa * (size_t)n;
    };
    return overlap(kin, sizeof(K), kout, sizeof(K)) || overlap(vin, sizeof(V), vout, sizeof(V));
}
inline bool single_fill_enabled() {
    static const bool on = [] { const char *s = std::getenv("LFR_SORT_ROCPRIM"); return LFR_SORT_OWN_DRIVER && !(s && s[0] == '1'); }();
    return on;
}
#if LFR_SORT_OWN_DRIVER
namespace rd = ::rocprim::detail;


// bytes == 0 on entry with tmp == nullptr: size query.  Keys/values must not alias their outputs.
// Tile shapes and digit widths.  rocPRIM has no tuned one-sweep configuration for gfx950; what it falls back to (8-bit digits) took
// 277 / 112 / 96 us for the three sorts of config 4 - 2.5 M (u64, u32) pairs over 52 bits, 2.5 M (u32, u32) over 18, 0.88 M (u32, u32)
// over 19 - on an MI355X.  Measured with scripts/probes/sort_probe.hip (profiles/r06_ab/sort_probe.txt):
//   tiles   histogram 1024 x 8 with sort 1024 x 8 is the best of ten shapes for the long sorts (248 / 105 us), sort 1024 x 4 for the
//           short one (65 us: below ~2 M pairs a tile of 8192 leaves half of the 256 CUs without a block);
//   digits  a pass costs about the same for 8 and 9 bits and a fifth more for 10 (11 bits do not fit the LDS), so the width that needs
//           the fewest passes wins, the narrower one on a tie: 52 bits in six 9-bit passes 226 us, 18 bits in two 9-bit passes 74 us,
//           19 bits in two 10-bit passes 55 us.
template <unsigned int Bits, unsigned int Items>
using OnesweepShape = rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, Items>, Bits, rocprim::block_radix_rank_algorithm::match>;
constexpr unsigned int kOnesweepShortBelow = 2u << 20;
inline unsigned int onesweep_digit_bits(unsigned int key_bits) {
    unsigned int best = 8, passes = (key_bits + 7) / 8;
    for (unsigned int d = 9; d <= 10; ++d) if ((key_bits + d - 1) / d < passes) { best = d; passes = (key_bits + d - 1) / d; }
    return best;
}

template <class sort_config, class K, class V>
hipError_t onesweep_pairs(void *tmp, size_t &bytes, const K *kin, K *kout, const V *vin, V *vout, unsigned int n, unsigned int begin_bit,
                          unsigned int end_bit, hipStream_t st) {
    using offset_type = unsigned int;
    using config = rd::wrapped_radix_sort_onesweep_config<sort_config, K, V>;
    using decomposer_t = ::rocprim::identity_decomposer;

    bool atomic_id = false;
    hipError_t e = rd::check_if_using_atomic_block_id(st, atomic_id);
    if (e != hipSuccess) return e;
    rd::target_arch arch;
    if ((e = rd::host_target_arch(st, arch)) != hipSuccess) return e;
    const rd::radix_sort_onesweep_config_params params = rd::dispatch_target_arch<config, false>(arch);

    const unsigned int radix_size = 1u << params.radix_bits_per_place;
    const unsigned int places = (end_bit - begin_bit + params.radix_bits_per_place - 1) / params.radix_bits_per_place;
    const unsigned int bins = radix_size * places;
    const unsigned int hist_items = params.histogram.block_size * params.histogram.items_per_thread;
    const unsigned int hist_blocks = (n + hist_items - 1) / hist_items, hist_full = n % hist_items == 0 ? hist_blocks : hist_blocks - 1;
    const unsigned int sort_items = params.sort.block_size * params.sort.items_per_thread;
    const unsigned int blocks = (n + sort_items - 1) / sort_items, full_blocks = n % sort_items == 0 ? blocks : blocks - 1;
    const size_t states_per_place = (size_t)radix_size * blocks;

    // [ histograms | look-back states of every place | block counter of every place ]  <- one fill   | scratch offsets | keys | values
    const size_t off_states = up256(sizeof(offset_type) * bins);
    const size_t off_ids = off_states + up256(sizeof(rd::onesweep_lookback_state) * states_per_place * places);
    const size_t zero_bytes = off_ids + up256(sizeof(unsigned int) * places);
    const size_t off_tmp_offsets = zero_bytes;
    const size_t off_keys = off_tmp_offsets + up256(sizeof(offset_type) * radix_size);
    const size_t off_vals = off_keys + up256(sizeof(K) * (size_t)n);
    const size_t total = off_vals + up256(sizeof(V) * (size_t)n);
    if (!tmp) { bytes = total; return hipSuccess; }
    if (bytes < total) return hipErrorInvalidValue;

    char *base = static_cast<char *>(tmp);
    offset_type *digit_offsets = reinterpret_cast<offset_type *>(base);
    rd::onesweep_lookback_state *states = reinterpret_cast<rd::onesweep_lookback_state *>(base + off_states);
    unsigned int *ids = reinterpret_cast<unsigned int *>(base + off_ids);
    offset_type *offsets_scratch = reinterpret_cast<offset_type *>(base + off_tmp_offsets);
    K *ktmp = reinterpret_cast<K *>(base + off_keys);
    V *vtmp = reinterpret_cast<V *>(base + off_vals);

    if ((e = hipMemsetAsync(base, 0, zero_bytes, st)) != hipSuccess) return e;

    const decomposer_t decomposer{};
    {   // digit histograms of every place in one pass over the keys, then their exclusive scans
        auto histograms = [=](auto arch_config) {
            static constexpr rd::radix_sort_onesweep_config_params p = decltype(arch_config)::params;
            rd::onesweep_histograms<p.histogram.block_size, p.histogram.items_per_thread, p.radix_bits_per_place, false>(
                kin, digit_offsets, (offset_type)n, (offset_type)hist_full, decomposer, begin_bit, end_bit);
        };
        if ((e = rd::execute_launch_plan<config, decltype(histograms), rd::radix_sort_onesweep_histogram_config_selector>(
                 arch, histograms, dim3(hist_blocks), dim3(params.histogram.block_size), 0, st)) != hipSuccess)
            return e;
        auto scans = [=](auto arch_config) {
            static constexpr rd::radix_sort_onesweep_config_params p = decltype(arch_config)::params;
            rd::onesweep_scan_histograms<p.histogram.block_size, p.radix_bits_per_place>(digit_offsets);
        };
        if ((e = rd::execute_launch_plan<config, decltype(scans), rd::radix_sort_onesweep_histogram_config_selector>(
                 arch, scans, dim3(places), dim3(params.histogram.block_size), 0, st)) != hipSuccess)
            return e;
    }

    const auto variant = rd::constexpr_value_variant<bool, false, true>::create(atomic_id);
    return std::visit(
        [&](auto use_atomic) -> hipError_t {
            using bid_type = rd::block_id_wrapper<unsigned int, use_atomic>;
            bool to_output = (places - 1) % 2 == 0;      // ping-pong so that the last place lands in the outputs
            bool from_input = true;
            unsigned int place = 0;
            for (unsigned int bit = begin_bit; bit < end_bit; bit += params.radix_bits_per_place, ++place) {
                const unsigned int cur_bits = ::rocprim::min(params.radix_bits_per_place, end_bit - bit);
                offset_type *offs_in = digit_offsets + (size_t)place * radix_size;
                rd::onesweep_lookback_state *lb = states + states_per_place * place;
                auto bid = bid_type::create(ids + place);
                auto launch = [&](const K *ki, K *ko, const V *vi, V *vo) {
                    auto iteration = [=](auto arch_config) {
                        static constexpr auto p = decltype(arch_config)::params;
                        rd::onesweep_iteration<p.sort.block_size, p.sort.items_per_thread, p.radix_bits_per_place, false, p.radix_rank_algorithm>(
                            ki, ko, vi, vo, n, offs_in, offsets_scratch, lb, decomposer, bit, cur_bits, full_blocks, bid);
                    };
                    return rd::execute_launch_plan<config, decltype(iteration), rd::radix_sort_onesweep_sort_config_selector>(
                        arch, iteration, dim3(blocks), dim3(params.sort.block_size), 0, st);
                };
                hipError_t r;
                if (from_input && to_output) r = launch(kin, kout, vin, vout);
                else if (from_input) r = launch(kin, ktmp, vin, vtmp);
                else if (to_output) r = launch(ktmp, kout, vtmp, vout);
                else r = launch(kout, ktmp, vout, vtmp);
                if (r != hipSuccess) return r;
                from_input = false;
                to_output = !to_output;
            }
            return hipSuccess;
        },
        variant);
}

template <class K, class V>
hipError_t onesweep_pairs(void *tmp, size_t &bytes, const K *kin, K *kout, const V *vin, V *vout, unsigned int n, unsigned int begin_bit,
                          unsigned int end_bit, hipStream_t st) {
#ifdef LFR_SORT_ONESWEEP_CONFIG
    return onesweep_pairs<LFR_SORT_ONESWEEP_CONFIG>(tmp, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, st);      // (sort_probe.hip: shapes for A/B)
#else
    static const int forced = [] { const char *e = std::getenv("LFR_SORT_TILES"); return e ? std::atoi(e) : 0; }();     // 1 = rocPRIM's default, 2 = 8-bit digits only
    if (forced == 1) return onesweep_pairs<typename LfrRadixSortConfig::onesweep_config>(tmp, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, st);
    switch (forced == 2 ? 8u : onesweep_digit_bits(end_bit - begin_bit)) {
        case 10: return onesweep_pairs<OnesweepShape<10, 8>>(tmp, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, st);
        case 9: return onesweep_pairs<OnesweepShape<9, 8>>(tmp, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, st);
        default: break;
    }
    if (n < kOnesweepShortBelow) return onesweep_pairs<OnesweepShape<8, 4>>(tmp, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, st);
    return onesweep_pairs<OnesweepShape<8, 8>>(tmp, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, st);
#endif
}
#endif   // LFR_SORT_OWN_DRIVER
}   // namespace sortdetail

// rocprim::radix_sort_pairs' two-call protocol (tmp == nullptr: size query, valid for either driver), ascending, stable, result in
// kout / vout.
template <class K, class V>
hipError_t sort_pairs_raw(void *tmp, size_t &bytes, const K *kin, K *kout, const V *vin, V *vout, int64_t n, int begin_bit, int end_bit, hipStream_t st) {
    // (the library's road below 256 K items is a block sort and log2(n / 4096) merge launches of ~6 us whatever the key width: a short key -
    // two digit places - over tens of thousands of items is five launches through the one-sweep driver: the 147 k batch-order keys of
    // config 4, 18 bits, 66 -> ~25 us)
    const bool short_key = n >= kOnesweepSmallFrom && end_bit - begin_bit <= 20;
    const bool big = sortdetail::single_fill_enabled() && (n > (int64_t)kSortMergeLimit || short_key) && n < ((int64_t)1 << 30);
    if (!tmp) {
        size_t lib = 0, own = 0;
        hipError_t e = rocprim::radix_sort_pairs<LfrRadixSortConfig>(nullptr, lib, kin, kout, vin, vout, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit, st);
        if (e != hipSuccess) return e;
#if LFR_SORT_OWN_DRIVER
        if (big && (e = sortdetail::onesweep_pairs(nullptr, own, kin, kout, vin, vout, (unsigned int)n, (unsigned)begin_bit, (unsigned)end_bit, st)) != hipSuccess) return e;
#endif
        bytes = lib > own ? lib : own;
        return hipSuccess;
    }
#if LFR_SORT_OWN_DRIVER
    if (big && !sortdetail::aliasing(kin, kout, vin, vout, n))
        return sortdetail::onesweep_pairs(tmp, bytes, kin, kout, vin, vout, (unsigned int)n, (unsigned)begin_bit, (unsigned)end_bit, st);
#endif
    return rocprim::radix_sort_pairs<LfrRadixSortConfig>(tmp, bytes, kin, kout, vin, vout, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit, st);
}

// ---- exclusive prefix sum in ONE launch ----
// rocprim::exclusive_scan is two launches (init_lookback_scan_state_kernel + the scan): seven scans in the config-4 pipeline, 33 us of
// init kernels.  The pipeline stages clear one zero block at their start anyway; a scan whose look-back states lie in that block needs no
// init launch.  Single pass, decoupled look-back: a block takes a ticket (blocks are numbered in the order they start, so every
// predecessor is running or done - no deadlock whatever the dispatch order), scans its tile of 2048 items, publishes PARTIAL | aggregate,
// wave 0 walks back 64 predecessors at a time to the nearest COMPLETE, publishes COMPLETE | inclusive prefix.  State word: flag in the top
// two bits, value below (sums of 64-bit items must stay under 2^62: they are counts of doubles / bytes here).
constexpr int kScanThreads = 256, kScanItems = 8, kScanTile = kScanThreads * kScanItems;
inline size_t scan_state_words(int64_t n) { return 1 + (size_t)((n + kScanTile - 1) / kScanTile); }       // ticket + one per tile; must be ZERO at launch

template <class T>
__global__ __launch_bounds__(kScanThreads) void k_exclusive_sum(const T *__restrict__ in, T *__restrict__ out, int64_t n, unsigned long long *state) {
    constexpr unsigned long long kPartial = 1ull << 62, kComplete = 2ull << 62, kFlags = 3ull << 62;
    constexpr unsigned long long kValue = sizeof(T) == 4 ? 0xffffffffull : ~kFlags;
    __shared__ unsigned long long s_wave[kScanThreads / 64];
    __shared__ unsigned long long s_prefix;
    __shared__ unsigned int s_bid;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_bid = atomicAdd(reinterpret_cast<unsigned int *>(state), 1u);
    __syncthreads();
    const unsigned int bid = s_bid;
    const int64_t base = (int64_t)bid * kScanTile + (int64_t)tid * kScanItems;
    T v[kScanItems];
    if (base + kScanItems <= n) {
        constexpr int per = 16 / sizeof(T);
        for (int q = 0; q < kScanItems / per; ++q) {
            const uint4 w = reinterpret_cast<const uint4 *>(in + base)[q];
            if constexpr (sizeof(T) == 4) { v[4 * q] = w.x; v[4 * q + 1] = w.y; v[4 * q + 2] = w.z; v[4 * q + 3] = w.w; }
            else { v[2 * q] = (T)w.x | ((T)w.y << 32); v[2 * q + 1] = (T)w.z | ((T)w.w << 32); }
        }
    } else {
        for (int k = 0; k < kScanItems; ++k) v[k] = base + k < n ? in[base + k] : (T)0;
    }
    unsigned long long tsum = 0;
    for (int k = 0; k < kScanItems; ++k) tsum += v[k];
    unsigned long long incl = tsum;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned long long wave_off = 0, agg = 0;
    for (int w = 0; w < kScanThreads / 64; ++w) { if (w < wave) wave_off += s_wave[w]; agg += s_wave[w]; }
    unsigned long long excl = wave_off + incl - tsum;
    if (wave == 0) {
        unsigned long long *bs = state + 1;
        if (lane == 0) __hip_atomic_store(bs + bid, (bid == 0 ? kComplete : kPartial) | (agg & kValue), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long prefix = 0;
        if (bid > 0) {
            for (int64_t j = (int64_t)bid - 1;; j -= 64) {
                const int64_t idx = j - lane;
                unsigned long long s, complete, upto;
                for (;;) {
                    s = idx >= 0 ? __hip_atomic_load(bs + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kComplete;      // before the first tile: nothing
                    complete = __ballot((s & kFlags) == kComplete);
                    upto = complete ? (((complete & (0ull - complete)) << 1) - 1ull) : ~0ull;          // lanes up to the nearest COMPLETE
                    if ((__ballot((s & kFlags) == 0ull) & upto) == 0ull) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                unsigned long long val = ((1ull << lane) & upto) ? (s & kValue) : 0ull;
                for (int d = 32; d >= 1; d >>= 1) val += __shfl_xor(val, d);
                prefix += val;
                if (complete) break;
            }
            if (lane == 0) __hip_atomic_store(bs + bid, kComplete | ((prefix + agg) & kValue), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s_prefix = prefix;
    }
    __syncthreads();
    excl += s_prefix;
    if (base + kScanItems <= n) {
        T o[kScanItems];
        for (int k = 0; k < kScanItems; ++k) { o[k] = (T)excl; excl += v[k]; }
        constexpr int per = 16 / sizeof(T);
        for (int q = 0; q < kScanItems / per; ++q) {
            uint4 w;
            if constexpr (sizeof(T) == 4) w = make_uint4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
            else w = make_uint4((unsigned int)o[2 * q], (unsigned int)(o[2 * q] >> 32), (unsigned int)o[2 * q + 1], (unsigned int)(o[2 * q + 1] >> 32));
            reinterpret_cast<uint4 *>(out + base)[q] = w;
        }
    } else {
        for (int k = 0; k < kScanItems && base + k < n; ++k) { out[base + k] = (T)excl; excl += v[k]; }
    }
}
// `state`: scan_state_words(n) zeroed 64-bit words (in the stage's zero block), used by this one call only
template <class T>
inline hipError_t exclusive_sum_one_launch(const T *in, T *out, int64_t n, unsigned long long *state, hipStream_t st) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "32- or 64-bit unsigned items");
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_exclusive_sum<T>, dim3((unsigned)((n + kScanTile - 1) / kScanTile)), dim3(kScanThreads), 0, st, in, out, n, state);
    return hipGetLastError();
}

}   // namespace lfr

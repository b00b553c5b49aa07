// Stable key/value sort for the device pipeline (graph stage, batch assembly, hand-out order).
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
#include <rocprim/rocprim.hpp>

#include <cstdint>
#include <cstdlib>
#include <variant>

namespace lfr {

using LfrRadixSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, (size_t)256 * 1024>;
constexpr size_t kSortMergeLimit = (size_t)256 * 1024;

namespace sortdetail {
namespace rd = ::rocprim::detail;

inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// bytes == 0 on entry with tmp == nullptr: size query.  Keys/values must not alias their outputs.
template <class K, class V>
hipError_t onesweep_pairs(void *tmp, size_t &bytes, const K *kin, K *kout, const V *vin, V *vout, unsigned int n, unsigned int begin_bit,
                          unsigned int end_bit, hipStream_t st) {
    using offset_type = unsigned int;
    using sort_config = typename LfrRadixSortConfig::onesweep_config;
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

inline bool single_fill_enabled() {
    static const bool on = [] { const char *s = std::getenv("LFR_SORT_ROCPRIM"); return !(s && s[0] == '1'); }();
    return on;
}
template <class K, class V>
inline bool aliasing(const K *kin, const K *kout, const V *vin, const V *vout, int64_t n) {
    auto overlap = [n](const void *a, size_t sa, const void *b, size_t sb) {
        const char *x = static_cast<const char *>(a), *y = static_cast<const char *>(b);
        return x < y + sb * (size_t)n && y < x + sa * (size_t)n;
    };
    return overlap(kin, sizeof(K), kout, sizeof(K)) || overlap(vin, sizeof(V), vout, sizeof(V));
}
}   // namespace sortdetail

// rocprim::radix_sort_pairs' two-call protocol (tmp == nullptr: size query, valid for either driver), ascending, stable, result in
// kout / vout.
template <class K, class V>
hipError_t sort_pairs_raw(void *tmp, size_t &bytes, const K *kin, K *kout, const V *vin, V *vout, int64_t n, int begin_bit, int end_bit, hipStream_t st) {
    const bool big = sortdetail::single_fill_enabled() && n > (int64_t)kSortMergeLimit && n < ((int64_t)1 << 30);
    if (!tmp) {
        size_t lib = 0, own = 0;
        hipError_t e = rocprim::radix_sort_pairs<LfrRadixSortConfig>(nullptr, lib, kin, kout, vin, vout, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit, st);
        if (e != hipSuccess) return e;
        if (big && (e = sortdetail::onesweep_pairs(nullptr, own, kin, kout, vin, vout, (unsigned int)n, (unsigned)begin_bit, (unsigned)end_bit, st)) != hipSuccess) return e;
        bytes = lib > own ? lib : own;
        return hipSuccess;
    }
    if (big && !sortdetail::aliasing(kin, kout, vin, vout, n))
        return sortdetail::onesweep_pairs(tmp, bytes, kin, kout, vin, vout, (unsigned int)n, (unsigned)begin_bit, (unsigned)end_bit, st);
    return rocprim::radix_sort_pairs<LfrRadixSortConfig>(tmp, bytes, kin, kout, vin, vout, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit, st);
}

}   // namespace lfr

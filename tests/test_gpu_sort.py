"""The pipeline's device sort (lfr_sort.hpp): rocPRIM's one-sweep radix kernels under a driver that clears the look-back states and block
counters of ALL digit places with one fill.  The reference sorts on the host (std::sort of the matches by similarity, solve.cc:489-497;
components visited in order, solve.cc:563-597); the yardsticks here are numpy's stable argsort and rocprim::radix_sort_pairs itself.
Bit-exact: a stable sort has one answer."""
import numpy as np
import pytest

from lfr_amd import capi

pytestmark = pytest.mark.gpu

MERGE_LIMIT = 256 * 1024


def expected(keys, vals, begin_bit, end_bit):
    width = end_bit - begin_bit
    mask = np.uint64(0xFFFFFFFFFFFFFFFF) if width == 64 else (np.uint64(1) << np.uint64(width)) - np.uint64(1)
    digits = (keys.astype(np.uint64) >> np.uint64(begin_bit)) & mask
    order = np.argsort(digits, kind="stable")
    return keys[order], vals[order]


@pytest.mark.parametrize("dtype,begin_bit,end_bit", [(np.uint32, 0, 20), (np.uint32, 0, 32), (np.uint32, 0, 3), (np.uint64, 0, 44),
                                                     (np.uint64, 0, 52), (np.uint64, 0, 64), (np.uint64, 5, 38)])
@pytest.mark.parametrize("n", [MERGE_LIMIT + 1, 1_000_003, 5_242_880])
def test_sort_is_the_stable_sort(lfr_lib, dtype, begin_bit, end_bit, n):
    rng = np.random.default_rng(n % 1000 + end_bit)
    bits = 8 * np.dtype(dtype).itemsize
    keys = rng.integers(0, 2 ** 63, n, dtype=np.uint64).astype(dtype) if bits == 64 else rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(dtype)
    if end_bit - begin_bit > 16:
        keys[: n // 2] = keys[n // 2: 2 * (n // 2)]            # every key twice: the order of equal keys is visible in the values
    vals = np.arange(n, dtype=np.uint32)
    ek, ev = expected(keys, vals, begin_bit, end_bit)
    k, v = capi.sort_pairs_hip(keys, vals, begin_bit, end_bit)
    assert (k == ek).all() and (v == ev).all()
    kl, vl = capi.sort_pairs_hip(keys, vals, begin_bit, end_bit, use_library=True)
    assert (kl == ek).all() and (vl == ev).all()


@pytest.mark.parametrize("dtype", [np.uint32, np.uint64])
@pytest.mark.parametrize("n,bits", [(32 * 1024, 18), (147_000, 18), (147_000, 20), (MERGE_LIMIT, 9), (32 * 1024 - 1, 18), (147_000, 21)])
def test_short_keys_of_small_sorts(lfr_lib, dtype, n, bits):
    """Round 6: from 32 K items on, keys of at most 20 bits go through the one-sweep driver below the merge-sort limit too (the batch-order
    keys of the assembly: 147 k components, 18 bits - five launches instead of a block sort and eight merges); 32 K - 1 items and 21 bits
    stay with the library.  Every key many times: stability is visible in the values."""
    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, 2 ** min(bits, 12), n, dtype=np.uint64).astype(dtype) << dtype(max(bits - 12, 0))
    keys |= rng.integers(0, 2, n, dtype=np.uint64).astype(dtype)
    vals = np.arange(n, dtype=np.uint32)
    ek, ev = expected(keys, vals, 0, bits)
    k, v = capi.sort_pairs_hip(keys, vals, 0, bits)
    assert (k == ek).all() and (v == ev).all()


@pytest.mark.parametrize("n", [1, 2, 1000, MERGE_LIMIT - 1, MERGE_LIMIT])
def test_small_sorts_take_the_library_road(lfr_lib, n):
    """At and below the merge-sort limit the call is rocprim::radix_sort_pairs as before (keys of more than 20 bits, or fewer than 32 K items); same answer."""
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 2 ** 24, n, dtype=np.uint64).astype(np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    ek, ev = expected(keys, vals, 0, 24)
    k, v = capi.sort_pairs_hip(keys, vals, 0, 24)
    assert (k == ek).all() and (v == ev).all()


def test_skewed_and_constant_keys(lfr_lib):
    """One digit holding (almost) every key: the look-back chain of that digit runs through every block."""
    n = 3_000_000
    vals = np.arange(n, dtype=np.uint32)
    keys = np.full(n, 0x00ABCDEF12345678, np.uint64)
    k, v = capi.sort_pairs_hip(keys, vals, 0, 56)
    assert (k == keys).all() and (v == vals).all()
    keys[::1000] = 7
    ek, ev = expected(keys, vals, 0, 56)
    k, v = capi.sort_pairs_hip(keys, vals, 0, 56)
    assert (k == ek).all() and (v == ev).all()


def test_bad_arguments(lfr_lib):
    k = np.zeros(4, np.uint32)
    with pytest.raises(capi.LfrError):
        capi.sort_pairs_hip(k, k, 0, 40)
    with pytest.raises(capi.LfrError):
        capi.sort_pairs_hip(k, k, 8, 8)


@pytest.mark.parametrize("dtype", [np.uint32, np.uint64])
@pytest.mark.parametrize("n", [1, 7, 2047, 2048, 2049, 150_001, 2_500_001, 20_000_000])
def test_one_launch_exclusive_sum(lfr_lib, dtype, n):
    """The stages' prefix sums (flags -> ids, sizes -> offsets): one launch, look-back states in the stage's zero block.  Exact integers."""
    rng = np.random.default_rng(n)
    hi = 2 if n > 10_000_000 else (200 if dtype == np.uint32 else 2 ** 40)
    v = rng.integers(0, hi, n, dtype=np.uint64).astype(dtype)
    got = capi.exclusive_sum_hip(v)
    want = np.zeros(n, np.uint64)                     # (uint64 throughout: a Python 0 in the concatenation would promote to float64)
    want[1:] = np.cumsum(v.astype(np.uint64))[:-1]
    assert (got == want.astype(dtype)).all()


def test_exclusive_sum_of_32_bit_items_wraps(lfr_lib):
    v = np.full(5000, 0xF0000000, np.uint32)
    got = capi.exclusive_sum_hip(v)
    want = (np.arange(5000, dtype=np.uint64) * np.uint64(0xF0000000)).astype(np.uint32)
    assert (got == want).all()

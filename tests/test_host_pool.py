"""The library's persistent host workers (lfr::run_on_pool: they make the elimination-tree plans at batch creation) through the C ABI's
self test: every item is done exactly once whatever the thread count, call after call, and several callers at once are served (one by the
workers, the others by threads of their own)."""
import threading

from lfr_amd import capi


def test_every_item_once(lfr_lib):
    L = capi.lib()
    for threads in (1, 2, 7, 64, 128):
        assert L.lfr_debug_pool_selftest(threads, 10_000, 5) == 50_000
    assert L.lfr_debug_pool_selftest(16, 0, 3) == 0
    assert L.lfr_debug_pool_selftest(16, 3, 200) == 600              # fewer items than threads, many wake-ups


def test_concurrent_callers(lfr_lib):
    L = capi.lib()
    out = {}

    def run(i):
        out[i] = L.lfr_debug_pool_selftest(8 + i, 5_000, 40)       # (ctypes releases the GIL: the calls overlap)
    th = [threading.Thread(target=run, args=(i,)) for i in range(6)]
    for t in th: t.start()
    for t in th: t.join()
    assert out == {i: 200_000 for i in range(6)}

"""factor_lds's barrier-free schedule (wave 0 ahead on the diagonal tiles, wave 1's `lead` / `lead_t` words, the workers' own counter barrier),
modelled at tile level on the CPU (scripts/factor_lds_sync_model.py): random interleavings with a vector-clock race detector, a
deadlock detector and a completeness check (every tile receives every panel's update and its substitution exactly once).  The model must also
CATCH the protocol with any one of its waits removed."""
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import factor_lds_sync_model as M                 # noqa: E402


# rows of the systems solve_block_kernel factors: 2 * n_var_nodes - up to 192 (one tile, partial last tiles, right-hand side in a tile of its own)
ROWS = [2, 16, 17, 31, 32, 33, 48, 64, 80, 88, 96, 112, 128, 129, 130, 144, 160, 176, 191, 192]


@pytest.mark.parametrize("waves", [2, 4, 8])       # the 128 / 256 / 512-thread instantiations
def test_every_access_is_ordered_by_the_four_words(waves):
    rng = random.Random(waves)
    for n in ROWS:
        for _ in range(6):
            M.run(n, waves, rng)


@pytest.mark.parametrize("broken", ["drop_lead_wait", "drop_ready_wait", "drop_lead_t_wait"])
@pytest.mark.parametrize("waves", [2, 4, 8])
def test_the_model_catches_a_removed_wait(waves, broken):
    rng = random.Random(11)
    with pytest.raises(AssertionError, match="race"):
        M.run(192, waves, rng, **{broken: True})


@pytest.mark.parametrize("waves", [4, 8])
def test_the_model_catches_a_removed_worker_barrier(waves):
    with pytest.raises(AssertionError, match="race"):
        M.run(192, waves, random.Random(5), drop_worker_barrier=True)


def test_dealing_covers_every_tile_once_per_panel():
    """the schedule as a work list: per panel k, the tiles written by the waves are exactly the trailing lower triangle + wave 0's diagonal tile"""
    for waves in (2, 4, 8):
        for n in (33, 96, 130, 192):
            prog, P, RT = M.programs(n, waves)
            writes = {}
            for w, p in enumerate(prog):
                for st in p:
                    if st[0] == "w" and st[1][0] == "t": writes.setdefault(st[1][1:], []).append(w)
            for J in range(P):
                for R in range(J, RT):
                    assert len(writes[(R, J)]) == J + 1, (n, waves, R, J)
            assert all(J < P and J <= R < RT for (R, J) in writes)

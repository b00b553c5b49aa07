"""The elimination-tree kernel's barrier-free synchronisation, modelled on the CPU (scripts/tree_sync_model.py): the state-word
protocol of the streaming left-looking factorization and of the back substitution on the REAL plans of cap-sized components, and the teams'
granule reductions / arrival-counter barriers - random interleavings with a vector-clock race detector and a deadlock detector.  The model
must also CATCH deliberately broken protocols."""
import os
import random
import sys

import numpy as np
import pytest

from lfr_amd import capi, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import tree_sync_model as M                       # noqa: E402
from tree_plan_stats import component_words       # noqa: E402


@pytest.fixture(scope="module")
def plans(lfr_lib):
    ma = synthetic.capsized_sparse(n_images=300, n_tracks=3000, seed=17)
    p = capi.Problem(capi.Graph.from_arrays(ma))
    out = []
    for nv, w in sorted(component_words(ma, p), key=lambda c: -c[0])[:3]:
        blob, info = capi.tree_plan(nv, w)
        out.append(M.Plan(blob))
    assert out and out[0].NB >= 20
    return out


@pytest.mark.parametrize("waves", [8, 16, 32, 64])
def test_state_words_order_every_read_after_its_write(plans, waves):
    rng = random.Random(waves)
    for plan in plans:
        # every column is taken by exactly one wave, whatever the rotation by level
        taken = sorted(q for g in range(waves) for q in plan.columns_of_wave(g, waves))
        assert taken == list(range(plan.NB))
        for _ in range(2):
            M.run_factor_and_solve(plan, waves, rng)
            M.run_factor_and_solve(plan, waves, rng, streaming=False)


def test_the_model_catches_a_forgotten_gate(plans):
    rng = random.Random(5)
    caught = 0
    for _ in range(20):
        try:
            M.run_factor_and_solve(plans[0], 8, rng, forget_gate=True)
        except AssertionError as e:
            assert "race" in str(e)
            caught += 1
    assert caught >= 15            # (a lucky interleaving may still order the read)


@pytest.mark.parametrize("members", [2, 4, 8])
def test_team_reductions_and_barriers(members):
    rng = random.Random(members)
    for _ in range(30):
        M.run_team_reductions(members, 40, rng)
        M.run_team_barriers(members, 40, rng)
    with pytest.raises(AssertionError):      # one set of slots instead of two: a fast member overwrites what a slow one has not read
        for _ in range(200):
            M.run_team_reductions(members, 40, rng, parities=1)


WANTS = [8] * 3 + [4] * 6 + [2] * 10 + [1] * 12


@pytest.mark.parametrize("capacity", [[None] * 4, [8, 8, 8, 8], [7, 7, 7, 7], [1, 1, 1, 1], [8, 3, 1, 5], [16, 1, 2, 9]])
def test_team_formation_under_any_residency(capacity):
    """Registration, unit states, queue, permanent splits and SOLO mode of solve_tree_team_kernel: 64 workgroups on 4
    XCDs, units of 8, with the CUs of each XCD limited to `capacity`.  Every component is solved exactly once - by a team of the size it
    asks for, or alone once no such team can form -, the members of a unit agree on its state, every workgroup exits."""
    rng = random.Random(len(capacity) + sum(c or 0 for c in capacity))
    for _ in range(25):
        off_size, _ = M.run_team_formation(4, 8, 64, capacity, WANTS, rng)
        can_form = all(c is None or c >= 8 for c in capacity) or any(c is not None and c >= 8 for c in capacity)
        assert (off_size == 0) == can_form, (capacity, off_size)


def test_the_model_catches_units_without_an_agreed_state_and_overtaking_messages():
    """Two deliberately broken protocols.  (1) Completeness read off the registration count (round 5) plus a solo flag: the members of a
    unit decide differently and a leader talks to a workgroup that has left.  (2) Mailboxes filled in ascending rank order (rounds 5-6
    until this model): the leader of a sub-team is through the queue before the old leader has reached its members, its message
    overtakes, and a member joins a component its team is not working on."""
    caught = 0
    for k in range(30):
        try:
            M.run_team_formation(4, 8, 64, [7, 7, 7, 9], WANTS, random.Random(k), consensus=False, max_steps=200000)
        except AssertionError:
            caught += 1
    assert caught >= 25
    caught = 0
    for k in range(30):
        try:
            M.run_team_formation(4, 8, 64, [None] * 4, WANTS, random.Random(k), descending_sends=False, max_steps=200000)
        except AssertionError:
            caught += 1
    assert caught >= 20

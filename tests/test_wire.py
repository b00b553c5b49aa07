"""Wire format (types.proto) — the Python codec and the native scanner/emitter against
google.protobuf built from a runtime descriptor, plus the file-handling rules of
solve.cc:416-451 (part files, banned images, first-wins facts)."""
import os

import numpy as np
import pytest

import pb_runtime
from lfr_amd import capi, synthetic, wire


@pytest.fixture(scope="module")
def pb():
    return pb_runtime.build()


def small_graph(seed=21, **kw):
    args = dict(seed=seed, n_images=12, n_tracks=25, eps_out=0.0)
    args.update(kw)
    return synthetic.generate(**args)


def test_python_codec_matches_google_protobuf(pb):
    MatchingFile, _ = pb
    ma = small_graph()
    pairs = ma.to_pairs()
    pairs[0]["fact1"] = 0.0                      # proto3: zero scalars omitted
    pairs[1]["matches"][0]["feature_idx1"] = 0
    pairs[1]["matches"][0]["disp1"][0] = (0.0, -0.0)
    mine = wire.encode_matching_file(pairs)
    theirs = pb_runtime.pairs_to_pb(MatchingFile, pairs).SerializeToString()
    assert mine == theirs
    assert wire.decode_matching_file(theirs) == pairs
    parsed = MatchingFile()
    parsed.ParseFromString(mine)
    assert len(parsed.image_pairs) == len(pairs)


def test_solution_codec_matches_google_protobuf(pb):
    _, SolutionFile = pb
    images = [{"image_name": "a.png", "fact": 1.0, "displacements": [(0, 0.0, 0.0), (7, 0.25, -0.125)]},
              {"image_name": "b.png", "fact": 0.5, "displacements": [(3, -1.0, 1.0)]}]
    mine = wire.encode_solution_file(images)
    msg = SolutionFile()
    for im in images:
        x = msg.images.add()
        x.image_name, x.fact = im["image_name"], im["fact"]
        for (f, di, dj) in im["displacements"]:
            d = x.displacements.add()
            d.feature_idx, d.di, d.dj = f, di, dj
    assert mine == msg.SerializeToString()
    assert wire.decode_solution_file(mine) == images


def test_native_writer_and_scanner(lfr_lib, pb, tmp_path):
    MatchingFile, _ = pb
    ma = small_graph(seed=22, eps_out=0.02, n_images=40)
    path = str(tmp_path / "m.pb")
    capi.write_matching_file(path, ma)
    data = open(path, "rb").read()
    assert data == wire.encode_matching_file(ma.to_pairs())
    assert data == pb_runtime.pairs_to_pb(MatchingFile, ma.to_pairs()).SerializeToString()
    g = capi.Graph.from_matches_file(path)
    g2 = capi.Graph.from_arrays(ma)
    assert g.n_nodes == g2.n_nodes and g.n_edges == g2.n_edges == 2 * ma.n_matches
    assert (g.nodes()[0] == g2.nodes()[0]).all() and (g.nodes()[1] == g2.nodes()[1]).all()
    assert g.image_names() == g2.image_names()


def test_part_files_and_banned_images(lfr_lib, tmp_path):
    ma = small_graph(seed=23)
    pairs = ma.to_pairs()
    base = str(tmp_path / "matches.pb")
    half = len(pairs) // 2
    open(base + ".part.0", "wb").write(wire.encode_matching_file(pairs[:half]))
    open(base + ".part.1", "wb").write(wire.encode_matching_file(pairs[half:]))
    open(base + ".part.3", "wb").write(wire.encode_matching_file(pairs))      # after a gap: ignored
    g = capi.Graph.from_matches_file(base)                                      # solve.cc:416-424
    whole = capi.Graph.from_arrays(ma)
    assert g.n_nodes == whole.n_nodes and g.n_edges == whole.n_edges
    banned = [ma.image_names[0], ma.image_names[3]]
    gb = capi.Graph.from_matches_file(base, banned)                             # solve.cc:444-446
    kept = [p for p in pairs if p["image_name1"] not in banned and p["image_name2"] not in banned]
    assert gb.n_edges == 2 * sum(len(p["matches"]) for p in kept)
    assert not set(banned) & set(gb.image_names())
    import lfr_ref
    ref = lfr_ref.MatchGraph(pairs, banned)
    img, feat = gb.nodes()
    names = gb.image_names()
    assert [(names[i], int(f)) for i, f in zip(img, feat)] == ref.node_key


def test_first_fact_wins_and_short_grids(lfr_lib, tmp_path):
    pairs = [{"image_name1": "a", "fact1": 2.0, "image_name2": "b", "fact2": 1.0, "matches": [
                 {"feature_idx1": 1, "feature_idx2": 2, "similarity": 0.9, "disp1": [(0.1, 0.2)] * 4, "disp2": []}]},
             {"image_name1": "b", "fact1": 7.0, "image_name2": "a", "fact2": 9.0, "matches": []},
             {"image_name1": "c", "fact1": 3.0, "image_name2": "a", "fact2": 9.0, "matches": []}]
    path = str(tmp_path / "m.pb")
    open(path, "wb").write(wire.encode_matching_file(pairs))
    g = capi.Graph.from_matches_file(path)
    assert g.image_names() == ["a", "b", "c"]          # images of empty pairs are 'seen' (solve.cc:448-451)
    assert g.image_facts() == [2.0, 1.0, 3.0]          # std::map::insert keeps the first (solve.cc:449,451)
    assert g.n_nodes == 2 and g.n_edges == 2


@pytest.mark.parametrize("damage", ["truncate", "bad_varint", "group"])
def test_malformed_input_is_a_parse_error(lfr_lib, tmp_path, damage):
    data = bytearray(wire.encode_matching_file(small_graph(seed=24).to_pairs()))
    if damage == "truncate":
        data = data[:len(data) // 2 + 1]
    elif damage == "bad_varint":
        data = bytearray(b"\x0a" + b"\xff" * 11)
    else:
        data = bytearray(b"\x0b\x0c")                  # start-group / end-group
    path = str(tmp_path / "bad.pb")
    open(path, "wb").write(bytes(data))
    with pytest.raises(capi.LfrError) as e:
        capi.Graph.from_matches_file(path)
    assert e.value.code == -3 and "Failed to parse proto object." in str(e.value)


def test_more_than_nine_grid_points_rejected(lfr_lib, tmp_path):
    pairs = [{"image_name1": "a", "fact1": 1.0, "image_name2": "b", "fact2": 1.0, "matches": [
        {"feature_idx1": 1, "feature_idx2": 2, "similarity": 0.9, "disp1": [(0.1, 0.2)] * 10, "disp2": []}]}]
    path = str(tmp_path / "m.pb")
    open(path, "wb").write(wire.encode_matching_file(pairs))
    with pytest.raises(capi.LfrError) as e:
        capi.Graph.from_matches_file(path)
    assert e.value.code == -5


def test_solution_emit(lfr_lib, pb, tmp_path):
    _, SolutionFile = pb
    import lfr_ref
    ma = small_graph(seed=25)
    g = capi.Graph.from_arrays(ma)
    rng = np.random.default_rng(0)
    pos = rng.uniform(-0.9, 0.9, size=(g.n_nodes, 2))
    pos[::3] = 0.0
    out = str(tmp_path / "s.pb")
    n_out = g.write_solution(pos, out)
    assert n_out == int((np.abs(pos) > 0.5).any(axis=1).sum())                  # solve.cc:666-670
    ref = lfr_ref.MatchGraph(ma.to_pairs())
    expect = lfr_ref.solution_images({"node_key": ref.node_key, "positions": pos, "images_facts": ref.images_facts})
    data = open(out, "rb").read()
    assert wire.decode_solution_file(data) == expect
    assert data == wire.encode_solution_file(expect)
    msg = SolutionFile()
    msg.ParseFromString(data)
    assert [im.image_name for im in msg.images] == [e["image_name"] for e in expect]


def test_apply_displacements_matches_the_consumer_arithmetic(lfr_lib):
    """colmap_utils.py:126-137 restated with numpy float32 ops vs lfr_apply_displacements."""
    ma = synthetic.generate(seed=26, n_images=10, n_tracks=60, fact=1.0)
    ma.facts[:] = np.float32([1.0, 0.5, 2.0, 1.0, 0.25, 1.0, 1.5, 1.0, 1.0, 3.0])
    g = capi.Graph.from_arrays(ma)
    rng = np.random.default_rng(3)
    pos = rng.uniform(-1, 1, size=(g.n_nodes, 2))
    img, feat = g.nodes()
    names = g.image_names()
    facts = g.image_facts()
    for im, name in enumerate(names + ["unknown.png"]):
        nf = 80
        kp = rng.uniform(0, 1000, size=(nf, 4)).astype(np.float32)
        want = kp.copy()
        disp = np.zeros([nf, 2]).astype(np.float32)
        if im < len(names):
            for n in np.nonzero(img == im)[0]:
                disp[feat[n], :] = [np.float32(pos[n, 1]), np.float32(pos[n, 0])]      # [dj, di]
            disp *= facts[im]
        want[:, :2] += disp * 16
        want[:, :2] += 0.5
        got = g.apply_displacements(pos, name, kp)
        assert (got == want).all()
    with pytest.raises(capi.LfrError):
        g.apply_displacements(pos, names[0], np.zeros((1, 2), np.float32))       # feature_idx out of range

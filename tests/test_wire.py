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


@pytest.mark.parametrize("damage", ["truncate", "bad_varint", "stray_end_group", "mismatched_end_group", "open_group"])
def test_malformed_input_is_a_parse_error(lfr_lib, tmp_path, damage):
    data = bytearray(wire.encode_matching_file(small_graph(seed=24).to_pairs()))
    if damage == "truncate":
        data = data[:len(data) // 2 + 1]
    elif damage == "bad_varint":
        data = bytearray(b"\x0a" + b"\xff" * 11)
    elif damage == "stray_end_group":
        data = bytearray(b"\x0c")
    elif damage == "mismatched_end_group":
        data = bytearray(b"\x0b\x14")                  # start-group of field 1, end-group of field 2
    else:
        data = bytearray(b"\x0b")
    path = str(tmp_path / "bad.pb")
    open(path, "wb").write(bytes(data))
    with pytest.raises(capi.LfrError) as e:
        capi.Graph.from_matches_file(path)
    assert e.value.code == -3 and "Failed to parse proto object." in str(e.value)


def test_groups_are_skipped_like_any_unknown_field(lfr_lib, pb, tmp_path):
    """protobuf skips unknown fields, groups (wire types 3/4) included: so does the reference's generated parser (solve.cc:427-436)."""
    MatchingFile, _ = pb
    body = wire.encode_matching_file(small_graph(seed=25).to_pairs())
    ref = capi.Graph.from_arrays(small_graph(seed=25))
    for extra in (b"\x0b\x0c", b"\x0b\x08\x01\x0c", b"\x13\x1b\x0d\x00\x00\x80\x3f\x1c\x14"):     # empty group, group with a varint, nested groups
        for data in (extra + body, body + extra):
            m = MatchingFile()
            m.ParseFromString(data)                                               # google.protobuf accepts it
            path = str(tmp_path / "g.pb")
            open(path, "wb").write(data)
            g = capi.Graph.from_matches_file(path)
            assert (g.n_nodes, g.n_edges) == (ref.n_nodes, ref.n_edges) and g.n_edges == 2 * sum(len(p.matches) for p in m.image_pairs)


def test_more_than_nine_grid_points_in_a_banned_pair_are_not_an_error(lfr_lib, tmp_path):
    """The reference `continue`s past banned pairs before it reads their matches (solve.cc:444-446)."""
    long_match = {"feature_idx1": 1, "feature_idx2": 2, "similarity": 0.9, "disp1": [(0.1, 0.2)] * 10, "disp2": []}
    ok_match = {"feature_idx1": 3, "feature_idx2": 4, "similarity": 0.8, "disp1": [(0.1, 0.2)] * 9, "disp2": [(0.0, 0.1)] * 9}
    pairs = [{"image_name1": "a", "fact1": 1.0, "image_name2": "x", "fact2": 1.0, "matches": [long_match]},
             {"image_name1": "a", "fact1": 1.0, "image_name2": "b", "fact2": 1.0, "matches": [ok_match]}]
    for name, segment in (("one.pb", None), ("many.pb", "64")):
        path = str(tmp_path / name)
        open(path, "wb").write(wire.encode_matching_file(pairs))
        g = capi.Graph.from_matches_file(path, ["x"])
        assert g.image_names() == ["a", "b"] and g.n_edges == 2
        with pytest.raises(capi.LfrError) as e:
            capi.Graph.from_matches_file(path)
        assert e.value.code == -5


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


def _graph_signature(g):
    import numpy as np
    from lfr_amd import capi
    img, feat = g.nodes()
    p = capi.Problem(g, device_assembly=True)              # labels only (host stage): a fingerprint of endpoints + similarities
    t, r, c = p.labels()
    return (g.n_nodes, g.n_edges, g.image_names(), [float(x) for x in g.image_facts()], img.tolist(), feat.tolist(),
            t.tolist(), r.tolist(), c.tolist())


@pytest.mark.parametrize("n_parts", [1, 3])
@pytest.mark.parametrize("banned", [(), ("000003.png", "000011.png")])
@pytest.mark.parametrize("features", ["dense", "sparse"])
def test_parallel_scanner_equals_the_sequential_one(lfr_lib, tmp_path, monkeypatch, n_parts, banned, features):
    """the whole-input parallel scanner (first-appearance = minimum position: images, first facts,
    node ids) against the file-by-file scanner with the sequential numbering pass, on `.part.N` files (solve.cc:416-424),
    with banned images (solve.cc:444-446), differing facts for one image (first wins, solve.cc:449-451) and
    duplicated matches - same nodes, same labels, same solution bytes."""
    import numpy as np
    from lfr_amd import capi, synthetic
    ma = synthetic.generate(seed=31, n_images=16, n_tracks=400, eps_out=0.02)
    if features == "sparse":        # feature indices all over 32 bits: the node table is the hash map, not the per-image dense table
        ma.feat1 = ((ma.feat1.astype(np.uint64) * 1000003 + 17) % (1 << 32)).astype(np.uint32)
        ma.feat2 = ((ma.feat2.astype(np.uint64) * 1000003 + 17) % (1 << 32)).astype(np.uint32)
    pairs = ma.to_pairs()
    pairs[5]["fact1"] = 0.5                                   # a later, different fact for an image seen before: ignored
    pairs[7]["matches"] = pairs[7]["matches"] + pairs[7]["matches"][:2]      # duplicates are kept (solve.cc:476-478)
    base = tmp_path / "m.pb"
    cut = [len(pairs) * k // n_parts for k in range(n_parts + 1)]
    for k in range(n_parts):
        path = base if n_parts == 1 else tmp_path / ("m.pb.part.%d" % k)
        path.write_bytes(wire.encode_matching_file(pairs[cut[k]:cut[k + 1]]))
    monkeypatch.setenv("LFR_HOST_THREADS", "5")
    monkeypatch.setenv("LFR_SCANNER_SEGMENT_BYTES", "3000")        # several speculative segments per (small) file
    monkeypatch.delenv("LFR_SCANNER_SEQUENTIAL", raising=False)
    g_par = capi.Graph.from_matches_file(str(base), banned)
    monkeypatch.setenv("LFR_SCANNER_SEQUENTIAL", "1")
    g_seq = capi.Graph.from_matches_file(str(base), banned)
    assert _graph_signature(g_par) == _graph_signature(g_seq)
    pos = np.random.default_rng(1).uniform(-0.4, 0.4, (g_par.n_nodes, 2))
    g_par.write_solution(pos, str(tmp_path / "a.sol"))
    g_seq.write_solution(pos, str(tmp_path / "b.sol"))
    assert (tmp_path / "a.sol").read_bytes() == (tmp_path / "b.sol").read_bytes()
    for b in banned:
        assert b not in g_par.image_names()


def test_parallel_scanner_rejects_what_the_sequential_one_rejects(lfr_lib, tmp_path):
    from lfr_amd import capi
    bad = tmp_path / "bad.pb"
    bad.write_bytes(b"\x0a\xff\xff\xff\xff\x0f" + b"\x00" * 10)       # a length that runs past the end of the file
    with pytest.raises(capi.LfrError) as e:
        capi.Graph.from_matches_file(str(bad))
    assert e.value.code == -3
    empty = tmp_path / "empty.pb"
    empty.write_bytes(b"")
    g = capi.Graph.from_matches_file(str(empty))
    assert g.n_nodes == 0 and g.n_edges == 0


def test_speculative_split_survives_misleading_bytes(lfr_lib, tmp_path, monkeypatch):
    """The parallel top-level split guesses record starts inside each segment; image names and flow values that look like
    record headers (0x0A, small lengths) must at worst cost a sequential re-walk of that stretch, never a different graph."""
    import numpy as np
    from lfr_amd import capi, synthetic
    ma = synthetic.generate(seed=33, n_images=9, n_tracks=300, eps_out=0.05)
    pairs = ma.to_pairs()
    for k, p in enumerate(pairs):                               # names full of record-start look-alikes
        p["image_name1"] = "\n\x02\n\x00" + p["image_name1"] + "\n\x03\n\x01a"
        p["image_name2"] = "\n\x04\n\x02ab" + p["image_name2"]
    tricky = np.frombuffer(b"\n\x02\n\x00", dtype=np.float32)[0]       # a float32 whose bytes spell a tiny record
    for p in pairs[::3]:
        for m in p["matches"]:
            m["disp1"] = [(float(tricky), float(tricky))] * 9
    path = tmp_path / "t.pb"
    path.write_bytes(wire.encode_matching_file(pairs))
    sigs = []
    for seg in ("64", "257", "5000", "100000000"):
        monkeypatch.setenv("LFR_SCANNER_SEGMENT_BYTES", seg)
        monkeypatch.setenv("LFR_HOST_THREADS", "7")
        sigs.append(_graph_signature(capi.Graph.from_matches_file(str(path))))
    monkeypatch.setenv("LFR_SCANNER_SEQUENTIAL", "1")
    ref = _graph_signature(capi.Graph.from_matches_file(str(path)))
    for sg in sigs:
        assert sg == ref

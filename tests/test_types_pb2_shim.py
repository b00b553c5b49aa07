"""the `types_pb2`-compatible shim as a product artifact (lfr_amd/types_pb2.py), exercised the way
the reference's producer (compute_match_graph.py:163-205) and consumer (colmap_utils.py:104-137) use the generated module."""
import os
import subprocess
import sys

import numpy as np

from lfr_amd import capi, synthetic, wire
from lfr_amd import types_pb2


def test_message_classes_follow_types_proto():
    mf = types_pb2.MatchingFile.DESCRIPTOR
    assert [f.name for f in mf.fields] == ["image_pairs"]
    ip = mf.nested_types_by_name["ImagePair"]
    assert [(f.name, f.number) for f in ip.fields] == [("image_name1", 1), ("fact1", 2), ("image_name2", 3), ("fact2", 4), ("matches", 5)]
    m = ip.nested_types_by_name["Match"]
    assert [(f.name, f.number) for f in m.fields] == [("feature_idx1", 1), ("feature_idx2", 2), ("similarity", 3), ("disp1", 4), ("disp2", 5)]
    sf = types_pb2.SolutionFile.DESCRIPTOR
    im = sf.nested_types_by_name["Image"]
    assert [(f.name, f.number) for f in im.fields] == [("image_name", 1), ("fact", 2), ("displacements", 3)]
    d = im.nested_types_by_name["Displacement"]
    assert [(f.name, f.number) for f in d.fields] == [("feature_idx", 1), ("di", 2), ("dj", 3)]


def test_producer_side_bytes_are_read_by_the_native_scanner(lfr_lib, tmp_path):
    """A MatchingFile written the way compute_match_graph.py:163-205 writes it (message API of the shim) parses into
    the same graph as the flat-array contract."""
    ma = synthetic.generate(seed=5, n_images=7, n_tracks=25, eps_out=0.05)
    msg = types_pb2.MatchingFile()
    for p in ma.to_pairs():
        ip = msg.image_pairs.add()
        ip.image_name1, ip.fact1, ip.image_name2, ip.fact2 = p["image_name1"], p["fact1"], p["image_name2"], p["fact2"]
        for m in p["matches"]:
            mm = ip.matches.add()
            mm.feature_idx1, mm.feature_idx2, mm.similarity = m["feature_idx1"], m["feature_idx2"], m["similarity"]
            for name in ("disp1", "disp2"):
                for (a, b) in m[name]:
                    dd = getattr(mm, name).add()
                    dd.di, dd.dj = a, b
    path = tmp_path / "m.pb"
    path.write_bytes(msg.SerializeToString())
    g_file = capi.Graph.from_matches_file(str(path))
    g_arr = capi.Graph.from_arrays(ma)
    assert g_file.n_nodes == g_arr.n_nodes and g_file.n_edges == g_arr.n_edges
    assert (g_file.nodes()[0] == g_arr.nodes()[0]).all() and (g_file.nodes()[1] == g_arr.nodes()[1]).all()
    assert (capi.Problem(g_file, device_assembly=True).labels()[0] == capi.Problem(g_arr, device_assembly=True).labels()[0]).all()


def test_consumer_side_parses_a_solve_output_like_import_features(lfr_lib, tmp_path):
    """colmap_utils.import_features (colmap_utils.py:104-137) restated on top of the shim: parse the SolutionFile the
    drop-in's emitter wrote, apply `displacements[feature_idx] = [dj, di]; *= fact; keypoints += displacements * 16;
    += 0.5` in float32, and compare with lfr_apply_displacements (the same arithmetic as a library call)."""
    ma = synthetic.generate(seed=6, n_images=6, n_tracks=30)
    ma.facts[:] = np.float32([1.0, 0.5, 2.0, 1.0, 0.25, 1.0])
    g = capi.Graph.from_arrays(ma)
    rng = np.random.default_rng(0)
    pos = rng.uniform(-0.6, 0.6, (g.n_nodes, 2))
    out = tmp_path / "s.pb"
    g.write_solution(pos, str(out))
    sol = types_pb2.SolutionFile()
    sol.ParseFromString(out.read_bytes())                                  # colmap_utils.py:105-107
    image_proto_idx = {image.image_name: idx for idx, image in enumerate(sol.images)}   # :109-111
    assert wire.decode_solution_file(out.read_bytes())[0]["image_name"] == sol.images[0].image_name
    img_of_node, feat = g.nodes()
    for image_name in g.image_names() + ["not_in_the_graph.png"]:
        n_feat = 40
        keypoints = rng.uniform(0, 500, (n_feat, 4)).astype(np.float32)
        want = keypoints.copy()
        if image_name in image_proto_idx:                                  # :126-136
            image_proto = sol.images[image_proto_idx[image_name]]
            displacements = np.zeros([n_feat, 2], np.float32)
            for dsp in image_proto.displacements:
                displacements[dsp.feature_idx, :] = [dsp.dj, dsp.di]
            displacements *= np.float32(image_proto.fact)
            want[:, :2] += displacements * np.float32(16)
        want[:, :2] += np.float32(0.5)                                     # :137
        got = g.apply_displacements(pos, image_name, keypoints.copy())
        assert (got == want).all(), image_name


def test_install_writes_an_importable_top_level_module(tmp_path):
    d = tmp_path / "reconstruction-scripts"
    d.mkdir()
    written = types_pb2.install([str(d)])
    assert written == [str(d / "types_pb2.py")]
    code = "import types_pb2; m = types_pb2.SolutionFile(); i = m.images.add(); i.image_name = 'a'; print(len(m.SerializeToString()))"
    r = subprocess.run([sys.executable, "-c", code], cwd=str(d), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env={k: v for k, v in os.environ.items() if k != "PYTHONPATH"})
    assert r.returncode == 0, r.stderr
    assert int(r.stdout) == 5                      # tag + len + (tag + len + 'a')
    (d / "types_pb2.py").write_text("# generated by protoc\n")
    try:
        types_pb2.install([str(d)])
        assert False, "must not overwrite a foreign file"
    except FileExistsError:
        pass

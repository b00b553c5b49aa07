"""google.protobuf message classes for types.proto (protoc is not available): the product's runtime-descriptor shim,
lfr_amd/types_pb2.py, plus a helper that fills a MatchingFile from the dict form of lfr_amd.wire."""
from lfr_amd import types_pb2


def build():
    return types_pb2.MatchingFile, types_pb2.SolutionFile


def pairs_to_pb(MatchingFile, pairs):
    msg = MatchingFile()
    for p in pairs:
        ip = msg.image_pairs.add()
        ip.image_name1, ip.fact1, ip.image_name2, ip.fact2 = p["image_name1"], p["fact1"], p["image_name2"], p["fact2"]
        for m in p["matches"]:
            mm = ip.matches.add()
            mm.feature_idx1, mm.feature_idx2, mm.similarity = m["feature_idx1"], m["feature_idx2"], m["similarity"]
            for (a, b) in m["disp1"]:
                d = mm.disp1.add()
                d.di, d.dj = a, b
            for (a, b) in m["disp2"]:
                d = mm.disp2.add()
                d.di, d.dj = a, b
    return msg

"""`types_pb2`-compatible message classes for the solver's file contracts without protoc.

The reference's Python sides import a protoc-generated module: `from types_pb2 import MatchingFile`
(two-view-refinement/compute_match_graph.py:24, the producer of the match graph) and `import types_pb2`
(reconstruction-scripts/colmap_utils.py:17,104-111,163-165, the consumer of the SolutionFile).  That module is
git-ignored there and needs `protoc` (README.md:20-23).  This one builds the same two message classes from a runtime
FileDescriptorProto that restates types.proto:3-46 field for field (names, numbers, types, nesting), so

    from lfr_amd.types_pb2 import MatchingFile, SolutionFile

gives classes with the generated API (`ParseFromString`, `SerializeToString`, `image_pairs.add()`, ...), byte-compatible
with the reference's files and with `liblfr_hip.so`'s native scanner / emitter (tests/test_types_pb2_shim.py).

    python -m lfr_amd.types_pb2 --install two-view-refinement reconstruction-scripts

drops a `types_pb2.py` stub into the directories where protoc would have written its output, so the reference's scripts
run unchanged on a machine without protoc.
"""
import os
import sys

from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_F = descriptor_pb2.FieldDescriptorProto


def _field(msg, name, number, ftype, label=_F.LABEL_OPTIONAL, type_name=None):
    f = msg.field.add()
    f.name, f.number, f.type, f.label = name, number, ftype, label
    if type_name:
        f.type_name = type_name


def _file_descriptor_proto():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "types.proto"
    fd.syntax = "proto3"
    mf = fd.message_type.add()                                   # types.proto:3-28
    mf.name = "MatchingFile"
    ip = mf.nested_type.add()
    ip.name = "ImagePair"
    _field(ip, "image_name1", 1, _F.TYPE_STRING)
    _field(ip, "fact1", 2, _F.TYPE_FLOAT)
    _field(ip, "image_name2", 3, _F.TYPE_STRING)
    _field(ip, "fact2", 4, _F.TYPE_FLOAT)
    m = ip.nested_type.add()
    m.name = "Match"
    _field(m, "feature_idx1", 1, _F.TYPE_UINT32)
    _field(m, "feature_idx2", 2, _F.TYPE_UINT32)
    _field(m, "similarity", 3, _F.TYPE_FLOAT)
    d = m.nested_type.add()
    d.name = "Displacement"
    _field(d, "di", 1, _F.TYPE_FLOAT)
    _field(d, "dj", 2, _F.TYPE_FLOAT)
    _field(m, "disp1", 4, _F.TYPE_MESSAGE, _F.LABEL_REPEATED, ".MatchingFile.ImagePair.Match.Displacement")
    _field(m, "disp2", 5, _F.TYPE_MESSAGE, _F.LABEL_REPEATED, ".MatchingFile.ImagePair.Match.Displacement")
    _field(ip, "matches", 5, _F.TYPE_MESSAGE, _F.LABEL_REPEATED, ".MatchingFile.ImagePair.Match")
    _field(mf, "image_pairs", 1, _F.TYPE_MESSAGE, _F.LABEL_REPEATED, ".MatchingFile.ImagePair")
    sf = fd.message_type.add()                                   # types.proto:30-46
    sf.name = "SolutionFile"
    im = sf.nested_type.add()
    im.name = "Image"
    _field(im, "image_name", 1, _F.TYPE_STRING)
    _field(im, "fact", 2, _F.TYPE_FLOAT)
    sd = im.nested_type.add()
    sd.name = "Displacement"
    _field(sd, "feature_idx", 1, _F.TYPE_UINT32)
    _field(sd, "di", 2, _F.TYPE_FLOAT)
    _field(sd, "dj", 3, _F.TYPE_FLOAT)
    _field(im, "displacements", 3, _F.TYPE_MESSAGE, _F.LABEL_REPEATED, ".SolutionFile.Image.Displacement")
    _field(sf, "images", 1, _F.TYPE_MESSAGE, _F.LABEL_REPEATED, ".SolutionFile.Image")
    return fd


def _build():
    pool = descriptor_pool.DescriptorPool()          # a private pool: a real generated types_pb2 may live in the default one
    fd = _file_descriptor_proto()
    if hasattr(pool, "AddSerializedFile"):           # every backend (python, upb, cpp) has it; Add() is gone from some builds
        pool.AddSerializedFile(fd.SerializeToString())
    else:
        pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    if get is None:                                   # protobuf < 4.21
        get = message_factory.MessageFactory(pool).GetPrototype
    mf = get(pool.FindMessageTypeByName("MatchingFile"))
    sf = get(pool.FindMessageTypeByName("SolutionFile"))
    return pool.FindFileByName("types.proto"), mf, sf


DESCRIPTOR, MatchingFile, SolutionFile = _build()
__all__ = ["DESCRIPTOR", "MatchingFile", "SolutionFile"]

_STUB = '''"""types_pb2 stand-in written by `python -m lfr_amd.types_pb2 --install` (no protoc on this machine): re-exports the
runtime-built message classes of types.proto (lfr_amd/types_pb2.py)."""
import sys

sys.path.insert(0, %r)
from lfr_amd.types_pb2 import DESCRIPTOR, MatchingFile, SolutionFile  # noqa: E402,F401
'''


def install(directories):
    """Write a `types_pb2.py` stub into each directory (where `protoc --python_out=DIR types.proto` would put its
    output, README.md:20-23).  Refuses to overwrite a file it did not write."""
    pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    written = []
    for d in directories:
        path = os.path.join(d, "types_pb2.py")
        if os.path.exists(path) and "lfr_amd.types_pb2" not in open(path).read():
            raise FileExistsError("%s exists and was not written by this tool (a protoc-generated module?)" % path)
        with open(path, "w") as f:
            f.write(_STUB % pkg_parent)
        written.append(path)
    return written


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--install":
        for p in install(sys.argv[2:]):
            print("wrote", p)
    else:
        sys.stderr.write("usage: python -m lfr_amd.types_pb2 --install DIR [DIR ...]\n")
        sys.exit(2)

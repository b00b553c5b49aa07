"""google.protobuf message classes for types.proto, built from a runtime descriptor
(protoc is not available).  Field numbers/types follow the reference's types.proto:3-46."""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_F = descriptor_pb2.FieldDescriptorProto


def _field(msg, name, number, ftype, label=_F.LABEL_OPTIONAL, type_name=None):
    f = msg.field.add()
    f.name, f.number, f.type, f.label = name, number, ftype, label
    if type_name:
        f.type_name = type_name


def build():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "lfr_types_runtime.proto"
    fd.syntax = "proto3"
    mf = fd.message_type.add()
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
    sf = fd.message_type.add()
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
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    if get is None:
        fac = message_factory.MessageFactory(pool)
        get = fac.GetPrototype
    return (get(pool.FindMessageTypeByName("MatchingFile")), get(pool.FindMessageTypeByName("SolutionFile")))


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

"""Dependency-free proto3 codec for the two messages of the solver boundary.

Schema: reference ``types.proto:3-28`` (MatchingFile) and ``types.proto:30-46``
(SolutionFile).  Only varint (wire type 0), fixed32 (5) and length-delimited (2)
fields occur.  This pure-Python codec is the *readable* statement of the wire
format: it is used by tests, by the golden-fixture generator and as the checker
of the native scanner/emitter in ``csrc/lfr_wire.cpp`` — the product path itself
reads and writes files natively.

Python-side representation
--------------------------
MatchingFile  -> list of ImagePair dicts::

    {"image_name1": str, "fact1": float, "image_name2": str, "fact2": float,
     "matches": [{"feature_idx1": int, "feature_idx2": int, "similarity": float,
                  "disp1": [(di, dj), ...], "disp2": [(di, dj), ...]}, ...]}

SolutionFile  -> list of Image dicts::

    {"image_name": str, "fact": float,
     "displacements": [(feature_idx, di, dj), ...]}

All floats are float32 on the wire (``types.proto:5,7,13,16-17,33,38-39``).
proto3 serializers omit scalar fields equal to zero; parsers must default them.
"""
import struct

_F32 = struct.Struct("<f")


# ----------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------
def _put_varint(out, v):
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)


def _get_varint(buf, pos):
    shift = 0
    result = 0
    while True:
        if pos >= len(buf):
            raise ValueError("truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def _put_tag(out, field, wt):
    _put_varint(out, (field << 3) | wt)


def _put_f32(out, field, v):
    # proto3: default-valued scalars are not emitted.  -0.0 is emitted (its bits
    # are non-zero), matching the C++/upb serializers.
    raw = _F32.pack(v)
    if raw != b"\x00\x00\x00\x00":
        _put_tag(out, field, 5)
        out += raw


def _put_u32(out, field, v):
    if v:
        _put_tag(out, field, 0)
        _put_varint(out, v)


def _put_bytes(out, field, payload, always=True):
    if payload or always:
        _put_tag(out, field, 2)
        _put_varint(out, len(payload))
        out += payload


def _put_str(out, field, s):
    raw = s.encode("utf-8")
    if raw:
        _put_bytes(out, field, raw)


def _fields(buf):
    """Yield (field, wire_type, value) over one message body; value is an int
    for varint, bytes(4) for fixed32, bytes(8) for fixed64, memoryview for
    length-delimited."""
    pos = 0
    n = len(buf)
    while pos < n:
        key, pos = _get_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if field == 0:
            raise ValueError("field number 0")
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 5:
            if pos + 4 > n:
                raise ValueError("truncated fixed32")
            v = bytes(buf[pos:pos + 4])
            pos += 4
        elif wt == 1:
            if pos + 8 > n:
                raise ValueError("truncated fixed64")
            v = bytes(buf[pos:pos + 8])
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            if pos + ln > n:
                raise ValueError("truncated length-delimited field")
            v = buf[pos:pos + ln]
            pos += ln
        else:
            raise ValueError("unsupported wire type %d" % wt)
        yield field, wt, v


def _f32(v):
    return _F32.unpack(v)[0]


# ----------------------------------------------------------------------------
# MatchingFile
# ----------------------------------------------------------------------------
def _enc_disp(d):
    out = bytearray()
    _put_f32(out, 1, d[0])
    _put_f32(out, 2, d[1])
    return out


def _enc_match(m):
    out = bytearray()
    _put_u32(out, 1, int(m["feature_idx1"]))
    _put_u32(out, 2, int(m["feature_idx2"]))
    _put_f32(out, 3, m["similarity"])
    for d in m.get("disp1", ()):
        _put_bytes(out, 4, _enc_disp(d))
    for d in m.get("disp2", ()):
        _put_bytes(out, 5, _enc_disp(d))
    return out


def _enc_pair(p):
    out = bytearray()
    _put_str(out, 1, p["image_name1"])
    _put_f32(out, 2, p["fact1"])
    _put_str(out, 3, p["image_name2"])
    _put_f32(out, 4, p["fact2"])
    for m in p.get("matches", ()):
        _put_bytes(out, 5, _enc_match(m))
    return out


def encode_matching_file(image_pairs):
    """Serialize a MatchingFile (``types.proto:3-28``)."""
    out = bytearray()
    for p in image_pairs:
        _put_bytes(out, 1, _enc_pair(p))
    return bytes(out)


def _dec_disp(buf):
    di = dj = 0.0
    for f, wt, v in _fields(buf):
        if f == 1 and wt == 5:
            di = _f32(v)
        elif f == 2 and wt == 5:
            dj = _f32(v)
    return (di, dj)


def _dec_match(buf):
    m = {"feature_idx1": 0, "feature_idx2": 0, "similarity": 0.0, "disp1": [], "disp2": []}
    for f, wt, v in _fields(buf):
        if f == 1 and wt == 0:
            m["feature_idx1"] = v & 0xFFFFFFFF
        elif f == 2 and wt == 0:
            m["feature_idx2"] = v & 0xFFFFFFFF
        elif f == 3 and wt == 5:
            m["similarity"] = _f32(v)
        elif f == 4 and wt == 2:
            m["disp1"].append(_dec_disp(v))
        elif f == 5 and wt == 2:
            m["disp2"].append(_dec_disp(v))
    return m


def _dec_pair(buf):
    p = {"image_name1": "", "fact1": 0.0, "image_name2": "", "fact2": 0.0, "matches": []}
    for f, wt, v in _fields(buf):
        if f == 1 and wt == 2:
            p["image_name1"] = bytes(v).decode("utf-8")
        elif f == 2 and wt == 5:
            p["fact1"] = _f32(v)
        elif f == 3 and wt == 2:
            p["image_name2"] = bytes(v).decode("utf-8")
        elif f == 4 and wt == 5:
            p["fact2"] = _f32(v)
        elif f == 5 and wt == 2:
            p["matches"].append(_dec_match(v))
    return p


def decode_matching_file(data):
    """Parse a serialized MatchingFile into a list of ImagePair dicts."""
    buf = memoryview(data)
    pairs = []
    for f, wt, v in _fields(buf):
        if f == 1 and wt == 2:
            pairs.append(_dec_pair(v))
    return pairs


# ----------------------------------------------------------------------------
# SolutionFile
# ----------------------------------------------------------------------------
def encode_solution_file(images):
    """Serialize a SolutionFile (``types.proto:30-46``)."""
    out = bytearray()
    for im in images:
        body = bytearray()
        _put_str(body, 1, im["image_name"])
        _put_f32(body, 2, im["fact"])
        for (fidx, di, dj) in im["displacements"]:
            d = bytearray()
            _put_u32(d, 1, int(fidx))
            _put_f32(d, 2, di)
            _put_f32(d, 3, dj)
            _put_bytes(body, 3, d)
        _put_bytes(out, 1, body)
    return bytes(out)


def decode_solution_file(data):
    """Parse a serialized SolutionFile into a list of Image dicts."""
    buf = memoryview(data)
    images = []
    for f, wt, v in _fields(buf):
        if not (f == 1 and wt == 2):
            continue
        im = {"image_name": "", "fact": 0.0, "displacements": []}
        for f2, wt2, v2 in _fields(v):
            if f2 == 1 and wt2 == 2:
                im["image_name"] = bytes(v2).decode("utf-8")
            elif f2 == 2 and wt2 == 5:
                im["fact"] = _f32(v2)
            elif f2 == 3 and wt2 == 2:
                fidx, di, dj = 0, 0.0, 0.0
                for f3, wt3, v3 in _fields(v2):
                    if f3 == 1 and wt3 == 0:
                        fidx = v3 & 0xFFFFFFFF
                    elif f3 == 2 and wt3 == 5:
                        di = _f32(v3)
                    elif f3 == 3 and wt3 == 5:
                        dj = _f32(v3)
                im["displacements"].append((fidx, di, dj))
        images.append(im)
    return images

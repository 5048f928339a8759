"""A minimal binary FBX writer for the importer tests: node records with 32-bit (version < 7500) or 64-bit offsets,
scalar / string / array properties, optional zlib compression of arrays."""
import struct
import zlib

import numpy as np

MAGIC = b"Kaydara FBX Binary  \x00\x1a\x00"


class Z:
    """Marks an array property to be stored deflated."""

    def __init__(self, array):
        self.array = array


def _prop(v):
    if isinstance(v, Z):
        return _array(v.array, True)
    if isinstance(v, np.ndarray):
        return _array(v, False)
    if isinstance(v, bool):
        return b"C" + struct.pack("<B", int(v))
    if isinstance(v, int):
        return (b"I" + struct.pack("<i", v)) if -2**31 <= v < 2**31 and not getattr(v, "_long", False) else (b"L" + struct.pack("<q", v))
    if isinstance(v, float):
        return b"D" + struct.pack("<d", v)
    if isinstance(v, str):
        v = v.encode()
    if isinstance(v, bytes):
        return b"S" + struct.pack("<I", len(v)) + v
    raise TypeError(type(v))


def _array(a, compress):
    code = {np.dtype("f8"): b"d", np.dtype("f4"): b"f", np.dtype("i4"): b"i", np.dtype("i8"): b"l"}[a.dtype]
    raw = np.ascontiguousarray(a).tobytes()
    stored = zlib.compress(raw) if compress else raw
    return code + struct.pack("<III", a.size, 1 if compress else 0, len(stored)) + stored


class L(int):
    """An int that is always written as a 64-bit 'L' property (object ids)."""
    _long = True


def node(name, props=(), children=()):
    return (name, list(props), list(children))


def _serialise(n, pos, wide):
    name, props, children = n
    header = 25 if wide else 13
    body = b"".join(_prop(p) for p in props)
    out = bytearray()
    at = pos + header + len(name) + len(body)
    kids = bytearray()
    for c in children:
        b = _serialise(c, at + len(kids), wide)
        kids += b
    if children:
        kids += bytes(header)
    end = at + len(kids)
    fmt = "<QQQB" if wide else "<IIIB"
    out += struct.pack(fmt, end, len(props), len(body), len(name)) + name.encode() + body + kids
    return bytes(out)


def write(path, top_nodes, version=7400):
    wide = version >= 7500
    data = bytearray(MAGIC + struct.pack("<I", version))
    for n in top_nodes:
        data += _serialise(n, len(data), wide)
    data += bytes(25 if wide else 13)
    data += b"\xfa\xbc\xab\x09\xd0\xc8\xd4\x66\xb1\x76\xfb\x83\x1c\xf7\x26\x7e" + bytes(4) + struct.pack("<I", version) + bytes(120) + \
        b"\xf8\x5a\x8c\x6a\xde\xf5\xd9\x7e\xec\xe9\x0c\xe3\x75\x8f\x29\x0b"
    with open(path, "wb") as f:
        f.write(data)


def p70(**values):
    """Properties70 with one P record per keyword: a 3-vector, a number, or an int."""
    kids = []
    for key, v in values.items():
        name = key.replace("_", " ")
        if isinstance(v, (tuple, list)):
            kids.append(node("P", [name, "Vector3D", "Vector", "A"] + [float(x) for x in v]))
        elif isinstance(v, int):
            kids.append(node("P", [name, "int", "Integer", "", v]))
        else:
            kids.append(node("P", [name, "double", "Number", "A", float(v)]))
    return node("Properties70", [], kids)


def obj(kind, oid, name, cls, sub, children=()):
    return node(kind, [L(oid), name.encode() + b"\x00\x01" + cls.encode(), sub], children)


def oo(child, parent):
    return node("C", ["OO", L(child), L(parent)])


def op(child, parent, prop):
    return node("C", ["OP", L(child), L(parent), prop])

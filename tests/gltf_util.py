"""A tiny glTF 2.0 writer for the importer tests: accessors over one binary buffer, .gltf (+ .bin / data URIs) or .glb."""
import base64
import json
import struct

import numpy as np

_CT = {np.dtype("f4"): 5126, np.dtype("u4"): 5125, np.dtype("u2"): 5123, np.dtype("u1"): 5121}
_TYPE = {1: "SCALAR", 2: "VEC2", 3: "VEC3", 4: "VEC4", 16: "MAT4"}


class GltfWriter:
    def __init__(self):
        self.bin = bytearray()
        self.doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": []}], "nodes": [], "meshes": [], "materials": [],
                    "accessors": [], "bufferViews": [], "buffers": [{}]}

    def view(self, data: bytes, stride=None):
        while len(self.bin) % 4:
            self.bin.append(0)
        v = {"buffer": 0, "byteOffset": len(self.bin), "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        self.bin += data
        self.doc["bufferViews"].append(v)
        return len(self.doc["bufferViews"]) - 1

    def accessor(self, array, minmax=False, normalized=False):
        a = np.ascontiguousarray(array)
        comps = 1 if a.ndim == 1 else a.shape[1]
        acc = {"bufferView": self.view(a.tobytes()), "componentType": _CT[a.dtype], "count": int(a.shape[0]), "type": _TYPE[comps]}
        if minmax:
            acc["min"], acc["max"] = a.min(axis=0).tolist(), a.max(axis=0).tolist()
        if normalized:
            acc["normalized"] = True
        self.doc["accessors"].append(acc)
        return len(self.doc["accessors"]) - 1

    def primitive(self, pos, idx=None, nrm=None, uv=None, tangent=None, material=None, joints=None, weights=None):
        attrs = {"POSITION": self.accessor(np.asarray(pos, np.float32), minmax=True)}
        if nrm is not None:
            attrs["NORMAL"] = self.accessor(np.asarray(nrm, np.float32))
        if uv is not None:
            attrs["TEXCOORD_0"] = self.accessor(np.asarray(uv, np.float32))
        if tangent is not None:
            attrs["TANGENT"] = self.accessor(np.asarray(tangent, np.float32))
        if joints is not None:
            attrs["JOINTS_0"] = self.accessor(np.asarray(joints, np.uint16))
            attrs["WEIGHTS_0"] = self.accessor(np.asarray(weights, np.float32))
        p = {"attributes": attrs}
        if idx is not None:
            p["indices"] = self.accessor(np.asarray(idx, np.uint16 if np.max(idx) < 65536 else np.uint32).reshape(-1))
        if material is not None:
            p["material"] = material
        return p

    def mesh(self, primitives, name=None):
        self.doc["meshes"].append({"primitives": primitives, **({"name": name} if name else {})})
        return len(self.doc["meshes"]) - 1

    def node(self, parent=None, **fields):
        self.doc["nodes"].append(dict(fields))
        i = len(self.doc["nodes"]) - 1
        if parent is None:
            self.doc["scenes"][0]["nodes"].append(i)
        else:
            self.doc["nodes"][parent].setdefault("children", []).append(i)
        return i

    def material(self, **fields):
        self.doc["materials"].append(dict(fields))
        return len(self.doc["materials"]) - 1

    def image_texture(self, png_bytes: bytes, mode: str, directory=None, filename="tex.png"):
        """mode: 'uri' (external file), 'data' (data URI) or 'view' (buffer view)."""
        self.doc.setdefault("images", [])
        self.doc.setdefault("textures", [])
        if mode == "uri":
            (directory / filename).write_bytes(png_bytes)
            self.doc["images"].append({"uri": filename})
        elif mode == "data":
            self.doc["images"].append({"uri": "data:image/png;base64," + base64.b64encode(png_bytes).decode()})
        else:
            self.doc["images"].append({"bufferView": self.view(png_bytes), "mimeType": "image/png"})
        self.doc["textures"].append({"source": len(self.doc["images"]) - 1})
        return len(self.doc["textures"]) - 1

    def _finish(self):
        doc = {k: v for k, v in self.doc.items() if v not in ([], {})}
        doc["buffers"] = [{"byteLength": len(self.bin)}]
        return doc

    def write_gltf(self, path, external_bin=True):
        doc = self._finish()
        if external_bin:
            (path.parent / (path.stem + ".bin")).write_bytes(bytes(self.bin))
            doc["buffers"][0]["uri"] = path.stem + ".bin"
        else:
            doc["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(bytes(self.bin)).decode()
        path.write_text(json.dumps(doc))

    def write_glb(self, path):
        js = json.dumps(self._finish()).encode()
        js += b" " * (-len(js) % 4)
        b = bytes(self.bin) + b"\0" * (-len(self.bin) % 4)
        total = 12 + 8 + len(js) + 8 + len(b)
        path.write_bytes(b"glTF" + struct.pack("<II", 2, total) + struct.pack("<I4s", len(js), b"JSON") + js + struct.pack("<I4s", len(b), b"BIN\0") + b)


def quad(size=1.0, y=0.0):
    """A y-up quad facing +y with counter-clockwise winding and uv 0..1."""
    s = size
    pos = np.float32([[-s, y, -s], [s, y, -s], [s, y, s], [-s, y, s]])
    nrm = np.float32([[0, 1, 0]] * 4)
    uv = np.float32([[0, 0], [1, 0], [1, 1], [0, 1]])
    idx = np.uint16([0, 2, 1, 0, 3, 2])
    return pos, nrm, uv, idx


def cube(h=0.5):
    pos, nrm, idx = [], [], []
    for axis in range(3):
        for sign in (-1.0, 1.0):
            n = np.zeros(3)
            n[axis] = sign
            u, v = np.zeros(3), np.zeros(3)
            u[(axis + 1) % 3] = 1
            v[(axis + 2) % 3] = 1
            if sign < 0:
                u, v = v, u
            base = len(pos)
            for a, b in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
                pos.append((n + a * u + b * v) * h)
                nrm.append(n)
            idx += [base, base + 1, base + 2, base, base + 2, base + 3]
    return np.float32(pos), np.float32(nrm), np.uint16(idx)

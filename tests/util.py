"""Helpers shared by the tests."""
import ctypes as C
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def open_golden(name):
    """A golden-vector file (text mode): they are committed gzip-compressed (decimal uint32 bit patterns compress 4x)."""
    import gzip

    return gzip.open(os.path.join(GOLDEN_DIR, name + ".gz"), "rt")

GEOMETRY_DT = np.dtype([("VertexOffset", "u4"), ("VertexLength", "u4"), ("IndexOffset", "u4"), ("IndexLength", "u4"),
                        ("IsOpaque", "u1"), ("IsAnimated", "u1"), ("pad", "u1", 2)])
MESH_DT = np.dtype([("GeometryIndex", "u4"), ("MaterialId", "u4"), ("TransformIndex", "u4")])
MODEL_DT = np.dtype([("MeshOffset", "u4"), ("MeshCount", "u4")])
INSTANCE_DT = np.dtype([("ModelIndex", "u4"), ("Transform", "f4", 12)])


def _view(ptr, count, dtype):
    if not ptr or not count:
        return np.zeros(0, dtype)
    buf = (C.c_uint8 * (count * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count)


def desc_arrays(desc):
    return {
        "vertices": _view(desc.vertices, desc.vertexCount * 14, np.dtype("f4")).reshape(-1, 14),
        "indices": _view(desc.indices, desc.indexCount, np.dtype("u4")),
        "transforms": _view(desc.transforms, desc.transformCount * 12, np.dtype("f4")).reshape(-1, 12),
        "geometries": _view(desc.geometries, desc.geometryCount, GEOMETRY_DT),
        "meshes": _view(desc.meshes, desc.meshCount, MESH_DT),
        "models": _view(desc.models, desc.modelCount, MODEL_DT),
        "instances": _view(desc.instances, desc.instanceCount, INSTANCE_DT),
    }


def pair_first(desc):
    """First global triangle id of every (instance, mesh) pair, in instance-then-mesh order --
    the numbering both the oracle (global id) and the HIP path ((pair, prim)) use."""
    a = desc_arrays(desc)
    first, tri = [], 0
    for inst in a["instances"]:
        m = a["models"][inst["ModelIndex"]]
        for k in range(m["MeshCount"]):
            rec = a["meshes"][m["MeshOffset"] + k]
            first.append(tri)
            tri += int(a["geometries"][rec["GeometryIndex"]]["IndexLength"]) // 3
    first.append(tri)
    return np.array(first, dtype=np.int64)


def load_golden(mode):
    with open_golden(f"golden_{mode}.json") as f:
        g = json.load(f)
    out = {}
    for name, c in g.items():
        out[name] = (c["fn"], np.array(c["in"], dtype=np.uint32).reshape(-1, c["nin"]),
                     np.array(c["out"], dtype=np.uint32).reshape(-1, c["nout"]))
    return out


def bits_equal_or_both_nan(a_u32, b_u32):
    af, bf = a_u32.view(np.float32), b_u32.view(np.float32)
    return (a_u32 == b_u32) | (np.isnan(af) & np.isnan(bf))


def rel_l2(img, ref):
    return float(np.linalg.norm((img[..., :3] - ref[..., :3]).astype(np.float64)) /
                 max(np.linalg.norm(ref[..., :3].astype(np.float64)), 1e-30))


def random_rays(rng, n, lo, hi, tmax=1e4):
    """Rays with origins in the box [lo,hi] pointing in random directions (float32, 8 per ray)."""
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = o
    rays[:, 3] = 1e-5
    rays[:, 4:7] = d
    rays[:, 7] = tmax
    return rays


def render_pair(pkg, orc, name, detail, W, H, frames, depth, backend=0, brute=False, lens=0.0, sample_count=1, tile=None):
    """Render `frames` launches with the HIP path and with the oracle; returns both accumulation images
    after checking that segment / shadow-ray / sample / retry counts agree launch by launch."""
    scene = pkg.Scene(name, detail)
    lights = scene.lights
    r = pkg.Renderer(backend=backend)
    r.upload(scene)
    r.resize(W, H)
    if tile:
        r.set_tile_shard(0, 1, tile)
    osc = orc.OracleScene(scene.desc, build_bvh=not brute)
    ref = np.zeros((H, W, 4), np.float32)
    seg = shadow = 0
    for f in range(frames):
        u = scene.uniform(W, H, bounces=depth, sample_count=sample_count, total_samples=f * sample_count, lens_radius=lens, focal_distance=6.0)
        r.render(u, lights)
        st = r.stats()
        _, ost = osc.render(u, lights, W, H, accum=ref, brute_force=brute)
        assert st.segments == ost.segments and st.shadowRays == ost.shadowRays, "segment counts differ"
        assert st.pathSamples == ost.pathSamples and st.retries == ost.retries
        seg += st.segments
        shadow += st.shadowRays
    img = r.readback()
    r.close()
    return img, ref


def known_answer_rays():
    """Rays found by tools/full_size_sweep.py on which a tree walk once disagreed with brute force:
    (scene, origin bits, direction bits).  street_like: rays crossing a triangle next to an edge, 12 - 40 units from the
    camera, where plain Moeller-Trumbore also accepted the neighbouring triangle (equal t, smaller id) although the point
    lay outside its box; atrium_like: a zero-area triangle (e1 == e2) that the triangle test used to "hit" at a
    meaningless t."""
    return [
        ("street_like", (0xC2080000, 0x3FD9999C, 0x3F800000), (0x3F4C1984, 0xBD9D0A84, 0xBF194709)),
        ("street_like", (0xC2080000, 0x3FD9999C, 0x3F800000), (0x3F7DA424, 0x3C974EA6, 0x3E09645E)),
        ("street_like", (0xC2080000, 0x3FD9999C, 0x3F800000), (0x3F7CF07B, 0xBCDD1A10, 0x3E1B6E52)),
        ("atrium_like", (0xC1880001, 0x3FE66663, 0xBF4CCCCD), (0x3F741F4C, 0x3E64162A, 0x3E4F6AC2)),
    ]


def known_answer_ray_array(scene_name):
    rays = []
    for name, o, d in known_answer_rays():
        if name == scene_name:
            ob, db = np.array(o, np.uint32).view(np.float32), np.array(d, np.uint32).view(np.float32)
            rays.append([ob[0], ob[1], ob[2], 1e-5, db[0], db[1], db[2], 1e4])
    return np.array(rays, np.float32)


def reciprocal_inputs():
    """Bit patterns for the specified reciprocal (PTX_FN_DIVIDE): all 2^23 mantissas of [1, 2) with both signs (the Newton step is
    exact-scaling invariant inside the normal range), every exponent with 4,096 mantissas each (the flush boundaries 2^-126 and
    2^126 lie in there), zero, denormals, infinities, NaNs and the patterns either side of both boundaries."""
    m = np.arange(1 << 23, dtype=np.uint32)
    parts = [np.uint32(0x3f800000) | m, np.uint32(0xbf800000) | m]
    rng = np.random.default_rng(11)
    mant = np.concatenate([np.uint32([0, 1, 0x7fffff, 0x7ffffe, 0x400000]), rng.integers(0, 1 << 23, 4091, dtype=np.uint32)])
    for e in range(256):
        parts.append((np.uint32(e << 23) | mant))
        parts.append((np.uint32(0x80000000 | (e << 23)) | mant))
    parts.append(np.uint32([0, 0x80000000, 1, 0x007fffff, 0x00800000, 0x00800001, 0x7e800000, 0x7e800001, 0x7e7fffff, 0x7f7fffff, 0x7f800000,
                            0xff800000, 0x7fc00000, 0xffc00001, 0x7f800001]))
    return np.concatenate(parts)


def reciprocal_spec(b):
    """The definition (oracle/pt_oracle_math.h pto_rcp) in numpy: RN(1 / b) on [2^-126, 2^126], +-inf below, +-0 above."""
    b = b.astype(np.float32)
    with np.errstate(all="ignore"):
        r = (np.float32(1.0) / b).astype(np.float32)
        ab = np.abs(b)
        r = np.where(ab < np.float32(1.17549435e-38), np.copysign(np.float32(np.inf), b), r)
        r = np.where(ab > np.float32(8.50705917e37), np.copysign(np.float32(0.0), b), r)
        r = np.where(np.isnan(b), b, r)
    return r.astype(np.float32)

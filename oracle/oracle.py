"""ctypes wrapper of the CPU ORACLE (oracle/libpt_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libpt_oracle.so")


class OracleStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("pathSamples", "segments", "shadowRays", "retries", "triangles", "nodesVisited", "trisTested")]


class OracleHit(C.Structure):
    _fields_ = [("t", C.c_float), ("u", C.c_float), ("v", C.c_float), ("tri", C.c_uint32)]


def build(force: bool = False) -> None:
    src = [os.path.join(HERE, f) for f in ("pt_oracle.c", "pt_oracle_post.c", "pt_oracle.h", "pt_oracle_math.h")]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in src):
        subprocess.check_call(["make", "-C", HERE, "-B" if force else "-s", "libpt_oracle.so"])


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        lib = C.CDLL(LIB)
        P = C.c_void_p
        lib.pto_scene_create.restype = P
        lib.pto_scene_create.argtypes = [P, C.c_int]
        lib.pto_scene_create_posed.restype = P
        lib.pto_scene_create_posed.argtypes = [P, P, P, C.c_uint32, C.c_int]
        lib.pto_scene_destroy.argtypes = [P]
        lib.pto_scene_triangle_count.restype = C.c_uint64
        lib.pto_scene_triangle_count.argtypes = [P]
        lib.pto_render.argtypes = [P, P, P] + [C.c_uint32] * 6 + [P, P, C.c_int, C.c_int, P]
        lib.pto_trace_closest.argtypes = [P, P, C.c_uint32, P, C.c_int]
        lib.pto_trace_any.argtypes = [P, P, C.c_uint32, P, C.c_int]
        lib.pto_test_eval.argtypes = [C.c_uint32, P, P, C.c_uint32]
        lib.pto_rsq_selfcheck.argtypes = [C.POINTER(C.c_uint64)]
        lib.pto_rsq_selfcheck.restype = C.c_uint64
        lib.pto_test_raygen.argtypes = [P, P, C.c_uint32]
        lib.pto_test_closest_hit.argtypes = [P, P, P, P, C.c_uint32]
        lib.pto_test_any_hit.argtypes = [P, P, P, C.c_uint32]
        lib.pto_test_miss.argtypes = [P, P, P, C.c_uint32]
        lib.pto_test_texture.argtypes = [P, P, P, C.c_uint32, C.c_int]
        lib.pto_postprocess.argtypes = [P, C.c_uint32, C.c_uint32, P, C.c_uint32, P]
        lib.pto_encode_output.argtypes = [P, C.c_uint32, C.c_uint32, C.c_uint32, P]
        _lib = lib
    return _lib


class OracleScene:
    def __init__(self, desc, build_bvh: bool = True, instance_transforms=None, bones=None):
        """desc: a ctypes image of PtxSceneDesc (anything with ctypes.byref support).  instance_transforms
        (n x 12 float32) / bones (m x 12 float32): the state of one animated frame (Scene::Update)."""
        self.lib = load()
        self._desc = desc  # keep alive
        it = None if instance_transforms is None else np.ascontiguousarray(instance_transforms, np.float32).reshape(-1, 12)
        bn = None if bones is None else np.ascontiguousarray(bones, np.float32).reshape(-1, 12)
        self.handle = self.lib.pto_scene_create_posed(C.addressof(desc), it.ctypes.data if it is not None else None,
                                                      bn.ctypes.data if bn is not None else None, 0 if bn is None else bn.shape[0],
                                                      int(build_bvh))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.pto_scene_destroy(self.handle)
            self.handle = None

    __del__ = close

    @property
    def triangle_count(self) -> int:
        return int(self.lib.pto_scene_triangle_count(self.handle))

    def render(self, uniform, lights, width, height, accum=None, region=None, shard=None, threads=0,
               brute_force=False):
        if accum is None:
            accum = np.zeros((height, width, 4), np.float32)
        x0, y0, x1, y1 = region if region else (0, 0, width, height)
        stats = OracleStats()
        shard_p = C.addressof(shard) if shard is not None else None
        rc = self.lib.pto_render(self.handle, C.addressof(uniform), C.addressof(lights), width, height, x0, y0, x1, y1,
                                 shard_p, accum.ctypes.data, threads, int(brute_force), C.addressof(stats))
        if rc:
            raise RuntimeError("pto_render failed")
        return accum, stats

    def test_texture(self, inputs: np.ndarray, implicit_lod: bool = False) -> np.ndarray:
        inputs = np.ascontiguousarray(inputs).view(np.uint32).reshape(-1, 7)
        out = np.zeros((inputs.shape[0], 4), np.uint32)
        self.lib.pto_test_texture(self.handle, inputs.ctypes.data, out.ctypes.data, inputs.shape[0], int(implicit_lod))
        return out

    def test_miss(self, directions: np.ndarray) -> np.ndarray:
        """pto_test_miss: miss.rmiss main() for rows of 3 words (the ray direction) -> Emissive, Pdf."""
        directions = np.ascontiguousarray(directions, np.uint32).reshape(-1, 3)
        out = np.zeros((directions.shape[0], 4), np.uint32)
        if self.lib.pto_test_miss(self.handle, directions.ctypes.data, out.ctypes.data, directions.shape[0]):
            raise RuntimeError("pto_test_miss failed")
        return out

    def test_any_hit(self, inputs: np.ndarray) -> np.ndarray:
        """pto_test_any_hit: anyhit.rahit / occlusionAnyhit.rahit main() on triangle 0 for rows of 8 words."""
        inputs = np.ascontiguousarray(inputs, np.uint32).reshape(-1, 8)
        out = np.zeros((inputs.shape[0], 7), np.uint32)
        if self.lib.pto_test_any_hit(self.handle, inputs.ctypes.data, out.ctypes.data, inputs.shape[0]):
            raise RuntimeError("pto_test_any_hit failed")
        return out

    def test_closest_hit(self, lights, inputs: np.ndarray) -> np.ndarray:
        """pto_test_closest_hit: closestHit.rchit main() on triangle 0 for rows of 28 words (layout in pt_oracle.c)."""
        inputs = np.ascontiguousarray(inputs, np.uint32).reshape(-1, 28)
        out = np.zeros((inputs.shape[0], 35), np.uint32)
        if self.lib.pto_test_closest_hit(self.handle, C.addressof(lights), inputs.ctypes.data, out.ctypes.data, inputs.shape[0]):
            raise RuntimeError("pto_test_closest_hit failed")
        return out

    def trace_closest(self, rays: np.ndarray, brute_force: bool = False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.zeros(rays.shape[0], dtype=[("t", "f4"), ("u", "f4"), ("v", "f4"), ("tri", "u4")])
        self.lib.pto_trace_closest(self.handle, rays.ctypes.data, rays.shape[0], hits.ctypes.data, int(brute_force))
        return hits

    def trace_any(self, rays: np.ndarray, brute_force: bool = False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        occ = np.zeros(rays.shape[0], np.uint32)
        self.lib.pto_trace_any(self.handle, rays.ctypes.data, rays.shape[0], occ.ctypes.data, int(brute_force))
        return occ


# words per case of pto_test_eval's inputs / outputs, by function id (include/ptx.h PTX_FN_*)
IN_STRIDE = (4, 4, 4, 2, 1, 10, 11, 6, 3, 14, 12, 4, 2, 2, 3, 6, 38, 1, 2, 31, 25, 42, 30, 24, 12, 34, 35, 4, 3, 3, 2, 6, 7, 3, 47, 2, 1, 1)
OUT_STRIDE = (1, 1, 1, 1, 1, 4, 4, 3, 4, 4, 8, 5, 2, 3, 9, 3, 18, 2, 1, 9, 3, 18, 12, 6, 4, 12, 12, 1, 2, 3, 2, 6, 3, 3, 17, 2, 1, 1)


def rsq_selfcheck():
    """(classes where pto_rsq or the device's float sequence differs from the exact RN(1 / sqrt(x)), classes where the seed matters)"""
    dep = C.c_uint64(0)
    wrong = load().pto_rsq_selfcheck(C.byref(dep))
    return int(wrong), int(dep.value)


def test_eval(fn: int, inputs: np.ndarray, nout: int) -> np.ndarray:
    lib = load()
    inputs = np.ascontiguousarray(inputs).view(np.uint32)
    if inputs.ndim != 2 or inputs.shape[1] != IN_STRIDE[fn] or nout != OUT_STRIDE[fn]:
        raise ValueError(f"function {fn} takes {IN_STRIDE[fn]} words and returns {OUT_STRIDE[fn]} per case, "
                         f"got {inputs.shape} and nout = {nout}")
    n = inputs.shape[0]
    out = np.zeros((n, nout), np.uint32)
    rc = lib.pto_test_eval(fn, inputs.ctypes.data, out.ctypes.data, n)
    if rc:
        raise RuntimeError("pto_test_eval failed")
    return out


RAYGEN_IN_WORDS, RAYGEN_OUT_WORDS = 44 + 12 * 23, 7


def test_raygen(inputs: np.ndarray) -> np.ndarray:
    """pto_test_raygen: raygen.rgen main() for one pixel per row, trace calls scripted (layout in pt_oracle.c)."""
    lib = load()
    inputs = np.ascontiguousarray(inputs, np.uint32)
    if inputs.ndim != 2 or inputs.shape[1] != RAYGEN_IN_WORDS:
        raise ValueError(f"a case is {RAYGEN_IN_WORDS} words, got {inputs.shape}")
    out = np.zeros((inputs.shape[0], RAYGEN_OUT_WORDS), np.uint32)
    if lib.pto_test_raygen(inputs.ctypes.data, out.ctypes.data, inputs.shape[0]):
        raise RuntimeError("pto_test_raygen failed")
    return out


def postprocess(accum: np.ndarray, total_samples: int, exposure: float = 1.0, bloom_threshold: float = 1.0,
                bloom_intensity: float = 1.0, tone_mapping: int = 0) -> np.ndarray:
    """ptx_postprocess on a host array: H x W x 4 running sum -> tone-mapped linear image (binary16-valued)."""
    lib = load()
    accum = np.ascontiguousarray(accum, dtype=np.float32)
    h, w = accum.shape[:2]
    u = (C.c_uint32 * 4)()
    u[0] = total_samples
    C.memmove(C.addressof(u) + 4, np.array([exposure, bloom_threshold, bloom_intensity], np.float32).ctypes.data, 12)
    out = np.empty((h, w, 4), np.float32)
    if lib.pto_postprocess(accum.ctypes.data, w, h, C.addressof(u), tone_mapping, out.ctypes.data):
        raise RuntimeError("pto_postprocess failed")
    return out


def encode_output(linear: np.ndarray, fmt: int = 0) -> np.ndarray:
    lib = load()
    linear = np.ascontiguousarray(linear, dtype=np.float32)
    h, w = linear.shape[:2]
    out = np.empty((h, w, 4), np.float32 if fmt == 1 else np.uint8)
    if lib.pto_encode_output(linear.ctypes.data, w, h, fmt, out.ctypes.data):
        raise RuntimeError("pto_encode_output failed")
    return out

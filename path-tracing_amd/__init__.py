"""path-tracing_amd -- Python plumbing over the C-ABIs of the MI355X path-tracing backend.

This package holds NO rendering logic: it only loads
  * libptx_hip.so   (include/ptx.h)       HIP kernels + renderer   -- the product
  * libptx_host.so  (include/ptx_host.h)  C++ mirror of the reference's Scene/SceneBuilder/
                                          Camera/ExampleScenes (CPU only)
through ctypes, mirroring the call sequence of the reference's Renderer
(Path-Tracing/Renderer/Renderer.h:42-85): create -> scene_upload -> build_accel -> resize ->
render ... -> readback.

The directory name contains a hyphen, so import it with `load_package()` from
`__graft_entry__` / `tests/conftest.py` (importlib by path) under the module name
`path_tracing_amd`.

There is deliberately no CPU fallback: `Renderer()` raises if the HIP library is missing or no
GPU is visible.  The CPU oracle lives in /oracle and is test infrastructure only.
"""
from __future__ import annotations

import atexit
import ctypes as C
import glob
import os
import subprocess
import sys
import weakref

import numpy as np

# The runtime maps all HIP streams of the process onto GPU_MAX_HW_QUEUES hardware queues (default 4); frames in flight need
# one per stream to overlap (csrc/pt_runtime.hpp, hardwareQueuesGranted).  It is read at the first HIP call, so it has to
# be in the environment before torch initialises the device -- and it is the host's to set, which for Python callers is
# this package at import time (no thread of ours exists yet); the C library only reports what it finds.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
HIP_LIB = os.environ.get("PTX_HIP_LIB") or os.path.join(PKG_DIR, "libptx_hip.so")  # PTX_HIP_LIB: an experimental build of the same ABI
HOST_LIB = os.environ.get("PTX_HOST_LIB") or os.path.join(PKG_DIR, "libptx_host.so")  # PTX_HOST_LIB: e.g. a sanitizer build

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fno-fast-math",  # arithmetic conventions of csrc/pt_device.hpp
    "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
    # No SLP vectorisation: under plain -O3 the compiler packs adjacent scalar f32 multiplies and adds into v_pk_mul_f32 /
    # v_pk_add_f32, whose operands must sit in aligned register PAIRS -- the shading kernels then spend a tenth of their VALU
    # instructions on v_mov shuffling (731 -> 391 in k_shade<false>) and every kernel pays in registers: k_trace_closest 64 -> 54
    # VGPRs, k_shade<false> 168 with 15 spilled -> 168 with none, k_shade<true> 238 -> 221, k_generate 57 -> 51 (8 waves).  Same
    # IEEE operations, same bits.  Measured (1 MI355X, 1080p, 8 spp, two runs each in one call): chess_like 2,139 / 2,158 ->
    # 2,265 / 2,290 Msamples/s, street_like 1,135 / 1,145 -> 1,176 / 1,174, temple_like 822 / 827 -> 843 / 846, atrium_like flat.
    "-fno-slp-vectorize",
]
HOST_FLAGS = ["-std=c++20", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", "-fvisibility=hidden"]

# every symbol include/ptx.h and include/ptx_host.h declare
PTX_SYMBOLS = [
    "ptx_create", "ptx_destroy", "ptx_last_error", "ptx_device_count", "ptx_abi_version", "ptx_scene_upload", "ptx_build_accel", "ptx_share_scene",
    "ptx_resize", "ptx_set_tile_shard", "ptx_set_backend", "ptx_reset_accumulation", "ptx_render",
    "ptx_render_frames", "ptx_synchronize", "ptx_readback", "ptx_readback_begin", "ptx_readback_end", "ptx_device_accum_ptr", "ptx_accum_bytes",
    "ptx_shard_bytes", "ptx_pack_shard", "ptx_unpack_shard", "ptx_unpack_shard_host", "ptx_unpack_shards", "ptx_bind_shard_accumulation", "ptx_get_stats", "ptx_bind_accumulation",
    "ptx_trace_rays", "ptx_test_input_stride", "ptx_test_output_stride", "ptx_test_eval", "ptx_test_texture",
    "ptx_postprocess", "ptx_read_output", "ptx_write_accumulation", "ptx_update_animation",
]
PTH_SYMBOLS = [
    "pth_scene_names", "pth_scene_create", "pth_scene_destroy", "pth_last_error", "pth_scene_desc",
    "pth_scene_lights", "pth_scene_triangle_count", "pth_scene_raygen_uniform", "pth_scene_set_active_camera",
    "pth_scene_set_camera_pose", "pth_scene_update", "pth_scene_bone_count", "pth_scene_animation_state", "pth_decode_image", "pth_decode_image_levels", "pth_write_image", "pth_save_checkpoint", "pth_load_checkpoint",
]

BACKEND_WAVEFRONT = 0
BACKEND_MEGAKERNEL = 1


# ---------------------------------------------------------------------------------------
# ctypes images of the PODs in include/ptx.h
# ---------------------------------------------------------------------------------------
class SceneDesc(C.Structure):
    _fields_ = [
        ("vertices", C.c_void_p), ("vertexCount", C.c_uint64),
        ("indices", C.c_void_p), ("indexCount", C.c_uint64),
        ("transforms", C.c_void_p), ("transformCount", C.c_uint32),
        ("geometries", C.c_void_p), ("geometryCount", C.c_uint32),
        ("metallicRoughnessMaterials", C.c_void_p), ("metallicRoughnessMaterialCount", C.c_uint32),
        ("specularGlossinessMaterials", C.c_void_p), ("specularGlossinessMaterialCount", C.c_uint32),
        ("phongMaterials", C.c_void_p), ("phongMaterialCount", C.c_uint32),
        ("meshes", C.c_void_p), ("meshCount", C.c_uint32),
        ("models", C.c_void_p), ("modelCount", C.c_uint32),
        ("instances", C.c_void_p), ("instanceCount", C.c_uint32),
        ("skyboxKind", C.c_uint32), ("dxNormalTextures", C.c_uint32),
        ("textures", C.c_void_p), ("textureCount", C.c_uint32), ("forceFullTextureSize", C.c_uint32),
        ("skybox", C.c_void_p),
        ("animatedVertices", C.c_void_p), ("animatedVertexCount", C.c_uint64),
        ("animatedIndices", C.c_void_p), ("animatedIndexCount", C.c_uint64),
        ("textureMemoryBudget", C.c_uint64),
    ]


class RaygenUniformData(C.Structure):
    _fields_ = [
        ("ViewInverse", C.c_float * 16), ("ProjInverse", C.c_float * 16), ("BounceCount", C.c_uint32),
        ("LensRadius", C.c_float), ("FocalDistance", C.c_float), ("SampleCount", C.c_uint32),
        ("TotalSamples", C.c_uint32),
    ]


class DirectionalLight(C.Structure):
    _fields_ = [("Color", C.c_float * 3), ("pad0", C.c_float), ("Direction", C.c_float * 3), ("pad1", C.c_float)]


class PointLight(C.Structure):
    _fields_ = [
        ("Color", C.c_float * 3), ("pad0", C.c_float), ("Position", C.c_float * 3), ("pad1", C.c_float),
        ("AttenuationConstant", C.c_float), ("AttenuationLinear", C.c_float), ("AttenuationQuadratic", C.c_float),
        ("pad2", C.c_float),
    ]


class LightsUbo(C.Structure):
    _fields_ = [("LightCount", C.c_uint32), ("pad", C.c_uint32 * 3), ("Directional", DirectionalLight),
                ("Lights", PointLight * 64)]


class PostProcessingUniformData(C.Structure):
    _fields_ = [("TotalSamples", C.c_uint32), ("Exposure", C.c_float), ("BloomThreshold", C.c_float), ("BloomIntensity", C.c_float)]


TONE_MAPPING_SDR, TONE_MAPPING_HDR = 0, 1
ACCEL_REFIT, ACCEL_REBUILD = 0, 1
OUTPUT_RGBA8_SRGB, OUTPUT_RGBA32F = 0, 1
ABI_VERSION = 5  # PTX_ABI_VERSION of include/ptx.h


DEVICE_SINGLE_STREAM = 1  # PTX_DEVICE_SINGLE_STREAM


class DeviceDesc(C.Structure):
    _fields_ = [("deviceIndex", C.c_int32), ("backend", C.c_uint32), ("stream", C.c_void_p), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class TileShard(C.Structure):
    _fields_ = [("rank", C.c_uint32), ("worldSize", C.c_uint32), ("tileSize", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [
        ("pathSamples", C.c_uint64), ("segments", C.c_uint64), ("shadowRays", C.c_uint64), ("retries", C.c_uint64),
        ("triangles", C.c_uint64), ("bvhNodes", C.c_uint64), ("lastRenderMs", C.c_double), ("lastTraceMs", C.c_double),
        ("lastBuildMs", C.c_double), ("traceLaunches", C.c_uint64), ("lastShadeMs", C.c_double), ("lastShadowMs", C.c_double), ("lastTailMs", C.c_double),
        ("tracedRays", C.c_uint64), ("hardwareQueues", C.c_uint64), ("treeTriangles", C.c_uint64), ("treeReferences", C.c_uint64),
    ]


assert C.sizeof(RaygenUniformData) == 148 and C.sizeof(LightsUbo) == 3120 and C.sizeof(PointLight) == 48

FN = {
    "GGXDistribution": 0, "Lambda": 1, "GGXSmith": 2, "DielectricFresnel": 3, "SchlickFresnel": 4,
    "EvaluateReflection": 5, "EvaluateRefraction": 6, "SampleGGX": 7, "sampleLobePdfs": 8, "evaluateBSDF": 9,
    "sampleBSDF": 10, "rng": 11, "sampleUniformDiskConcentric": 12, "sampleCosineHemisphere": 13,
    "computeTangentSpace": 14, "offsetRayOriginSelfIntersection": 15, "constructPrimaryRay": 16, "sincos": 17,
    "pow": 18, "sampleLight": 19, "offsetRayOriginShadowTerminator": 20, "constructPrimaryRayLens": 21,
    "computeDpnDuv": 22, "computeDpDxy": 23, "computeDerivatives": 24, "computeReflectedDifferentialRays": 25,
    "computeRefractedDifferentialRays": 26, "computeLod": 27, "missSkyboxTexCoords": 28, "hdrToLdr": 29, "atanAsin": 30,
    "postprocessPixel": 31, "compositionPixel": 32, "toneMapPixel": 33, "sampleMaterial": 34,
    "divide": 35, "sqrt": 36, "rsq": 37,
}


# ---------------------------------------------------------------------------------------
# build (used by __graft_entry__.build)
# ---------------------------------------------------------------------------------------
def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


HIP_TRANSLATION_UNIT = "ptx_capi.hip"  # the one translation unit of libptx_hip.so; it includes every header of csrc/


def hip_sources():
    """Every source file of the HIP library, the translation unit first (build dependencies, source digests of the profiles)."""
    csrc = os.path.join(PKG_DIR, "csrc")
    return [os.path.join(csrc, HIP_TRANSLATION_UNIT)] + sorted(glob.glob(os.path.join(csrc, "*.hpp")))


def build(force: bool = False, verbose: bool = True) -> None:
    """Compile the HIP extension for gfx950 and the C++ host mirror, in-tree."""
    csrc = os.path.join(PKG_DIR, "csrc")
    hip_src = [os.path.join(csrc, HIP_TRANSLATION_UNIT)] + hip_sources()[1:] + [os.path.join(REPO_DIR, "include", "ptx.h")]
    if force or _newer(HIP_LIB, hip_src):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc] + HIPCC_FLAGS + ["-o", HIP_LIB, hip_src[0]]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    host = os.path.join(PKG_DIR, "host")
    host_src = [os.path.join(host, f) for f in ("Scene.cpp", "Camera.cpp", "ExampleScenes.cpp", "OutputSaver.cpp", "TextureImporter.cpp", "JpegDecoder.cpp", "SceneImporter.cpp", "SceneDescription.cpp", "FbxReader.cpp", "ObjReader.cpp", "host_capi.cpp")]
    # every header and source of host/ (a stale library after an edit to a header that is not listed is a silent wrong test)
    host_dep = sorted(glob.glob(os.path.join(host, "*.h")) + glob.glob(os.path.join(host, "*.cpp"))) + [
        os.path.join(REPO_DIR, "include", "ptx_host.h"), os.path.join(REPO_DIR, "include", "ptx.h")]
    if force or _newer(HOST_LIB, host_dep):
        cmd = ["g++"] + HOST_FLAGS + ["-o", HOST_LIB] + host_src
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)


# ---------------------------------------------------------------------------------------
# library loading
# ---------------------------------------------------------------------------------------
_hip = None
_host = None


def load_host() -> C.CDLL:
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise RuntimeError(f"{HOST_LIB} missing: run __graft_entry__.build()")
        lib = C.CDLL(HOST_LIB)
        lib.pth_scene_names.restype = C.c_char_p
        lib.pth_last_error.restype = C.c_char_p
        lib.pth_scene_create.restype = C.c_void_p
        lib.pth_scene_create.argtypes = [C.c_char_p, C.c_float, C.c_uint32]
        lib.pth_scene_destroy.argtypes = [C.c_void_p]
        lib.pth_scene_desc.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        lib.pth_scene_lights.argtypes = [C.c_void_p, C.POINTER(LightsUbo)]
        lib.pth_scene_triangle_count.restype = C.c_uint64
        lib.pth_scene_triangle_count.argtypes = [C.c_void_p]
        lib.pth_scene_raygen_uniform.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float,
                                                 C.c_uint32, C.c_uint32, C.POINTER(RaygenUniformData)]
        lib.pth_scene_set_active_camera.argtypes = [C.c_void_p, C.c_int32]
        lib.pth_scene_set_camera_pose.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.pth_scene_update.argtypes = [C.c_void_p, C.c_float]
        lib.pth_scene_bone_count.argtypes = [C.c_void_p]
        lib.pth_scene_bone_count.restype = C.c_uint32
        lib.pth_scene_animation_state.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        lib.pth_decode_image.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.c_void_p, C.c_size_t]
        lib.pth_decode_image_levels.argtypes = [C.c_char_p, C.c_size_t]
        lib.pth_decode_image_levels.restype = C.c_uint32
        lib.pth_write_image.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]
        lib.pth_save_checkpoint.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.pth_load_checkpoint.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p, C.c_size_t]
        _host = lib
    return _host


def load_hip() -> C.CDLL:
    """Load the HIP renderer.  If torch is in use, import it BEFORE calling this so both share
    one HIP runtime (the soname libamdhip64.so.7 is resolved to the copy already loaded)."""
    global _hip
    if _hip is None:
        if "torch" not in sys.modules:
            # torch brings its own libamdhip64; loaded after this library the process would hold two HIP runtimes, and the
            # one this library is bound to then finds no device (seen as ptx_create -> PTX_ERROR_NO_DEVICE when
            # __graft_entry__.build() and smoke() ran in one process)
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        if not os.path.exists(HIP_LIB):
            raise RuntimeError(f"{HIP_LIB} missing: the HIP extension is not built (run __graft_entry__.build())")
        lib = C.CDLL(HIP_LIB)
        have = None
        if hasattr(lib, "ptx_abi_version"):
            lib.ptx_abi_version.restype = C.c_uint32
            have = lib.ptx_abi_version()
        if have != ABI_VERSION and not os.environ.get("PTX_ABI_UNCHECKED"):  # PTX_ABI_UNCHECKED: A/B runs against an older experimental build
            raise RuntimeError(f"{HIP_LIB} has ABI {have}, this package was written against {ABI_VERSION} (include/ptx.h): rebuild")
        P = C.c_void_p
        lib.ptx_create.argtypes = [C.POINTER(DeviceDesc), C.POINTER(P)]
        lib.ptx_destroy.argtypes = [P]
        lib.ptx_destroy.restype = None
        lib.ptx_last_error.argtypes = [P]
        lib.ptx_last_error.restype = C.c_char_p
        lib.ptx_scene_upload.argtypes = [P, C.POINTER(SceneDesc)]
        lib.ptx_build_accel.argtypes = [P]
        lib.ptx_share_scene.argtypes = [P, P]
        lib.ptx_resize.argtypes = [P, C.c_uint32, C.c_uint32]
        lib.ptx_set_tile_shard.argtypes = [P, C.POINTER(TileShard)]
        lib.ptx_set_backend.argtypes = [P, C.c_uint32]
        lib.ptx_reset_accumulation.argtypes = [P]
        lib.ptx_render.argtypes = [P, C.POINTER(RaygenUniformData), C.POINTER(LightsUbo)]
        lib.ptx_render_frames.argtypes = [P, C.POINTER(RaygenUniformData), C.POINTER(LightsUbo), C.c_uint32, C.c_uint32]
        lib.ptx_synchronize.argtypes = [P]
        lib.ptx_readback.argtypes = [P, P, C.c_size_t]
        lib.ptx_readback_begin.argtypes = [P, P, C.c_size_t]
        lib.ptx_readback_end.argtypes = [P]
        lib.ptx_device_accum_ptr.argtypes = [P]
        lib.ptx_device_accum_ptr.restype = P
        lib.ptx_accum_bytes.argtypes = [P]
        lib.ptx_accum_bytes.restype = C.c_size_t
        lib.ptx_shard_bytes.argtypes = [P, C.c_uint32]
        lib.ptx_shard_bytes.restype = C.c_size_t
        lib.ptx_pack_shard.argtypes = [P, P]
        lib.ptx_unpack_shard.argtypes = [P, C.c_uint32, P]
        lib.ptx_unpack_shard_host.argtypes = [P, C.c_uint32, P, P, C.c_size_t]
        lib.ptx_unpack_shards.argtypes = [P, P, C.c_size_t, C.c_int, P, C.c_size_t]
        lib.ptx_bind_shard_accumulation.argtypes = [P, P, C.c_size_t]
        lib.ptx_get_stats.argtypes = [P, C.POINTER(Stats)]
        lib.ptx_postprocess.argtypes = [P, C.POINTER(PostProcessingUniformData), C.c_uint32]
        lib.ptx_read_output.argtypes = [P, C.c_uint32, P, C.c_size_t]
        lib.ptx_write_accumulation.argtypes = [P, P, C.c_size_t]
        lib.ptx_update_animation.argtypes = [P, P, C.c_uint32, P, C.c_uint32, C.c_uint32]
        lib.ptx_bind_accumulation.argtypes = [P, P, C.c_size_t]
        lib.ptx_trace_rays.argtypes = [P, P, C.c_uint32, C.c_int, P, P]
        lib.ptx_test_input_stride.argtypes = [C.c_uint32]
        lib.ptx_test_output_stride.argtypes = [C.c_uint32]
        lib.ptx_test_eval.argtypes = [P, C.c_uint32, P, P, C.c_uint32]
        lib.ptx_test_texture.argtypes = [P, P, P, C.c_uint32, C.c_int]
        _hip = lib
    return _hip


class PtxError(RuntimeError):
    pass


# ---------------------------------------------------------------------------------------
# thin object wrappers
# ---------------------------------------------------------------------------------------
class Scene:
    """A host-side scene (PathTracing::Scene built by ExampleScenes::CreateScene)."""

    def __init__(self, name: str = "default", detail: float = 1.0, seed: int = 0):
        self.lib = load_host()
        self.name = name
        self.handle = self.lib.pth_scene_create(name.encode(), float(detail), int(seed))
        if not self.handle:
            raise PtxError(self.lib.pth_last_error().decode())

    def close(self):
        if getattr(self, "handle", None):
            self.lib.pth_scene_destroy(self.handle)
            self.handle = None

    __del__ = close

    @property
    def desc(self) -> SceneDesc:
        d = SceneDesc()
        if self.lib.pth_scene_desc(self.handle, C.byref(d)):
            raise PtxError("pth_scene_desc failed")
        return d

    @property
    def lights(self) -> LightsUbo:
        l = LightsUbo()
        self.lib.pth_scene_lights(self.handle, C.byref(l))
        return l

    @property
    def triangle_count(self) -> int:
        return int(self.lib.pth_scene_triangle_count(self.handle))

    def uniform(self, width, height, bounces=4, sample_count=1, total_samples=0, lens_radius=0.0,
                focal_distance=10.0) -> RaygenUniformData:
        u = RaygenUniformData()
        rc = self.lib.pth_scene_raygen_uniform(self.handle, width, height, bounces, lens_radius, focal_distance,
                                               sample_count, total_samples, C.byref(u))
        if rc:
            raise PtxError("pth_scene_raygen_uniform failed")
        return u

    def update(self, time_step: float) -> bool:
        """Scene::Update(timeStep): True if something the renderer consumes has moved."""
        rc = self.lib.pth_scene_update(self.handle, float(time_step))
        if rc < 0:
            raise PtxError("pth_scene_update failed")
        return bool(rc)

    def animation_state(self):
        """(instance transforms n x 12, bone matrices m x 12) as Scene::Update left them."""
        n, m = self.desc.instanceCount, self.lib.pth_scene_bone_count(self.handle)
        it, bn = np.zeros((n, 12), np.float32), np.zeros((m, 12), np.float32)
        if self.lib.pth_scene_animation_state(self.handle, it.ctypes.data, n, bn.ctypes.data if m else None, m):
            raise PtxError("pth_scene_animation_state failed")
        return it, bn

    def set_active_camera(self, camera_id: int):
        if self.lib.pth_scene_set_active_camera(self.handle, camera_id):
            raise PtxError("bad camera id")

    def set_camera_pose(self, position, direction):
        p = (C.c_float * 3)(*position)
        d = (C.c_float * 3)(*direction)
        self.lib.pth_scene_set_camera_pose(self.handle, p, d)


_live_renderers: "weakref.WeakSet[Renderer]" = weakref.WeakSet()


@atexit.register
def _close_renderers():
    # A renderer still alive at interpreter exit would be destroyed by __del__ during module teardown, possibly after
    # the HIP runtime's own exit handlers have run (seen as "double free or corruption" at exit): destroy them first.
    for r in list(_live_renderers):
        r.close()


class Renderer:
    """include/ptx.h as an object.  Raises PtxError (with ptx_last_error) on any failure."""

    def __init__(self, device: int = 0, backend: int = BACKEND_WAVEFRONT, stream: int | None = None, single_stream: bool = False):
        self.lib = load_hip()
        self.handle = C.c_void_p()
        desc = DeviceDesc(device, backend, stream, DEVICE_SINGLE_STREAM if single_stream else 0, 0)
        rc = self.lib.ptx_create(C.byref(desc), C.byref(self.handle))
        if rc:
            self.handle = None
            raise PtxError(f"ptx_create failed with status {rc} (no HIP device? there is no CPU fallback)")
        self.width = self.height = 0
        _live_renderers.add(self)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ptx_destroy(self.handle)
            self.handle = None

    __del__ = close

    def _check(self, rc):
        if rc:
            raise PtxError(f"status {rc}: {self.lib.ptx_last_error(self.handle).decode()}")

    def upload(self, scene: Scene | SceneDesc):
        d = scene.desc if isinstance(scene, Scene) else scene
        self._check(self.lib.ptx_scene_upload(self.handle, C.byref(d)))
        self._check(self.lib.ptx_build_accel(self.handle))

    def share_scene(self, owner: "Renderer"):
        """Render `owner`'s scene and tree instead of holding copies (ptx_share_scene): frames in flight share one scene."""
        self._check(self.lib.ptx_share_scene(self.handle, owner.handle))

    def resize(self, width: int, height: int):
        self._check(self.lib.ptx_resize(self.handle, width, height))
        self.width, self.height = width, height

    def set_tile_shard(self, rank: int, world: int, tile: int = 32):
        s = TileShard(rank, world, tile)
        self._check(self.lib.ptx_set_tile_shard(self.handle, C.byref(s)))

    def set_backend(self, backend: int):
        self._check(self.lib.ptx_set_backend(self.handle, backend))

    def reset(self):
        self._check(self.lib.ptx_reset_accumulation(self.handle))

    def render(self, uniform: RaygenUniformData, lights: LightsUbo):
        self._check(self.lib.ptx_render(self.handle, C.byref(uniform), C.byref(lights)))

    def render_frames(self, uniform: RaygenUniformData, lights: LightsUbo, first_frame: int, frames: int):
        self._check(self.lib.ptx_render_frames(self.handle, C.byref(uniform), C.byref(lights), first_frame, frames))

    def synchronize(self):
        self._check(self.lib.ptx_synchronize(self.handle))

    def readback(self) -> np.ndarray:
        img = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(self.lib.ptx_readback(self.handle, img.ctypes.data, img.nbytes))
        return img

    def readback_begin(self, pinned_ptr: int, nbytes: int):
        """ptx_readback_begin into page-locked host memory (e.g. a torch tensor with pin_memory=True)."""
        self._check(self.lib.ptx_readback_begin(self.handle, pinned_ptr, nbytes))

    def readback_end(self):
        self._check(self.lib.ptx_readback_end(self.handle))

    def update_animation(self, instance_transforms=None, bones=None, rebuild: bool = False):
        """ptx_update_animation: n x 12 instance transforms and / or m x 12 bone matrices, then refit (or rebuild)."""
        it = None if instance_transforms is None else np.ascontiguousarray(instance_transforms, np.float32).reshape(-1, 12)
        bn = None if bones is None else np.ascontiguousarray(bones, np.float32).reshape(-1, 12)
        self._check(self.lib.ptx_update_animation(self.handle, it.ctypes.data if it is not None else None, 0 if it is None else it.shape[0],
                                                  bn.ctypes.data if bn is not None else None, 0 if bn is None else bn.shape[0],
                                                  ACCEL_REBUILD if rebuild else ACCEL_REFIT))

    def write_accumulation(self, img: np.ndarray):
        img = np.ascontiguousarray(img, dtype=np.float32)
        self._check(self.lib.ptx_write_accumulation(self.handle, img.ctypes.data, img.nbytes))

    def postprocess(self, total_samples: int, exposure: float = 1.0, bloom_threshold: float = 1.0, bloom_intensity: float = 1.0,
                    tone_mapping: int = TONE_MAPPING_SDR):
        u = PostProcessingUniformData(total_samples, exposure, bloom_threshold, bloom_intensity)
        self._check(self.lib.ptx_postprocess(self.handle, C.byref(u), tone_mapping))

    def read_output(self, fmt: int = OUTPUT_RGBA8_SRGB) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), dtype=np.float32 if fmt == OUTPUT_RGBA32F else np.uint8)
        self._check(self.lib.ptx_read_output(self.handle, fmt, out.ctypes.data, out.nbytes))
        return out

    def stats(self) -> Stats:
        s = Stats()
        self._check(self.lib.ptx_get_stats(self.handle, C.byref(s)))
        return s

    def accum_ptr(self) -> int:
        return int(self.lib.ptx_device_accum_ptr(self.handle) or 0)

    def bind_accumulation(self, dev_ptr: int, nbytes: int):
        self._check(self.lib.ptx_bind_accumulation(self.handle, dev_ptr, nbytes))

    def shard_bytes(self, rank: int) -> int:
        return int(self.lib.ptx_shard_bytes(self.handle, rank))

    def pack_shard(self, dev_dst: int):
        self._check(self.lib.ptx_pack_shard(self.handle, dev_dst))

    def unpack_shard(self, rank: int, dev_src: int, pinned_host: int = 0, nbytes: int = 0):
        """ptx_unpack_shard; with `pinned_host` (address of a page-locked width*height*16-byte buffer) ptx_unpack_shard_host: the
        shard's pixels also go straight to the host's frame (complete after every rank's shard and readback_end())."""
        if pinned_host:
            self._check(self.lib.ptx_unpack_shard_host(self.handle, rank, dev_src, pinned_host, nbytes))
        else:
            self._check(self.lib.ptx_unpack_shard(self.handle, rank, dev_src))

    def unpack_shards(self, dev_src: int, stride_bytes: int, to_device_image: bool = True, pinned_host: int = 0, nbytes: int = 0):
        """ptx_unpack_shards: the whole gathered frame (every rank's shard, `stride_bytes` apart in `dev_src`) in one launch, to
        the device image and / or the host's page-locked frame (complete after readback_end())."""
        self._check(self.lib.ptx_unpack_shards(self.handle, dev_src, stride_bytes, 1 if to_device_image else 0, pinned_host or None, nbytes))

    def bind_shard_accumulation(self, dev_shard: int, nbytes: int = 0):
        """ptx_bind_shard_accumulation: accumulate this rank's samples in the dense tile-major shard buffer (0 unbinds)."""
        self._check(self.lib.ptx_bind_shard_accumulation(self.handle, dev_shard or None, nbytes))

    def trace_rays(self, rays: np.ndarray, any_hit: bool = False):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        hits = np.zeros((n, 4), np.float32)
        ids = np.zeros((n, 2), np.uint32)
        self._check(self.lib.ptx_trace_rays(self.handle, rays.ctypes.data, n, int(any_hit), hits.ctypes.data,
                                            ids.ctypes.data))
        return hits, ids

    def test_texture(self, inputs: np.ndarray, implicit_lod: bool = False) -> np.ndarray:
        inputs = np.ascontiguousarray(inputs).view(np.uint32).reshape(-1, 7)
        out = np.zeros((inputs.shape[0], 4), np.uint32)
        self._check(self.lib.ptx_test_texture(self.handle, inputs.ctypes.data, out.ctypes.data, inputs.shape[0], int(implicit_lod)))
        return out

    def test_eval(self, fn: int, inputs: np.ndarray) -> np.ndarray:
        nin, nout = self.lib.ptx_test_input_stride(fn), self.lib.ptx_test_output_stride(fn)
        inputs = np.ascontiguousarray(inputs).view(np.uint32).reshape(-1, nin)
        out = np.zeros((inputs.shape[0], nout), np.uint32)
        self._check(self.lib.ptx_test_eval(self.handle, fn, inputs.ctypes.data, out.ctypes.data, inputs.shape[0]))
        return out


OUTPUT_PNG, OUTPUT_JPG, OUTPUT_TGA, OUTPUT_HDR = 0, 1, 2, 3


def decode_image(data: bytes):
    """TextureImporter::Decode: returns (H x W x 4 array, uint8 or float32; channels in the file)."""
    lib = load_host()
    info = (C.c_uint32 * 4)()
    if lib.pth_decode_image(data, len(data), info, None, 0):
        raise PtxError(lib.pth_last_error().decode())
    img = np.empty((info[1], info[0], 4), np.float32 if info[3] else np.uint8)
    if lib.pth_decode_image(data, len(data), info, img.ctypes.data, img.nbytes):
        raise PtxError(lib.pth_last_error().decode())
    return img, int(info[2])


def decode_image_levels(data: bytes):
    """The file's whole mip chain (DDS): list of H_l x W_l x 4 arrays, level 0 first."""
    lib = load_host()
    info = (C.c_uint32 * 4)()
    levels = lib.pth_decode_image_levels(data, len(data))
    if not levels or lib.pth_decode_image(data, len(data), info, None, 0):
        raise PtxError(lib.pth_last_error().decode())
    dims = [(max(info[1] >> l, 1), max(info[0] >> l, 1)) for l in range(levels)]
    flat = np.empty(sum(h * w for h, w in dims) * 4, np.float32 if info[3] else np.uint8)
    if lib.pth_decode_image(data, len(data), info, flat.ctypes.data, flat.nbytes):
        raise PtxError(lib.pth_last_error().decode())
    out, at = [], 0
    for h, w in dims:
        out.append(flat[at:at + h * w * 4].reshape(h, w, 4))
        at += h * w * 4
    return out


def write_image(path: str, img: np.ndarray, fmt: int = OUTPUT_PNG):
    """OutputSaver::WriteImage: H x W x 4 uint8 (Png / Tga) or float32 (Hdr)."""
    img = np.ascontiguousarray(img)
    if load_host().pth_write_image(str(path).encode(), fmt, img.shape[1], img.shape[0], img.ctypes.data, img.nbytes):
        raise PtxError(f"pth_write_image({path}) failed")


def save_checkpoint(path: str, accum: np.ndarray, total_samples: int):
    accum = np.ascontiguousarray(accum, dtype=np.float32)
    if load_host().pth_save_checkpoint(str(path).encode(), accum.shape[1], accum.shape[0], total_samples, accum.ctypes.data):
        raise PtxError(f"pth_save_checkpoint({path}) failed")


def load_checkpoint(path: str):
    lib = load_host()
    w, h, n = C.c_uint32(), C.c_uint32(), C.c_uint32()
    if lib.pth_load_checkpoint(str(path).encode(), C.byref(w), C.byref(h), C.byref(n), None, 0):
        raise PtxError(f"pth_load_checkpoint({path}) failed")
    img = np.empty((h.value, w.value, 4), np.float32)
    if lib.pth_load_checkpoint(str(path).encode(), C.byref(w), C.byref(h), C.byref(n), img.ctypes.data, img.nbytes):
        raise PtxError(f"pth_load_checkpoint({path}) failed")
    return img, n.value


def owned_tiles(width: int, height: int, rank: int, world: int, tile: int = 32):
    """Tile ids (row-major) a rank owns under the round-robin pixel-tile shard."""
    tiles_x = (width + tile - 1) // tile
    tiles_y = (height + tile - 1) // tile
    return list(range(rank, tiles_x * tiles_y, world))


def shard_mask(width: int, height: int, rank: int, world: int, tile: int = 32) -> np.ndarray:
    """Boolean H x W mask of the pixels a rank owns."""
    tiles_x = (width + tile - 1) // tile
    ys, xs = np.mgrid[0:height, 0:width]
    tid = (ys // tile) * tiles_x + (xs // tile)
    return (tid % world) == rank

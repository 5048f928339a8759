"""Host-side mirror (Scene / SceneBuilder / Camera / ExampleScenes) and the C-ABI surface.
No GPU needed: nothing here launches a kernel."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import util


def test_struct_layouts(pkg):
    # the reference's PaddingTest.cpp checks host struct <-> std430; sizes here are also
    # static_assert-ed in include/ptx.h
    assert C.sizeof(pkg.RaygenUniformData) == 148
    assert C.sizeof(pkg.DirectionalLight) == 32 and C.sizeof(pkg.PointLight) == 48
    assert C.sizeof(pkg.LightsUbo) == 3120
    assert pkg.LightsUbo.Directional.offset == 16 and pkg.LightsUbo.Lights.offset == 48  # Renderer.h:152-156
    assert util.GEOMETRY_DT.itemsize == 20 and util.MESH_DT.itemsize == 12 and util.INSTANCE_DT.itemsize == 52


def test_headers_and_libraries_agree(pkg):
    """Both shared libraries load and export every symbol the public headers declare."""
    inc = os.path.join(pkg.REPO_DIR, "include")
    declared = set(re.findall(r"PTX_API[^;]*?\b(pt[xh]_\w+)\s*\(", open(os.path.join(inc, "ptx.h")).read()))
    declared_host = set(re.findall(r"PTX_API[^;]*?\b(pth_\w+)\s*\(", open(os.path.join(inc, "ptx_host.h")).read()))
    assert declared == set(pkg.PTX_SYMBOLS)
    assert declared_host == set(pkg.PTH_SYMBOLS)
    hip, host = pkg.load_hip(), pkg.load_host()
    for name in declared:
        assert hasattr(hip, name), name
    for name in declared_host:
        assert hasattr(host, name), name


def test_hardware_queue_default_is_the_hosts_not_the_librarys(pkg):
    """The HIP library leaves the process environment alone (a load-time setenv races with the host's threads and cannot
    know whether HIP is up already); the Python package -- the host, for Python callers -- sets GPU_MAX_HW_QUEUES=16 at
    import unless the caller has chosen a value; the library reports its ABI version."""
    import subprocess
    import sys

    lib_code = ("import ctypes, os, sys; lib = ctypes.CDLL(sys.argv[1]); libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; "
                "v = libc.getenv(b'GPU_MAX_HW_QUEUES'); lib.ptx_abi_version.restype = ctypes.c_uint32; "
                "print(v.decode() if v else 'unset', lib.ptx_abi_version())")
    pkg_code = "import os, sys; sys.path.insert(0, sys.argv[1]); import __graft_entry__ as g; g.load_package(); print(os.environ['GPU_MAX_HW_QUEUES'])"
    for preset in (None, "6"):
        env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
        if preset:
            env["GPU_MAX_HW_QUEUES"] = preset
        out = subprocess.run([sys.executable, "-c", lib_code, pkg.HIP_LIB], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert out.stdout.split() == [preset or "unset", str(pkg.ABI_VERSION)]
        out = subprocess.run([sys.executable, "-c", pkg_code, pkg.REPO_DIR], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert out.stdout.strip() == (preset or "16")
    header = open(os.path.join(pkg.REPO_DIR, "include", "ptx.h")).read()
    assert f"#define PTX_ABI_VERSION {pkg.ABI_VERSION}u" in header


def test_no_cpu_fallback(pkg):
    """Without a GPU the product refuses to run instead of falling back to a CPU path."""
    hip = pkg.load_hip()
    if hip.ptx_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(pkg.PtxError):
        pkg.Renderer()


def test_product_does_not_reference_oracle(pkg):
    for root, _, files in os.walk(pkg.PKG_DIR):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(root, f), errors="ignore").read()
                assert "pt_oracle" not in text and "oracle/" not in text.replace("/oracle and is test", ""), f


def test_default_scene_matches_reference_description(pkg):
    # ExampleScenes.cpp:320-545: 5 box quads + 2 cubes of 6 quads + light quad = 36 triangles,
    # 4 models, 4 instances, 10 materials, emissive-only lighting
    s = pkg.Scene("default")
    d = s.desc
    a = util.desc_arrays(d)
    assert s.triangle_count == 36
    assert d.modelCount == 4 and d.instanceCount == 4 and d.metallicRoughnessMaterialCount == 10
    assert d.geometryCount == 12 and d.meshCount == 18 and d.transformCount == 1
    assert list(a["models"]["MeshOffset"]) == [0, 5, 11, 17]  # running MeshOffset, Scene.cpp:337-355
    assert (a["transforms"][0] == np.float32([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0])).all()
    l = s.lights
    assert l.LightCount == 0 and list(l.Directional.Color) == [0, 0, 0] and list(l.Directional.Direction) == [0, -1, 0]
    # box instance: scale 2 then translate (-2.25, 0.5, 0) in the scaled frame
    box = a["instances"][0]["Transform"].reshape(3, 4)
    assert np.allclose(box, [[2, 0, 0, -4.5], [0, 2, 0, 1], [0, 0, 2, 0]])
    # light quad sits just under the ceiling of the box
    light = a["instances"][3]["Transform"].reshape(3, 4)
    assert np.allclose(light[:, 3], [-4.5, 1 + 2 * 1.099, 0], atol=1e-5)
    # indices are relative to the geometry's VertexOffset (Renderer.cpp:343-347)
    assert a["indices"].max() == 3


def test_camera_matrices(pkg):
    # InputCamera(45 deg, pos (3,1,0), dir (-1,0,0), up (0,-1,0)): Scene.h:259-260, Camera.cpp:62-76
    s = pkg.Scene("default")
    u = s.uniform(1920, 1080, bounces=4)
    vi = np.array(u.ViewInverse, np.float32).reshape(4, 4).T  # column-major -> math
    pi = np.array(u.ProjInverse, np.float32).reshape(4, 4).T
    assert np.allclose(vi[:3, 3], [3, 1, 0], atol=1e-6)
    fwd = vi[:3, :3] @ np.float32([0, 0, 1])
    assert np.allclose(fwd, [-1, 0, 0], atol=1e-6)
    up = vi[:3, :3] @ np.float32([0, 1, 0])
    assert np.allclose(up, [0, -1, 0], atol=1e-6)  # image row 0 looks towards world +y
    t = pi @ np.float32([1, 1, 1, 1])
    assert np.isclose(t[1] / t[2], np.tan(np.radians(22.5)), rtol=1e-5)
    assert np.isclose(t[0] / t[1], 1920 / 1080, rtol=1e-5)
    assert u.BounceCount == 4 and u.SampleCount == 1 and u.TotalSamples == 0


@pytest.mark.parametrize("name", ["roughness_cubes", "attenuation_blob", "chess_like", "temple_like", "atrium_like", "street_like"])
def test_scene_descs_are_consistent(pkg, name):
    s = pkg.Scene(name, 0.05)
    d = s.desc
    a = util.desc_arrays(d)
    assert d.instanceCount > 0 and s.triangle_count > 0
    assert (a["instances"]["ModelIndex"] < d.modelCount).all()
    assert ((a["models"]["MeshOffset"] + a["models"]["MeshCount"]) <= d.meshCount).all()
    assert (a["meshes"]["GeometryIndex"] < d.geometryCount).all() and (a["meshes"]["TransformIndex"] < d.transformCount).all()
    g = a["geometries"]
    assert ((g["VertexOffset"] + g["VertexLength"]) <= d.vertexCount).all()
    assert ((g["IndexOffset"] + g["IndexLength"]) <= d.indexCount).all() and (g["IndexLength"] % 3 == 0).all()
    v = a["vertices"]
    assert np.isfinite(v).all()
    for lo in (5, 8, 11):  # unit normal / tangent / bitangent
        assert np.abs(np.linalg.norm(v[:, lo:lo + 3], axis=1) - 1).max() < 1e-3
    assert np.abs((v[:, 5:8] * v[:, 8:11]).sum(axis=1)).max() < 1e-3  # N is not parallel to T
    assert s.lights.LightCount <= 64
    if name == "street_like":
        assert s.lights.LightCount == 64  # MaxLightCount cap (ShaderTypes.incl:30)


def test_full_detail_triangle_budgets(pkg):
    # SURVEY.md 8d sizes of the stand-ins at detail 1
    s = pkg.Scene("chess_like", 1.0)
    assert 1.9e6 < s.triangle_count < 2.1e6
    s = pkg.Scene("attenuation_blob", 1.0)
    assert 0.9e5 < s.triangle_count < 1.2e5


def test_material_id_packing(pkg):
    s = pkg.Scene("default")
    ids = util.desc_arrays(s.desc)["meshes"]["MaterialId"]
    assert ((ids & 0xFF) == 0).all() and ((ids >> 8) < 10).all()  # (index << 8) | type, ShaderTypes.incl:155-158


def test_shard_helpers(pkg):
    W, H, world = 200, 120, 3
    masks = [pkg.shard_mask(W, H, r, world, 32) for r in range(world)]
    total = sum(m.astype(int) for m in masks)
    assert (total == 1).all()  # every pixel is owned exactly once
    hip = pkg.load_hip()
    assert sorted(sum((pkg.owned_tiles(W, H, r, world) for r in range(world)), [])) == list(range(7 * 4))


def test_instrumentation_patch_applies_to_the_current_kernels(pkg):
    """tools/experiments/visit_stats.patch (the instrumentation tools/visit_histogram.py builds with) is kept out of the kernels
    but must keep applying to them."""
    import shutil
    import subprocess
    import tempfile

    patch = os.path.join(pkg.REPO_DIR, "tools", "experiments", "visit_stats.patch")
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copytree(os.path.join(pkg.PKG_DIR, "csrc"), os.path.join(tmp, "path-tracing_amd", "csrc"))
        p = subprocess.run(["patch", "-p1", "-s", "--dry-run", "-i", patch], cwd=tmp, capture_output=True, text=True)
        assert p.returncode == 0, p.stdout + p.stderr

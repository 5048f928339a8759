"""Row N1, stage 4: the skybox branches of miss.rmiss -- equirectangular 2-D lookup (atan / asin
texture coordinates, hdrToLdr) and the cube map (face selection of the Vulkan spec, layer order
+X -X +Y -Y +Z -Z).  The reference's sky images are downloaded assets; the stand-ins are procedural."""
import ctypes as C

import numpy as np
import pytest

import util


def _atan_asin_inputs():
    rng = np.random.default_rng(17)
    yx = rng.uniform(-2.0, 2.0, size=(8192, 2)).astype(np.float32)
    yx[:10] = [[0, 1], [0, -1], [1, 0], [-1, 0], [1, 1], [-1, -1], [1e-20, 1], [1, 1e-20], [0.5, -0.5], [0.25, -3]]
    return yx


def _empty_copy(desc):
    d = type(desc)()
    C.memmove(C.byref(d), C.byref(desc), C.sizeof(d))
    d.instanceCount = 0  # every path escapes: the image is the sky
    return d


def test_atan_asin_kernels_against_numpy(orc, pkg):
    yx = _atan_asin_inputs()
    out = orc.test_eval(pkg.FN["atanAsin"], yx, 2).view(np.float32).astype(np.float64)
    y, x = yx[:, 0].astype(np.float64), yx[:, 1].astype(np.float64)
    ref_atan = np.arctan2(y, x)
    assert np.abs(out[:, 0] - ref_atan).max() <= 2.5e-7  # half an ULP at pi
    assert np.abs(out[:, 1] - np.arcsin(np.clip(y, -1, 1))).max() <= 1.3e-7
    assert out[1, 0] == np.float32(np.pi) and out[2, 0] == np.float32(np.pi / 2) and out[0, 0] == 0.0


def test_oracle_equirect_and_cube_skies_agree(pkg, orc):
    """The 2-D and the cube sky are rasterised from ONE procedural direction -> colour function through
    the inverse of each lookup; sampled back through miss.rmiss they must agree up to texel filtering
    (2-D additionally goes through hdrToLdr, miss.rmiss:28)."""
    s2, sc = pkg.Scene("roughness_cubes"), pkg.Scene("reuse_mesh_cubes")
    assert s2.desc.skyboxKind == 1 and sc.desc.skyboxKind == 2
    W, H = 96, 54
    imgs = {}
    for look in ((0, 0.3, 1), (1, 0.6, 0.2), (0, -1, 0.05), (-1, 0.2, -0.4)):
        s2.set_camera_pose((0, 0, 0), look)
        u = s2.uniform(W, H, bounces=2, sample_count=1)
        a, sa = orc.OracleScene(_empty_copy(s2.desc)).render(u, s2.lights, W, H)
        b, _ = orc.OracleScene(_empty_copy(sc.desc)).render(u, sc.lights, W, H)
        assert sa.segments == W * H and sa.shadowRays == 0
        ldr = b[..., :3] / (1.0 + b[..., :3].max(axis=-1, keepdims=True))
        diff = np.abs(a[..., :3] - ldr)
        assert np.median(diff) < 2e-3 and diff.mean() < 0.01, (look, float(diff.mean()))
        imgs[look] = a
    # looking up: blue dominates; looking down: the grey-brown ground
    up, down = imgs[(0, 0.3, 1)][: H // 3].mean(axis=(0, 1)), imgs[(0, -1, 0.05)].mean(axis=(0, 1))
    assert up[2] > up[0] and abs(down[2] - down[0]) < 0.1


@pytest.mark.gpu
def test_sky_functions_match_oracle(pkg, orc, gpu_renderer):
    yx = _atan_asin_inputs()
    assert (gpu_renderer.test_eval(pkg.FN["atanAsin"], yx) == orc.test_eval(pkg.FN["atanAsin"], yx, 2)).all()
    rng = np.random.default_rng(4)
    d = rng.normal(size=(8192, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:6] = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]]
    assert (gpu_renderer.test_eval(pkg.FN["missSkyboxTexCoords"], d) == orc.test_eval(pkg.FN["missSkyboxTexCoords"], d, 2)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["roughness_cubes", "reuse_mesh_cubes"])
def test_sky_only_images_match_oracle(pkg, orc, name):
    scene = pkg.Scene(name)
    d = _empty_copy(scene.desc)
    W, H = 128, 72
    r = pkg.Renderer()
    r.upload(d)
    r.resize(W, H)
    osc = orc.OracleScene(d)
    for look in ((0.2, 0.4, 1), (1, -0.3, 0.1), (-0.5, 0.9, -0.3), (0, -1, 0.01)):
        scene.set_camera_pose((0, 1, 0), look)
        u = scene.uniform(W, H, bounces=3, sample_count=2)
        r.reset()
        r.render(u, scene.lights)
        ref, _ = osc.render(u, scene.lights, W, H)
        assert (r.readback().view(np.uint32) == ref.view(np.uint32)).all(), look
    r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("backend", [0, 1])
def test_reuse_mesh_cubes_image_matches_oracle(pkg, orc, backend):
    img, ref = util.render_pair(pkg, orc, "reuse_mesh_cubes", 1.0, 160, 90, frames=2, depth=6, backend=backend)
    differing = int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())
    assert differing == 0, f"{differing} pixels are not bit-identical (rel-L2 {util.rel_l2(img, ref)})"


class _TextureDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32), ("levels", C.c_uint32), ("data", C.c_void_p)]


_FACE_COLOURS = np.float32([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [0, 1, 1], [1, 0, 1]])  # +X -X +Y -Y +Z -Z


def _flat_cube_sky(scene, n=4):
    """The scene's description with no geometry and a cube sky of six constant n x n float faces."""
    faces = [np.broadcast_to(np.append(c, 1).astype(np.float32), (n, n, 4)).copy() for c in _FACE_COLOURS]
    arr = (_TextureDesc * 6)(*[_TextureDesc(n, n, 2, 1, f.ctypes.data) for f in faces])
    d = _empty_copy(scene.desc)
    d.skyboxKind = 2
    d.skybox = C.addressof(arr)
    return d, (arr, faces, scene)


def _look(scene, direction, W, H):
    scene.set_camera_pose((0, 0, 0), direction)
    return scene.uniform(W, H, bounces=1, sample_count=1)


def test_cube_sky_filters_across_face_edges(pkg, orc):
    """Vulkan cube maps are seamless: within half a texel of a face border the bilinear footprint continues on the
    neighbouring face, and at a corner of the cube the missing fourth texel is the mean of the other three."""
    s = pkg.Scene("reuse_mesh_cubes")
    d, keep = _flat_cube_sky(s)
    osc = orc.OracleScene(d)
    W, H = 97, 55
    # the +X / +Z edge: the centre column looks exactly along the border
    img, _ = osc.render(_look(s, (1, 0, 1), W, H), s.lights, W, H)
    row = img[H // 2, :, :3]
    px, pz = _FACE_COLOURS[0], _FACE_COLOURS[4]
    wx = row @ px / (px @ px)  # share of the +X colour along the row (the two colours are orthogonal)
    wz = row @ pz / (pz @ pz)
    assert np.allclose(wx + wz, 1, atol=1e-5) and abs(wx[W // 2] - 0.5) < 0.06
    mixed = (wx > 0.05) & (wx < 0.95)
    assert mixed.sum() >= 5, "a hard step between the faces: the footprint stopped at the border"
    assert (np.abs(np.diff(wx)) < 0.2).all(), "no jump along the row"
    assert {round(float(wx[0])), round(float(wx[-1]))} == {0, 1}
    # the +X +Y +Z corner: its centre pixel sees the three faces in equal parts
    img, _ = osc.render(_look(s, (1, 1, 1), W, H), s.lights, W, H)
    c = img[H // 2, W // 2, :3]
    assert np.allclose(c, (_FACE_COLOURS[0] + _FACE_COLOURS[2] + _FACE_COLOURS[4]) / 3, atol=0.05), c
    # far from any border nothing changes: a face's own colour
    img, _ = osc.render(_look(s, (0, -1, 0.001), W, H), s.lights, W, H)
    assert np.allclose(img[H // 2, W // 2, :3], _FACE_COLOURS[3], atol=1e-6)


@pytest.mark.gpu
def test_seamless_cube_sky_matches_oracle(pkg, orc, gpu_renderer):
    s = pkg.Scene("reuse_mesh_cubes")
    W, H = 160, 90
    for n in (1, 4, 7):
        d, keep = _flat_cube_sky(s, n)
        rng = np.random.default_rng(n)
        for f in keep[1]:  # texels that differ: every neighbour relation shows
            f[...] = rng.uniform(0, 2, f.shape).astype(np.float32)
        gpu_renderer.upload(d)
        gpu_renderer.resize(W, H)
        osc = orc.OracleScene(d)
        for look in ((1, 0, 1), (1, 1, 1), (-1, 1, -1), (0, 1, 0.01), (-1, -0.02, 1), (0.3, -1, -1)):
            u = _look(s, look, W, H)
            gpu_renderer.reset()
            gpu_renderer.render(u, s.lights)
            ref, _ = osc.render(u, s.lights, W, H)
            img = gpu_renderer.readback()
            assert (img.view(np.uint32) == ref.view(np.uint32)).all(), (n, look)

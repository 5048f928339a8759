"""sampleMaterial beyond MetallicRoughness (material.glsl:86-171, closestHit.rchit:101-102): SpecularGlossiness and Phong
materials with and without textures, the unknown-type default, and HasDxNormalTextures -- the branches the ORCA scenes
of BASELINE configs[2..4] take (ExampleScenes.cpp:93,96-131).  Function level: tests/golden (fn 34, from the
reference's own material.glsl text).  Here: the `materials_test` scene through oracle and HIP."""
import ctypes as C

import numpy as np
import pytest

import util

SG, PHONG = 1, 2


def _material_types(desc):
    a = util.desc_arrays(desc)
    return sorted(set(int(m) & 0xFF for m in a["meshes"]["MaterialId"]))


def test_scene_reaches_every_material_branch(pkg):
    s = pkg.Scene("materials_test", 0.5)
    d = s.desc
    assert d.dxNormalTextures == 1
    assert d.specularGlossinessMaterialCount == 3 and d.phongMaterialCount == 2 and d.metallicRoughnessMaterialCount >= 2
    assert _material_types(d) == [0, SG, PHONG, 7]  # 7: no such material type (material.glsl:161-165)
    # the SpecularGlossiness / Phong records use their own texture slots (Specular sRGB, Glossiness / Shininess alpha)
    sg = np.frombuffer((C.c_uint8 * (96 * d.specularGlossinessMaterialCount)).from_address(d.specularGlossinessMaterials), np.uint32).reshape(-1, 24)
    assert sg[0, 18] == 4 and (sg[0, 19:23] >= 9).all() and sg[1, 21] == 5 and sg[1, 22] == 6  # textured; defaults Specular = 5, Glossiness = 6


def _copy_desc(desc):
    d = type(desc)()
    C.memmove(C.byref(d), C.byref(desc), C.sizeof(d))
    return d


def test_oracle_branches_are_visible_in_the_image(pkg, orc):
    s = pkg.Scene("materials_test", 0.5)
    W, H = 160, 90
    u = s.uniform(W, H, bounces=3, sample_count=8)
    img, st = orc.OracleScene(s.desc).render(u, s.lights, W, H)
    assert np.isfinite(img).all() and st.retries == 0
    # the unknown-type sphere (x = +5.4, leftmost in the mirrored view) is emissive red whatever the light does
    ys, xs = np.mgrid[0:H, 0:W]
    red = (img[..., 0] > 4.0 * np.maximum(img[..., 1], img[..., 2])) & (img[..., 0] > 4.0)
    assert red.sum() > 40 and xs[red].mean() < W * 0.25
    # flipping the DirectX flag changes only what normal-mapped materials reflect
    d = _copy_desc(s.desc)
    d.dxNormalTextures = 0
    gl, _ = orc.OracleScene(d).render(u, s.lights, W, H)
    assert util.rel_l2(gl, img) > 1e-3
    # SpecularGlossiness / Phong records matter: replacing them by the MetallicRoughness default changes the image
    d2 = _copy_desc(s.desc)
    a = util.desc_arrays(s.desc)
    meshes = a["meshes"].copy()
    meshes["MaterialId"] = np.where((meshes["MaterialId"] & 0xFF) == 0, meshes["MaterialId"], 0)
    d2.meshes = meshes.ctypes.data
    mr, _ = orc.OracleScene(d2).render(u, s.lights, W, H)
    assert util.rel_l2(mr, img) > 0.05


def _np_material(kind, m, tex, inside):
    """material.glsl:86-142 in numpy float32 (same operation order), for the untextured records of the scene."""
    f = np.float32
    color = tex["color"][:3] * m["Color"][:3]
    specular = tex["a"][:3] * m["Specular"]
    gloss = f(tex["b"][3] * m["Gloss"])
    diff = np.maximum(specular - f(0.04), f(0)) / ((color - f(0.04)) + f(0.00001))
    return {"Color": color, "Roughness": f(1) - gloss, "Metalness": f(f(f(diff[0] + diff[1]) + diff[2]) / f(3)),
            "Eta": m["Ior"] if inside else f(1) / m["Ior"]}


def test_oracle_sample_material_matches_numpy_restatement(pkg, orc):
    rng = np.random.default_rng(5)
    n = 300
    inp = np.zeros((n, 47), np.float32)
    iu = inp.view(np.uint32)
    for i in range(n):
        kind = SG if i % 2 == 0 else PHONG
        iu[i, 0], iu[i, 1], iu[i, 2] = kind, i % 3 == 0, 0
        rec = rng.uniform(0.05, 1.0, 24).astype(np.float32)
        rec[17] = 1.0 + rec[17]  # Ior >= 1
        inp[i, 3:27] = rec
        inp[i, 27:47] = rng.uniform(0, 1, 20)
    out = orc.test_eval(pkg.FN["sampleMaterial"], inp, 17).view(np.float32)
    for i in range(n):
        rec, tx = inp[i, 3:27], inp[i, 27:47].reshape(5, 4)
        m = {"Color": rec[4:8], "Specular": rec[8:11], "Gloss": rec[11], "Ior": rec[16]}
        e = _np_material(0, m, {"color": tx[1], "a": tx[3], "b": tx[4]}, bool(iu[i, 1]))
        assert (out[i, 3:6] == e["Color"]).all() and out[i, 9] == e["Roughness"] and out[i, 12] == e["Eta"]
        assert abs(out[i, 10] - e["Metalness"]) <= 1e-6 * max(1.0, abs(e["Metalness"]))


# ---------------------------------------------------------------------------------------
# GPU: the HIP path against the oracle
# ---------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_sample_material_matches_oracle_bitexact(pkg, orc, gpu_renderer):
    rng = np.random.default_rng(11)
    n = 4000
    inp = np.zeros((n, 47), np.float32)
    iu = inp.view(np.uint32)
    iu[:, 0] = rng.choice([0, 1, 2, 3, 200], n)
    iu[:, 1] = rng.integers(0, 2, n)
    iu[:, 2] = rng.integers(0, 2, n)
    inp[:, 3:27] = rng.uniform(0.0, 1.5, (n, 24))
    inp[:, 27:47] = rng.uniform(0.0, 1.0, (n, 20))
    inp[::17, 27:47] = 0.0
    a = gpu_renderer.test_eval(pkg.FN["sampleMaterial"], inp)
    b = orc.test_eval(pkg.FN["sampleMaterial"], inp, 17)
    assert util.bits_equal_or_both_nan(a, b).all()


@pytest.mark.gpu
@pytest.mark.parametrize("backend", [0, 1])
def test_materials_scene_image_matches_oracle(pkg, orc, backend):
    img, ref = util.render_pair(pkg, orc, "materials_test", 0.5, 192, 108, frames=3, depth=6, backend=backend)
    assert np.isfinite(img).all()
    differing = int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())
    assert differing == 0, f"{differing} pixels are not bit-identical (rel-L2 {util.rel_l2(img, ref)})"


@pytest.mark.gpu
@pytest.mark.parametrize("sort", ["0", "1"])
def test_material_sorted_shade_queue_changes_nothing(pkg, orc, monkeypatch, sort):
    """k_shade's material-sorted queue (on by default for scenes that mix material types, PTX_SHADE_SORT overrides) only
    reorders which lane shades which path: image and counters equal the oracle's with and without it, above and below
    the k_tail threshold (PTX_TAIL_THRESHOLD=0: every bounce goes through k_shade)."""
    monkeypatch.setenv("PTX_SHADE_SORT", sort)
    monkeypatch.setenv("PTX_TAIL_THRESHOLD", "0")
    img, ref = util.render_pair(pkg, orc, "materials_test", 0.4, 200, 120, frames=2, depth=6)
    assert (img.view(np.uint32) == ref.view(np.uint32)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("tile", [8, 24])
def test_material_sorted_shade_queue_on_ragged_tiles(pkg, orc, monkeypatch, tile):
    """The sorted queue pads a block's share of the queue to kBlock x kShadeItems entries; padding and the dead slots of ragged
    edge tiles sort under the same (last) key.  An image whose size is not a multiple of the tile and a first-bounce queue that
    is not a multiple of 1024 entries: no slot may be shaded twice (padding used to read as slot 0), so image and counters
    still equal the oracle's."""
    monkeypatch.setenv("PTX_SHADE_SORT", "1")
    monkeypatch.setenv("PTX_TAIL_THRESHOLD", "0")
    img, ref = util.render_pair(pkg, orc, "materials_test", 0.4, 203, 117, frames=2, depth=6, tile=tile)
    assert (img.view(np.uint32) == ref.view(np.uint32)).all()


@pytest.mark.gpu
def test_materials_scene_multi_sample_and_lens(pkg, orc):
    img, ref = util.render_pair(pkg, orc, "materials_test", 0.3, 96, 54, frames=2, depth=5, lens=0.04, sample_count=3)
    assert (img.view(np.uint32) == ref.view(np.uint32)).all()


@pytest.mark.gpu
def test_dx_flag_reaches_the_device(pkg, orc):
    """HitFlagsDxNormalTextures is a pipeline specialisation in the reference (Renderer.cpp:676-709): here a field
    of the scene.  Same scene with the flag cleared: HIP == oracle again, and != the flagged image."""
    import torch  # noqa: F401

    s = pkg.Scene("materials_test", 0.3)
    W, H = 128, 72
    u = s.uniform(W, H, bounces=4)
    imgs = []
    for flag in (1, 0):
        d = _copy_desc(s.desc)
        d.dxNormalTextures = flag
        r = pkg.Renderer()
        r.upload(d)
        r.resize(W, H)
        r.render(u, s.lights)
        img = r.readback()
        r.close()
        ref, _ = orc.OracleScene(d).render(u, s.lights, W, H)
        assert (img.view(np.uint32) == ref.view(np.uint32)).all()
        imgs.append(img)
    assert util.rel_l2(imgs[0], imgs[1]) > 1e-3

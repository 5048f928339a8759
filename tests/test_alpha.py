"""Row N1, stage 3: the any-hit stages (anyhit.rahit, occlusionAnyhit.rahit) on non-opaque geometry:
alpha-tested candidates inside the traversal, decals, and shadow rays that pass through alpha < 1.
Scene `alpha_test` (path-tracing_amd/host/ExampleScenes.cpp): leaves (cut-out, soft rim), a decal panel
(alpha 0.3 / 0), a ghost pane (colour alpha 0.4) and a film (colour alpha 0.8) in front of a wall."""
import numpy as np
import pytest

import util

# geometry order inside the single model of alpha_test
FLOOR, WALL, DECAL, GHOST, FILM, LEAVES = range(6)


def _ray(o, d, tmax=1e4):
    d = np.asarray(d, np.float32)
    d = d / np.linalg.norm(d)
    return np.array([o[0], o[1], o[2], 1e-5, d[0], d[1], d[2], tmax], np.float32)


def _pair_of(desc, tri):
    first = util.pair_first(desc)
    return np.searchsorted(first, tri, side="right") - 1


def test_oracle_any_hit_semantics(pkg, orc):
    s = pkg.Scene("alpha_test")
    osc = orc.OracleScene(s.desc, build_bvh=True)
    a = util.desc_arrays(s.desc)
    assert [bool(g["IsOpaque"]) for g in a["geometries"]] == [True, True, False, False, False, False]
    rays = np.stack([
        _ray((1.5, 1.2, -4.0), (0, 0, 1)),   # through the ghost pane (alpha 0.4): ignored, the wall is hit
        _ray((3.4, 1.0, -4.0), (0, 0, 1)),   # the film (alpha 0.8): kept by closest rays
        _ray((2.0, 3.8, -4.0), (0, 0, 1)),   # the decal panel (alpha <= 0.3): ignored, the wall is hit
        _ray((4.8, 4.5, -4.0), (0, 0, 1)),   # bare wall
    ])
    for brute in (True, False):
        hit = osc.trace_closest(rays, brute_force=brute)
        pairs = _pair_of(s.desc, hit["tri"])
        assert list(pairs) == [WALL, FILM, WALL, WALL]
        occ = osc.trace_any(np.stack([_ray((1.5, 1.2, -4.0), (0, 0, 1), tmax=6.0),    # ghost only in range: passes
                                      _ray((3.4, 1.0, -4.0), (0, 0, 1), tmax=4.5),    # film only (alpha 0.8 < 1): passes
                                      _ray((4.8, 4.5, -4.0), (0, 0, 1), tmax=8.0)]),  # reaches the opaque wall: blocked
                            brute_force=brute)
        assert list(occ != 0) == [False, False, True]
    # BVH and brute force agree on random rays, and on the image (decal tie-breaks included)
    rng = np.random.default_rng(3)
    rr = util.random_rays(rng, 4000, -4.0, 4.0)
    rr[:, 1] = np.abs(rr[:, 1])
    x, y = osc.trace_closest(rr, brute_force=False), osc.trace_closest(rr, brute_force=True)
    assert (x["tri"] == y["tri"]).all() and (x["t"].view(np.uint32) == y["t"].view(np.uint32)).all()
    assert (osc.trace_any(rr, False) == osc.trace_any(rr, True)).all()
    W, H = 64, 36
    u = s.uniform(W, H, bounces=4, sample_count=2)
    img_bvh, sa = osc.render(u, s.lights, W, H)
    img_bf, sb = osc.render(u, s.lights, W, H, brute_force=True)
    assert (img_bvh.view(np.uint32) == img_bf.view(np.uint32)).all() and sa.shadowRays == sb.shadowRays


def test_oracle_decal_tints_and_film_casts_no_shadow(pkg, orc):
    import ctypes as C

    s = pkg.Scene("alpha_test")
    W, H = 96, 54
    u = s.uniform(W, H, bounces=1, sample_count=8)
    img, _ = orc.OracleScene(s.desc).render(u, s.lights, W, H)
    # the same scene with every geometry flagged opaque: the decal / ghost panes become solid surfaces
    d = type(s.desc)()
    C.memmove(C.byref(d), C.byref(s.desc), C.sizeof(d))
    geo = util.desc_arrays(s.desc)["geometries"].copy()
    geo["IsOpaque"] = 1
    d.geometries = geo.ctypes.data
    solid, _ = orc.OracleScene(d).render(u, s.lights, W, H)
    assert util.rel_l2(img, solid) > 0.05
    assert np.isfinite(img).all()


@pytest.mark.gpu
def test_alpha_traversal_matches_oracle(pkg, orc):
    scene = pkg.Scene("alpha_test")
    r = pkg.Renderer()
    r.upload(scene)
    osc = orc.OracleScene(scene.desc, build_bvh=False)
    rng = np.random.default_rng(21)
    rays = util.random_rays(rng, 30000, -4.0, 4.0)
    rays[:, 1] = np.abs(rays[:, 1])
    hits, ids = r.trace_rays(rays, any_hit=False)
    ref = osc.trace_closest(rays, brute_force=True)
    first = util.pair_first(scene.desc)
    miss = ids[:, 0] == 0xFFFFFFFF
    gid = np.where(miss, 0xFFFFFFFF, first[np.minimum(ids[:, 0], len(first) - 2)] + ids[:, 1]).astype(np.uint32)
    assert (gid == ref["tri"]).all(), f"{int((gid != ref['tri']).sum())} rays hit a different triangle"
    h = ~miss
    assert (ids[h, 0] == LEAVES).sum() > 500 and (ids[h, 0] == WALL).sum() > 500
    for k, f in enumerate(("t", "u", "v")):
        assert (hits[h, k].view(np.uint32) == ref[f][h].view(np.uint32)).all(), f
    occ_hits, _ = r.trace_rays(rays, any_hit=True)
    occ_ref = osc.trace_any(rays, brute_force=True)
    assert ((occ_hits[:, 3] != 0) == (occ_ref != 0)).all()
    r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("backend", [0, 1])
def test_alpha_scene_image_matches_oracle(pkg, orc, backend):
    img, ref = util.render_pair(pkg, orc, "alpha_test", 1.0, 160, 90, frames=2, depth=6, backend=backend)
    assert np.isfinite(img).all()
    differing = int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())
    assert differing == 0, f"{differing} pixels are not bit-identical (rel-L2 {util.rel_l2(img, ref)})"


@pytest.mark.gpu
def test_atrium_stand_in_image_matches_oracle(pkg, orc):
    # the Sponza stand-in of BASELINE configs[3] (north_star's 10x-CPU / 1e-3 target scene): textures, alpha-tested ivy
    # cards and curtains, i.e. kernel mode 2, at a size the oracle renders in seconds
    img, ref = util.render_pair(pkg, orc, "atrium_like", 0.04, 128, 72, frames=2, depth=8)
    assert np.isfinite(img).all()
    assert ref[..., :3].max() > 0
    differing = int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())
    assert differing == 0, f"{differing} pixels are not bit-identical (rel-L2 {util.rel_l2(img, ref)})"


@pytest.mark.gpu
def test_alpha_scene_tail_multi_sample_and_lens(pkg, orc, monkeypatch):
    img, ref = util.render_pair(pkg, orc, "alpha_test", 1.0, 96, 54, frames=2, depth=5, lens=0.04, sample_count=2)
    assert (img.view(np.uint32) == ref.view(np.uint32)).all()
    scene = pkg.Scene("alpha_test")
    W, H = 128, 72
    imgs = []
    for thr in ("0", "100000000"):
        monkeypatch.setenv("PTX_TAIL_THRESHOLD", thr)
        r = pkg.Renderer()
        r.upload(scene)
        r.resize(W, H)
        r.render_frames(scene.uniform(W, H, bounces=6), scene.lights, 0, 3)
        imgs.append(r.readback())
        r.close()
    assert (imgs[0].view(np.uint32) == imgs[1].view(np.uint32)).all()


@pytest.mark.gpu
def test_alpha_textures_of_odd_shapes_match_oracle(pkg, orc):
    """The any-hit record carries the alpha texture's extent in 15 + 15 bits beside the triangle and reads 2 x 2 footprints
    ("quads") with repeat addressing: the same scene with its three colour textures replaced by a 37 x 21, a 1 x 1 (the
    constant shortcut), and a 300 x 2 image of random alphas around both thresholds -- traversal decisions (closest and
    shadow queries, ids and t, u, v bit for bit against brute force) and a rendered frame against the oracle."""
    import torch  # noqa: F401

    from test_textures import SRGB, _desc_with

    scene = pkg.Scene("alpha_test")
    rng = np.random.default_rng(77)
    shapes = [(37, 21), (1, 1), (300, 2)]
    texels = []
    for w, h in shapes:
        px = rng.integers(0, 256, (h, w, 4)).astype(np.uint8)
        px[..., 3] = rng.choice([0, 90, 127, 128, 200, 254, 255], (h, w))  # 127 / 128 straddle 0.5; 254 / 255 straddle 1
        texels.append(px.reshape(-1))
    texels[1][3] = 255
    d, keep = _desc_with(pkg, scene, [(w, h, SRGB, 1, t) for (w, h), t in zip(shapes, texels)], budget=2**64 - 1)
    r = pkg.Renderer()
    r.upload(d)
    osc = orc.OracleScene(d, build_bvh=False)
    rays = util.random_rays(rng, 30000, -4.0, 4.0)
    rays[:, 1] = np.abs(rays[:, 1])
    hits, ids = r.trace_rays(rays, any_hit=False)
    ref = osc.trace_closest(rays, brute_force=True)
    first = util.pair_first(d)
    miss = ids[:, 0] == 0xFFFFFFFF
    gid = np.where(miss, 0xFFFFFFFF, first[np.minimum(ids[:, 0], len(first) - 2)] + ids[:, 1]).astype(np.uint32)
    assert (gid == ref["tri"]).all(), f"{int((gid != ref['tri']).sum())} rays hit a different triangle"
    h = ~miss
    for k, f in enumerate(("t", "u", "v")):
        assert (hits[h, k].view(np.uint32) == ref[f][h].view(np.uint32)).all(), f
    occ_hits, _ = r.trace_rays(rays, any_hit=True)
    assert ((occ_hits[:, 3] != 0) == (osc.trace_any(rays, brute_force=True) != 0)).all()
    W, H = 128, 72
    u = scene.uniform(W, H, bounces=5)
    r.resize(W, H)
    r.render_frames(u, scene.lights, 0, 2)
    img = r.readback()
    osc2 = orc.OracleScene(d)
    acc = np.zeros((H, W, 4), np.float32)
    for f in range(2):
        osc2.render(scene.uniform(W, H, bounces=5, total_samples=f), scene.lights, W, H, accum=acc)
    assert (img.view(np.uint32) == acc.view(np.uint32)).all()
    r.close()

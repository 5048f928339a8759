"""Self-consistency of the oracle's render path (CPU only): its SAH BVH against brute force,
shard composition, determinism, retry bookkeeping."""
import numpy as np
import pytest

import util


@pytest.mark.parametrize("name,detail", [("default", 1.0), ("chess_like", 0.03), ("attenuation_blob", 0.05)])
def test_bvh_equals_bruteforce(pkg, orc, name, detail):
    s = pkg.Scene(name, detail)
    osc = orc.OracleScene(s.desc, build_bvh=True)
    rng = np.random.default_rng(5)
    rays = util.random_rays(rng, 4000, -6, 6)
    a = osc.trace_closest(rays, brute_force=False)
    b = osc.trace_closest(rays, brute_force=True)
    assert (a["tri"] == b["tri"]).all()
    h = b["tri"] != 0xFFFFFFFF
    assert h.sum() > 100
    for f in ("t", "u", "v"):
        assert (a[f][h].view(np.uint32) == b[f][h].view(np.uint32)).all()
    assert (osc.trace_any(rays, False) == osc.trace_any(rays, True)).all()


def test_bvh_equal_t_tie_behind_its_own_box(pkg, orc):
    """Known-answer regression: the primary ray of pixel (303, 666) of the 1920x1080 street_like frame crosses two
    overlapping coplanar triangles that report the bit-identical t, while o + t d lies 2.6e-6 outside the padded box of
    the one with the smaller id (Moeller-Trumbore from 12.5 units away).  The walk must still visit it for the id
    tie-break, i.e. agree with brute force."""
    s = pkg.Scene("street_like", 1.0)
    osc = orc.OracleScene(s.desc, build_bvh=True)
    o = np.array([0xC2080000, 0x3FD9999C, 0x3F800000], np.uint32).view(np.float32)
    d = np.array([0x3F4C1984, 0xBD9D0A84, 0xBF194709], np.uint32).view(np.float32)
    ray = np.array([[o[0], o[1], o[2], 1e-5, d[0], d[1], d[2], 1e4]], np.float32)
    a = osc.trace_closest(ray, brute_force=False)
    b = osc.trace_closest(ray, brute_force=True)
    assert b["tri"][0] == 2153290 and b["t"].view(np.uint32)[0] == 0x4148FDE3
    assert a["tri"][0] == b["tri"][0] and a["t"][0] == b["t"][0]


def test_render_bvh_equals_bruteforce_and_is_deterministic(pkg, orc):
    s = pkg.Scene("default")
    osc = orc.OracleScene(s.desc, build_bvh=True)
    W, H = 64, 36
    u = s.uniform(W, H, bounces=4, total_samples=2)
    a, sa = osc.render(u, s.lights, W, H, brute_force=False)
    b, sb = osc.render(u, s.lights, W, H, brute_force=True, threads=2)
    assert (a.view(np.uint32) == b.view(np.uint32)).all()
    assert sa.segments == sb.segments and sa.shadowRays == sb.shadowRays and sa.pathSamples == W * H + sa.retries
    assert np.isfinite(a).all() and (a[..., 3] == 1).all()


def test_accumulation_is_a_running_sum(pkg, orc):
    s = pkg.Scene("chess_like", 0.03)
    osc = orc.OracleScene(s.desc)
    W, H = 48, 27
    acc = np.zeros((H, W, 4), np.float32)
    parts = []
    for f in range(3):
        u = s.uniform(W, H, bounces=5, total_samples=f)
        osc.render(u, s.lights, W, H, accum=acc)
        one, _ = osc.render(u, s.lights, W, H)
        parts.append(one)
    expect = (parts[0][..., :3] + parts[1][..., :3]) + parts[2][..., :3]
    assert (acc[..., :3].view(np.uint32) == expect.view(np.uint32)).all()
    assert not (parts[0] == parts[1]).all()  # the RNG frame changes the image


def test_tile_shards_compose(pkg, orc):
    s = pkg.Scene("default")
    osc = orc.OracleScene(s.desc)
    W, H, world = 100, 70, 3
    u = s.uniform(W, H, bounces=4)
    full, _ = osc.render(u, s.lights, W, H)
    out = np.zeros_like(full)
    for r in range(world):
        shard = pkg.TileShard(r, world, 32)
        part, _ = osc.render(u, s.lights, W, H, shard=shard)
        mask = pkg.shard_mask(W, H, r, world, 32)
        assert (part[~mask] == 0).all()
        out[mask] = part[mask]
    assert (out.view(np.uint32) == full.view(np.uint32)).all()


def test_empty_and_degenerate_inputs(pkg, orc):
    import ctypes as C
    d = pkg.SceneDesc()
    ident = (C.c_float * 12)(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0)
    d.transforms = C.addressof(ident)
    d.transformCount = 1
    osc = orc.OracleScene(d)
    assert osc.triangle_count == 0
    s = pkg.Scene("default")
    img, st = osc.render(s.uniform(16, 9), s.lights, 16, 9)
    assert np.allclose(img[..., :3], np.float32([0.08, 0.09, 0.1]))  # miss.rmiss:37 constant sky
    assert st.segments == 16 * 9 and st.shadowRays == 0
    # zero bounces: nothing is traced, radiance stays 0 (raygen.rgen:62)
    osc2 = orc.OracleScene(s.desc)
    img, st = osc2.render(s.uniform(16, 9, bounces=0), s.lights, 16, 9)
    assert (img[..., :3] == 0).all() and st.segments == 0

"""Row N1, stage 1: the software sampler (textureGrad / texture()), mip generation and texture
formats.  The Vulkan sampler's filtering is implementation-defined, so parity here is HIP vs the
oracle (bit-exact, same fixed arithmetic) plus analytic properties of the oracle itself."""
import numpy as np
import pytest

import util

CHECKER, NOISE, ROUGH, METAL, BUMP, GLOW, RAMP = range(9, 16)


def _inputs(idx, u, v, dudx=0.0, dvdx=0.0, dudy=0.0, dvdy=0.0):
    a = np.zeros((len(u), 7), np.float32)
    a.view(np.uint32)[:, 0] = idx
    a[:, 1], a[:, 2], a[:, 3], a[:, 4], a[:, 5], a[:, 6] = u, v, dudx, dvdx, dudy, dvdy
    return a


def _srgb_to_linear(c8):
    c = np.asarray(c8, np.float64) / 255.0
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


def test_texture_descs_follow_reference_format_rule(pkg):
    import ctypes as C

    class TextureDesc(C.Structure):
        _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32), ("levels", C.c_uint32), ("data", C.c_void_p)]

    s = pkg.Scene("texture_test")
    d = s.desc
    assert d.textureCount == 7
    t = (TextureDesc * d.textureCount).from_address(d.textures)
    # TextureUploader.cpp:571-594: Color / Emissive -> sRGB, Roughness / Metallic / Normal -> UNORM, RGBAF32 as is
    assert [x.format for x in t] == [1, 1, 0, 0, 0, 1, 2]
    assert (t[1].width, t[1].height) == (37, 21) and (t[2].width, t[2].height) == (128, 8)
    # a scene whose texel data is unavailable still presents 1x1 placeholders (default scene's embedded PNGs)
    s0 = pkg.Scene("default")  # keep alive: the desc points into it
    d0 = s0.desc
    t0 = (TextureDesc * d0.textureCount).from_address(d0.textures)
    assert d0.textureCount == 4 and all((x.width, x.height) == (1, 1) for x in t0)


def test_oracle_sampler_properties(pkg, orc):
    s = pkg.Scene("texture_test")
    osc = orc.OracleScene(s.desc, build_bvh=False)
    # texel centres of the checker at LOD 0: exactly the decoded sRGB texel
    xs = np.array([0, 7, 8, 15, 63], np.float32)
    u = (xs + 0.5) / 64
    out = osc.test_texture(_inputs(CHECKER, u, np.full_like(u, 0.5 / 64))).view(np.float32)
    on = ((xs // 8).astype(int) + 0) & 1
    expect = np.where(on[:, None] == 1, _srgb_to_linear([230, 200, 60]), _srgb_to_linear([30, 40, 150]))
    assert np.abs(out[:, :3] - expect).max() < 2e-7 and (out[:, 3] == 1).all()
    # repeat addressing: whole-number shifts of uv hit the same texel
    a = osc.test_texture(_inputs(CHECKER, u, u))
    b = osc.test_texture(_inputs(CHECKER, u + 1, u - 3))
    assert (a == b).all()
    # texture() at implicit LOD 0 == textureGrad with zero gradients
    rng = np.random.default_rng(2)
    uv = rng.uniform(-2, 3, (500, 2)).astype(np.float32)
    for idx in (CHECKER, NOISE, ROUGH, RAMP):
        g = osc.test_texture(_inputs(idx, uv[:, 0], uv[:, 1]))
        t = osc.test_texture(_inputs(idx, uv[:, 0], uv[:, 1]), implicit_lod=True)
        assert (g == t).all()
    # bilinear results stay inside the range of the texture and the filter is continuous
    n = osc.test_texture(_inputs(NOISE, uv[:, 0], uv[:, 1])).view(np.float32)
    assert (n >= 0).all() and (n <= 1).all()
    # huge footprint -> the 1x1 top level = (approximately) the average colour of the checker
    top = osc.test_texture(_inputs(CHECKER, u, u, dudx=8.0, dvdy=8.0)).view(np.float32)
    mean = 0.5 * (_srgb_to_linear([230, 200, 60]) + _srgb_to_linear([30, 40, 150]))
    assert np.abs(top[:, :3] - mean).max() < 0.02
    # trilinear: the LOD grows monotonically with the footprint (less contrast), NaN / inf gradients are safe
    c = [osc.test_texture(_inputs(CHECKER, u[:1], u[:1], dudx=g, dvdy=g)).view(np.float32)[0, 0] for g in (1e-4, 0.05, 0.2, 1.0)]
    assert c[0] <= c[1] <= c[2] <= c[3] + 1e-6  # texel (0,0) is the dark colour: blends towards the mean
    bad = osc.test_texture(_inputs(CHECKER, u, u, dudx=np.nan, dvdy=np.inf)).view(np.float32)
    assert np.isfinite(bad).all()
    # anisotropy (the reference's sampler enables it at the device maximum): a footprint 1 texel wide and 8 long is the
    # average of N = 8 trilinear taps at LOD log2(8 / 8) = 0, spaced along the long gradient at i / (N + 1) - 1/2
    uu, vv = np.float32([0.3, 0.61, 0.87]), np.float32([0.2, 0.45, 0.7])
    for idx, (tw, th) in ((CHECKER, (64, 64)), (NOISE, (37, 21))):
        thin = osc.test_texture(_inputs(idx, uu, vv, dudx=1.0 / tw, dvdy=8.0 / th)).view(np.float32)
        taps = [osc.test_texture(_inputs(idx, uu, vv + np.float32(8.0 / th) * np.float32(i / 9.0 - 0.5))).view(np.float32) for i in range(1, 9)]
        assert np.abs(thin - np.mean(taps, axis=0)).max() < 2e-6
    # ... which keeps the resolution across the short side where an isotropic 8 x 8 footprint does not
    u2 = np.float32(np.arange(40) / 40.0)
    thin = osc.test_texture(_inputs(NOISE, u2, np.full_like(u2, 0.5), dudx=1.0 / 37, dvdy=8.0 / 21)).view(np.float32)
    wide = osc.test_texture(_inputs(NOISE, u2, np.full_like(u2, 0.5), dudx=8.0 / 37, dvdy=8.0 / 21)).view(np.float32)
    assert thin[:, 0].std() > 1.5 * wide[:, 0].std()
    # the float texture passes through untouched at texel centres
    r = osc.test_texture(_inputs(RAMP, np.float32([(3 + 0.5) / 16]), np.float32([(2 + 0.5) / 4]))).view(np.float32)[0]
    assert np.allclose(r, [0.05 + 0.06 * 3, 0.2 + 0.2 * 2, 0.9 - 0.05 * 3, 1.0], atol=1e-6)
    # indices outside the table (and the fixed slots) are the white placeholder here
    w = osc.test_texture(_inputs(200, u, u)).view(np.float32)
    assert (w == 1).all()


@pytest.mark.gpu
def test_sampler_matches_oracle_bitexact(pkg, orc, gpu_renderer):
    s = pkg.Scene("texture_test")
    gpu_renderer.upload(s)
    osc = orc.OracleScene(s.desc, build_bvh=False)
    rng = np.random.default_rng(5)
    n = 20000
    idx = rng.integers(9, 17, n).astype(np.uint32)  # includes one index past the table
    uv = rng.uniform(-2.5, 3.5, (n, 2)).astype(np.float32)
    g = (10.0 ** rng.uniform(-5, 0.5, (n, 4)) * rng.choice([-1, 1], (n, 4))).astype(np.float32)
    g[:50] = 0
    g[50:60, 0] = np.nan
    g[60:70, 3] = np.inf
    uv[70:80, 0] = np.nan
    uv[80:90, 1] = 1e30
    inp = _inputs(idx, uv[:, 0], uv[:, 1], g[:, 0], g[:, 1], g[:, 2], g[:, 3])
    for implicit in (False, True):
        a = gpu_renderer.test_texture(inp, implicit)
        b = osc.test_texture(inp, implicit)
        assert (a == b).all(), f"{int((a != b).any(axis=1).sum())} samples differ (implicit={implicit})"


# ---------------------------------------------------------------------------------------
# stage 2: ray differentials through the path, textureGrad in the closest-hit stage
# ---------------------------------------------------------------------------------------
def test_oracle_textures_change_the_image(pkg, orc):
    import ctypes as C

    s = pkg.Scene("texture_test")
    W, H = 96, 54
    u = s.uniform(W, H, bounces=2, sample_count=4)
    textured, _ = orc.OracleScene(s.desc).render(u, s.lights, W, H)
    d = type(s.desc)()
    C.memmove(C.byref(d), C.byref(s.desc), C.sizeof(d))
    d.textureCount = 0  # same geometry, every index >= 9 now samples the white placeholder
    plain, _ = orc.OracleScene(d).render(u, s.lights, W, H)
    assert np.isfinite(textured).all()
    assert util.rel_l2(textured, plain) > 0.1
    # the near floor rows show both checker colours: albedo estimate = textured / plain (same RNG streams)
    ratio = textured[H - 3, 8:-8, 2] / np.maximum(plain[H - 3, 8:-8, 2], 1e-6)
    assert ratio.max() > 3.0 * ratio.min()


@pytest.mark.gpu
@pytest.mark.parametrize("backend", [0, 1])
def test_textured_scene_image_matches_oracle(pkg, orc, backend):
    img, ref = util.render_pair(pkg, orc, "texture_test", 1.0, 160, 90, frames=2, depth=6, backend=backend)
    assert np.isfinite(img).all()
    differing = int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())
    assert differing == 0, f"{differing} pixels are not bit-identical (rel-L2 {util.rel_l2(img, ref)})"


@pytest.mark.gpu
def test_textured_scene_thin_lens_and_multi_sample(pkg, orc):
    # thin-lens offset rays (ray.glsl:16-56) and differentials restarted per sample (raygen.rgen:56-58)
    img, ref = util.render_pair(pkg, orc, "texture_test", 1.0, 96, 54, frames=2, depth=4, lens=0.05, sample_count=3)
    assert (img.view(np.uint32) == ref.view(np.uint32)).all()


@pytest.mark.gpu
def test_textured_tail_and_wavefront_agree(pkg, monkeypatch):
    # k_tail picks the differentials up from the slot state: force it on / off and compare
    scene = pkg.Scene("texture_test")
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


# ---------------------------------------------------------------------------------------
# upload rules of TextureUploader::UploadTexture: file-supplied mip chains and the memory budget
# ---------------------------------------------------------------------------------------
import ctypes as C  # noqa: E402


class _TextureDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32), ("levels", C.c_uint32), ("data", C.c_void_p)]


UNORM, SRGB, F32 = 0, 1, 2


def _chain(w, h, texel_of_level, dtype=np.uint8):
    """All levels of a w x h image, level 0 first: texel_of_level(l, ys, xs) -> (h_l, w_l, 4)."""
    out = []
    l = 0
    while True:
        lw, lh = max(w >> l, 1), max(h >> l, 1)
        ys, xs = np.mgrid[0:lh, 0:lw]
        out.append(np.ascontiguousarray(texel_of_level(l, ys, xs), dtype).reshape(-1))
        if lw == 1 and lh == 1:
            return out
        l += 1


def _desc_with(pkg, scene, textures, budget=0, force_full=False):
    """A copy of the scene's description whose texture table is `textures` = [(w, h, format, levels, flat array)]."""
    arr = (_TextureDesc * len(textures))()
    for i, (w, h, fmt, levels, data) in enumerate(textures):
        arr[i] = _TextureDesc(w, h, fmt, levels, data.ctypes.data)
    d = type(scene.desc).from_buffer_copy(scene.desc)
    d.textures = C.addressof(arr)
    d.textureCount = len(textures)
    d.textureMemoryBudget = budget
    d.forceFullTextureSize = 1 if force_full else 0
    return d, (arr, textures, scene)  # keep alive


def _upload_rule_textures():
    rng = np.random.default_rng(11)
    flat = lambda l, ys, xs: np.broadcast_to(np.uint8([20 + 30 * l, 200 - 25 * l, 7 * l, 255]), ys.shape + (4,))  # noqa: E731
    blocks = lambda l, ys, xs: np.stack([((xs // 2) * 7 + (ys // 2) * 3) % 256, (xs // 2) * 4, (ys // 2) * 4, np.full_like(xs, 255)], -1)  # noqa: E731
    noise37 = rng.integers(0, 256, (21, 37, 4)).astype(np.uint8)
    ramp = rng.uniform(0, 4, (16, 64, 4)).astype(np.float32)
    return [
        (8, 8, UNORM, 4, np.concatenate(_chain(8, 8, flat))),           # 0: complete file chain
        (8, 8, UNORM, 2, np.concatenate(_chain(8, 8, flat)[:2])),       # 1: incomplete chain -> generated from level 0
        (64, 64, UNORM, 1, _chain(64, 64, blocks)[0]),                  # 2: level 0 only, constant 2 x 2 blocks
        (64, 64, SRGB, 7, np.concatenate(_chain(64, 64, flat))),        # 3: complete chain, scaled by skipping levels
        (37, 21, SRGB, 1, noise37.reshape(-1)),                         # 4: odd size: halve once, then a 18 x 10 -> 12 x 7 blit
        (64, 16, F32, 1, ramp.reshape(-1)),                             # 5: float pool
    ]


# the per-texture share that makes 32 x 32 the largest RGBA8 extent (chain of 32: 5,460 B; of 64: 21,844 B) and, for the
# float pool, 16 x 16 (5,456 B; chain of 32: 21,840 B)
_BUDGET_32 = 6 * 6000


def test_oracle_upload_rules_file_chain_and_budget(pkg, orc):
    s = pkg.Scene("texture_test")
    tex = _upload_rule_textures()
    lod = lambda k: dict(dudx=np.float32(2.0 ** k / 8), dvdy=np.float32(2.0 ** k / 8))  # noqa: E731  (8 x 8 texture: LOD k)
    u = np.float32([0.3, 0.7])
    # no limit: a complete file chain is used level by level, an incomplete one is replaced by blits of level 0
    d, keep = _desc_with(pkg, s, tex, budget=2**64 - 1)
    osc = orc.OracleScene(d, build_bvh=False)
    for k in range(4):
        a = osc.test_texture(_inputs(9, u, u, **lod(k))).view(np.float32)
        assert np.allclose(a[:, :3], np.float32([20 + 30 * k, 200 - 25 * k, 7 * k]) / 255, atol=1e-7), k
        b = osc.test_texture(_inputs(10, u, u, **lod(k))).view(np.float32)
        assert np.allclose(b[:, :3], np.float32([20, 200, 0]) / 255, atol=1e-6), k
    full = osc.test_texture(_inputs(11, (np.arange(64, dtype=np.float32) + 0.5) / 64, np.full(64, 0.5 / 64, np.float32)), implicit_lod=True).view(np.float32)
    assert np.allclose(full[:, 0] * 255, ((np.arange(64) // 2) * 7) % 256, atol=1e-4)
    # budget: 64 x 64 becomes 32 x 32 -- one halving blit of level 0 (constant 2 x 2 blocks survive it exactly) ...
    d, keep2 = _desc_with(pkg, s, tex, budget=_BUDGET_32)
    osc = orc.OracleScene(d, build_bvh=False)
    xs = np.arange(32, dtype=np.float32)
    got = osc.test_texture(_inputs(11, (xs + 0.5) / 32, np.full(32, 4.5 / 32, np.float32)), implicit_lod=True).view(np.float32)
    assert np.allclose(got[:, 0] * 255, (np.arange(32) * 7 + 4 * 3) % 256, atol=1e-4) and np.allclose(got[:, 1] * 255, np.arange(32) * 4, atol=1e-4)
    # ... a file with its own chain gives up its first level instead (level k of the image = level k + 1 of the file) ...
    for k in range(3):
        g = dict(dudx=np.float32(2.0 ** k / 32), dvdy=np.float32(2.0 ** k / 32))
        a = osc.test_texture(_inputs(12, u, u, **g)).view(np.float32)
        assert np.allclose(a[:, :3], _srgb_to_linear([20 + 30 * (k + 1), 200 - 25 * (k + 1), 7 * (k + 1)]), atol=1e-6), k
    # ... small textures are untouched, the float pool has its own limit (64 x 16 -> 16 x 4), forceFullTextureSize switches it off
    a = osc.test_texture(_inputs(9, u, u, **lod(2))).view(np.float32)
    assert np.allclose(a[:, :3], np.float32([80, 150, 14]) / 255, atol=1e-7)
    ramp = tex[5][4].reshape(16, 64, 4)
    f = osc.test_texture(_inputs(14, np.float32([(3 + 0.5) / 16]), np.float32([(1 + 0.5) / 4])), implicit_lod=True).view(np.float32)[0]
    assert np.allclose(f, ramp[4:8, 12:16].mean(axis=(0, 1)), rtol=2e-6)
    d, keep3 = _desc_with(pkg, s, tex, budget=_BUDGET_32, force_full=True)
    osc = orc.OracleScene(d, build_bvh=False)
    again = osc.test_texture(_inputs(11, (np.arange(64, dtype=np.float32) + 0.5) / 64, np.full(64, 0.5 / 64, np.float32)), implicit_lod=True)
    assert (again.view(np.float32) == full).all()


@pytest.mark.gpu
def test_upload_rules_match_oracle_bitexact(pkg, orc, gpu_renderer):
    s = pkg.Scene("texture_test")
    tex = _upload_rule_textures()
    rng = np.random.default_rng(6)
    n = 20000
    idx = rng.integers(9, 16, n).astype(np.uint32)
    uv = rng.uniform(-1.5, 2.5, (n, 2)).astype(np.float32)
    g = (10.0 ** rng.uniform(-4, 0.3, (n, 4)) * rng.choice([-1, 1], (n, 4))).astype(np.float32)
    inp = _inputs(idx, uv[:, 0], uv[:, 1], g[:, 0], g[:, 1], g[:, 2], g[:, 3])
    for budget, force in ((2**64 - 1, False), (_BUDGET_32, False), (_BUDGET_32 // 5, False), (_BUDGET_32, True), (0, False)):
        d, keep = _desc_with(pkg, s, tex, budget=budget, force_full=force)
        gpu_renderer.upload(d)
        osc = orc.OracleScene(d, build_bvh=False)
        for implicit in (False, True):
            a, b = gpu_renderer.test_texture(inp, implicit), osc.test_texture(inp, implicit)
            assert (a == b).all(), f"{int((a != b).any(axis=1).sum())} samples differ (budget {budget}, force {force}, implicit {implicit})"
    # and through the path: the scene rendered with its textures squeezed to a quarter
    d = type(s.desc).from_buffer_copy(s.desc)
    d.textureMemoryBudget = 7 * 1500
    gpu_renderer.upload(d)
    gpu_renderer.resize(96, 64)
    gpu_renderer.reset()
    u = s.uniform(96, 64, bounces=3)
    gpu_renderer.render(u, s.lights)
    ref, _ = orc.OracleScene(d).render(u, s.lights, 96, 64)
    full, _ = orc.OracleScene(s.desc).render(u, s.lights, 96, 64)
    img = gpu_renderer.readback()
    assert (img.view(np.uint32) == ref.view(np.uint32)).all() and not (ref == full).all()


def test_isotropic_texture_grad_against_float64_formulas(pkg, orc):
    """An independent restatement of the sampler for the isotropic case, in float64 from the Vulkan specification's
    formulas (texel coordinates u * w - 1/2 with repeat addressing, bilinear weights, lambda = log2(rho), linear blend of
    the two nearest levels of a 2 x 2-mean chain), on a float texture (no quantisation anywhere): the oracle's
    textureGrad agrees to float precision."""
    rng = np.random.default_rng(8)
    W = H = 32
    base = rng.uniform(0, 3, (H, W, 4)).astype(np.float32)
    s = pkg.Scene("texture_test")
    d, keep = _desc_with(pkg, s, [(W, H, F32, 1, base.reshape(-1))], budget=2**64 - 1)
    osc = orc.OracleScene(d, build_bvh=False)
    chain = [base.astype(np.float64)]
    while chain[-1].shape[0] > 1:
        c = chain[-1]
        chain.append((c[0::2, 0::2] + c[1::2, 0::2] + c[0::2, 1::2] + c[1::2, 1::2]) / 4)

    def bilinear(level, u, v):
        img = chain[level]
        h, w = img.shape[:2]
        x, y = u * w - 0.5, v * h - 0.5
        x0, y0 = np.floor(x), np.floor(y)
        ax, ay = (x - x0)[:, None], (y - y0)[:, None]
        i0, i1, j0, j1 = (x0.astype(int) % w), ((x0.astype(int) + 1) % w), (y0.astype(int) % h), ((y0.astype(int) + 1) % h)
        return (img[j0, i0] * (1 - ax) + img[j0, i1] * ax) * (1 - ay) + (img[j1, i0] * (1 - ax) + img[j1, i1] * ax) * ay

    n = 4000
    u, v = rng.uniform(-1.5, 2.5, n), rng.uniform(-1.5, 2.5, n)
    rho = 2.0 ** rng.uniform(-1.0, 5.5, n)          # footprint in texels: below one texel up to beyond the 1 x 1 level
    ang = rng.uniform(0, 2 * np.pi, n)               # a square footprint, rotated: isotropic (eta = 1)
    dudx, dvdx = rho * np.cos(ang) / W, rho * np.sin(ang) / H
    dudy, dvdy = -rho * np.sin(ang) / W, rho * np.cos(ang) / H
    got = osc.test_texture(_inputs(9, np.float32(u), np.float32(v), np.float32(dudx), np.float32(dvdx), np.float32(dudy), np.float32(dvdy))).view(np.float32)
    uf, vf = np.float64(np.float32(u)), np.float64(np.float32(v))
    lam = np.clip(np.log2(rho), 0, len(chain) - 1)
    l0 = np.floor(lam).astype(int)
    l1 = np.minimum(l0 + 1, len(chain) - 1)
    f = (lam - l0)[:, None]
    want = np.zeros((n, 4))
    for level in range(len(chain)):
        for sel, wgt in ((l0 == level, 1 - f), (l1 == level, f)):
            if sel.any():
                want[sel] += bilinear(level, uf[sel], vf[sel]) * wgt[sel]
    err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    assert err.max() < 3e-5, float(err.max())


def test_anisotropic_texture_grad_against_float64_formulas(pkg, orc):
    """The anisotropic branch of the sampler (material.glsl:72-76 calls textureGrad on a sampler with anisotropy at the device
    maximum, Renderer.cpp:103-110) against an independent float64 restatement of the Vulkan / EXT_texture_filter_anisotropic
    formulas: rho_x, rho_y = lengths of the two gradients in texels, eta = min(rho_max / rho_min, 16), N = ceil(eta) taps at
    lambda = log2(rho_max / eta), spaced along the longer gradient at i / (N + 1) - 1/2, each the linear blend of the two
    nearest levels' bilinear lookups, averaged.  Float texture, no quantisation; footprints from 1:1.05 to 1:40 (clamped at
    16), both axes as the major one.  Cases within 1e-3 of an integer eta are left out: there the float32 and float64 tap
    counts may differ by one, which is a property of ceil, not of the sampler."""
    rng = np.random.default_rng(9)
    W, H = 64, 32
    base = rng.uniform(0, 3, (H, W, 4)).astype(np.float32)
    s = pkg.Scene("texture_test")
    d, keep = _desc_with(pkg, s, [(W, H, F32, 1, base.reshape(-1))], budget=2**64 - 1)
    osc = orc.OracleScene(d, build_bvh=False)
    chain = [base.astype(np.float64)]
    while max(chain[-1].shape[:2]) > 1:  # bilinear clamp-to-edge halving = the 2 x 2 mean while both extents are even, 2 x 1 after
        c = chain[-1]
        hh, ww = c.shape[:2]
        if hh > 1 and ww > 1:
            chain.append((c[0::2, 0::2] + c[1::2, 0::2] + c[0::2, 1::2] + c[1::2, 1::2]) / 4)
        elif ww > 1:
            chain.append((c[:, 0::2] + c[:, 1::2]) / 2)
        else:
            chain.append((c[0::2] + c[1::2]) / 2)

    def bilinear(level, u, v):
        img = chain[level]
        h, w = img.shape[:2]
        x, y = u * w - 0.5, v * h - 0.5
        x0, y0 = np.floor(x), np.floor(y)
        ax, ay = (x - x0)[:, None], (y - y0)[:, None]
        i0, i1, j0, j1 = (x0.astype(int) % w), ((x0.astype(int) + 1) % w), (y0.astype(int) % h), ((y0.astype(int) + 1) % h)
        return (img[j0, i0] * (1 - ax) + img[j0, i1] * ax) * (1 - ay) + (img[j1, i0] * (1 - ax) + img[j1, i1] * ax) * ay

    def trilinear(lam, u, v):
        lam = np.clip(lam, 0, len(chain) - 1)
        l0 = np.floor(lam).astype(int)
        l1 = np.minimum(l0 + 1, len(chain) - 1)
        f = (lam - l0)[:, None]
        out = np.zeros((len(u), 4))
        for level in range(len(chain)):
            for sel, wgt in ((l0 == level, 1 - f), (l1 == level, f)):
                if sel.any():
                    out[sel] += bilinear(level, u[sel], v[sel]) * wgt[sel]
        return out

    n = 6000
    u, v = rng.uniform(-1.5, 2.5, n), rng.uniform(-1.5, 2.5, n)
    major = 2.0 ** rng.uniform(0.0, 5.0, n)            # longer gradient, in texels
    ratio = np.exp(rng.uniform(np.log(1.05), np.log(40.0), n))
    minor = major / ratio
    ang = rng.uniform(0, 2 * np.pi, n)
    swap = rng.integers(0, 2, n).astype(bool)           # which of (d/dx, d/dy) is the longer one
    skew = rng.uniform(-0.4, 0.4, n)                    # the two gradients need not be perpendicular
    ax_, ay_ = np.cos(ang), np.sin(ang)
    bx_, by_ = np.cos(ang + np.pi / 2 + skew), np.sin(ang + np.pi / 2 + skew)
    gx = np.where(swap[:, None], np.stack([minor * bx_, minor * by_], 1), np.stack([major * ax_, major * ay_], 1))  # texels
    gy = np.where(swap[:, None], np.stack([major * ax_, major * ay_], 1), np.stack([minor * bx_, minor * by_], 1))
    dudx, dvdx, dudy, dvdy = (np.float32(gx[:, 0] / W), np.float32(gx[:, 1] / H), np.float32(gy[:, 0] / W), np.float32(gy[:, 1] / H))
    got = osc.test_texture(_inputs(9, np.float32(u), np.float32(v), dudx, dvdx, dudy, dvdy)).view(np.float32)

    uf, vf = np.float64(np.float32(u)), np.float64(np.float32(v))
    mux, mvx, muy, mvy = np.float64(dudx) * W, np.float64(dvdx) * H, np.float64(dudy) * W, np.float64(dvdy) * H
    rx, ry = np.hypot(mux, mvx), np.hypot(muy, mvy)
    rmax, rmin = np.maximum(rx, ry), np.minimum(rx, ry)
    eta = np.minimum(rmax / rmin, 16.0)
    N = np.ceil(eta)
    lam = np.log2(rmax / eta)
    du = np.where(rx >= ry, np.float64(dudx), np.float64(dudy))
    dv = np.where(rx >= ry, np.float64(dvdx), np.float64(dvdy))
    want = np.zeros((n, 4))
    for taps in range(1, 17):
        sel = N == taps
        if not sel.any():
            continue
        acc = np.zeros((int(sel.sum()), 4))
        for i in range(1, taps + 1):
            w = i / (taps + 1.0) - 0.5
            acc += trilinear(lam[sel], uf[sel] + du[sel] * w, vf[sel] + dv[sel] * w)
        want[sel] = acc / taps
    ok = (np.abs(eta - np.round(eta)) > 1e-3) | (eta >= 16.0)
    assert ok.sum() > 0.97 * n and (N[ok] >= 2).sum() > 0.9 * ok.sum() and (N[ok] == 16).sum() > 300
    err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    assert err[ok].max() < 5e-5, float(err[ok].max())


def test_exact_wrap_fast_paths_equal_the_division_formula():
    """Repeat addressing (pt_device.hpp wrapRepeat / wrapRepeatInt / wrapRepeatNext, oracle/pt_oracle.c wrapRepeat): the
    quotient is x0 * rcp(n) -- the specified division, two roundings -- so for an n that is not a power of two floor() can be one
    off either way (x0 = n = 41); the remainder is corrected by one period.  For an integer-valued x0 with |x0| < 2^22 the result
    is the mathematical floor(x0) mod n, so (a) for n = 2^k the quotient may be taken as x0 * 2^-k without a correction, and
    (b) the index of x0 + 1 is the next index.  Emulated here in float32 (every numpy float32 operation rounds like the
    device's), against np.mod, for every extent up to 69, the seams the round-5 review named (41, 47, 55, 61, 82, 83 at x0 = n;
    15 ... 1920 further out), the powers of two and their neighbours up to 32768, 200 random extents; outside the range both
    fall back to the formula itself."""
    f = np.float32

    def wrap_repeat(x0, n):
        fn = f(n)
        rcp = f(np.float64(1.0) / np.float64(fn))     # correctly rounded reciprocal (double rounding cannot occur for 1 / integer < 2^15)
        q = (x0 * rcp).astype(np.float32)
        m = (x0 - (np.floor(q) * fn).astype(np.float32)).astype(np.float32)
        m = np.where(m < 0, (m + fn).astype(np.float32), np.where(m >= fn, (m - fn).astype(np.float32), m)).astype(np.float32)
        m = np.where(m >= 0, m, f(0))
        i = m.astype(np.uint32)
        return np.where(i >= n, n - 1, i)

    def wrap_int(x0, n):
        fn = f(n)
        if n & (n - 1):
            return wrap_repeat(x0, n)
        inv = np.uint32(0x7F000000 - int(fn.view(np.uint32))).view(np.float32)
        assert inv * fn == 1.0
        fast = (x0 - (np.floor((x0 * inv).astype(np.float32)) * fn).astype(np.float32)).astype(np.float32).astype(np.uint32)
        return np.where(np.abs(x0) < f(4194304.0), fast, wrap_repeat(x0, n))

    def wrap_next(x0, i0, n):
        nxt = np.where(i0 + 1 == n, 0, i0 + 1).astype(np.uint32)
        return np.where(np.abs(x0) < f(4194304.0), nxt, wrap_repeat((x0 + f(1)).astype(np.float32), n))

    rng = np.random.default_rng(0)
    extents = list(range(1, 70)) + [82, 83, 120, 125, 250, 500, 1920, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513, 1000, 1023, 1024,
                                    1025, 2047, 2048, 4095, 4096, 4097, 8191, 8192, 16384, 16385, 32767, 32768] \
        + [int(x) for x in rng.integers(2, 32768, 200)]
    for n in extents:
        xs = np.concatenate([rng.integers(-4194303, 4194304, 50000), np.arange(-9 * n - 2, 9 * n + 3), rng.integers(-2**31, 2**31, 5000),
                             n * np.arange(-4194303 // n, 4194303 // n + 1, max(1, (8388606 // n) // 20000)),   # the seams themselves
                             [-4194304, 4194303, 4194304, -4194305, 8388608, -8388608, 2**24, 2**30]]).astype(np.float32)
        a, b = wrap_repeat(xs, n), wrap_int(xs, n)
        assert (a == b).all(), n
        assert (wrap_repeat((xs + f(1)).astype(np.float32), n) == wrap_next(xs, b, n)).all(), n
        inside = np.abs(xs) < 4194304
        assert (a[inside] == np.mod(xs[inside].astype(np.int64), n)).all(), n


# ---------------------------------------------------------------------------------------
# repeat seams of extents that are not a power of two (round-5 review: x0 * rcp(n) floors one off at x0 = k * n)
# ---------------------------------------------------------------------------------------
_SEAM_EXTENTS = [(41, 47), (61, 55), (125, 83), (1000, 3), (82, 15), (1920, 30), (120, 250), (500, 60)]


class _SeamScene:
    """texture_test's PtxSceneDesc with its texture table replaced by RGBA32F images whose texel (x, y) holds
    (x, y, x + 1000 y, 1): a lookup at a texel centre names the texel it fetched."""

    def __init__(self, pkg):
        import ctypes as C

        class TextureDesc(C.Structure):
            _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32), ("levels", C.c_uint32), ("data", C.c_void_p)]

        self.base = pkg.Scene("texture_test")
        self.desc = self.base.desc
        self.images = []
        self.table = (TextureDesc * len(_SEAM_EXTENTS))()
        for i, (w, h) in enumerate(_SEAM_EXTENTS):
            x, y = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
            img = np.ascontiguousarray(np.stack([x, y, x + 1000 * y, np.ones_like(x)], -1), np.float32)
            self.images.append(img)
            self.table[i] = TextureDesc(w, h, 2, 1, img.ctypes.data)
        self.desc.textures = C.addressof(self.table)
        self.desc.textureCount = len(_SEAM_EXTENTS)
        self.desc.forceFullTextureSize = 1


def _seam_inputs():
    """Texel centres displaced by whole periods: u = (i + 1/2) / w + k (k = 0, +-1 ... +-8 and a few thousand), plus the
    seams themselves u = k - 1/(2w) (x0 = k w - 1, next texel 0) and u = k + 1/(2w) (x0 = k w)."""
    rows, want = [], []
    rng = np.random.default_rng(11)
    for t, (w, h) in enumerate(_SEAM_EXTENTS):
        ks = np.concatenate([np.arange(-8, 9), rng.integers(-3000, 3000, 40)])
        for k in ks:
            for i in (0, 1, w // 2, w - 2, w - 1):
                for j in (0, h - 1):
                    u, v = np.float64(i + 0.5) / w + k, np.float64(j + 0.5) / h - k
                    rows.append((9 + t, u, v))
                    want.append((t, i % w, j % h))
    a = np.zeros((len(rows), 7), np.float32)
    a.view(np.uint32)[:, 0] = [r[0] for r in rows]
    a[:, 1] = [r[1] for r in rows]
    a[:, 2] = [r[2] for r in rows]
    return a, np.array(want)


def _expected_texels(inputs):
    """The four texels and weights of the bilinear footprint in exact integer arithmetic (np.mod), from the float32 inputs as
    the sampler sees them."""
    f = np.float32
    out = np.zeros((len(inputs), 4), np.float64)
    for r, a in enumerate(inputs):
        t = int(a.view(np.uint32)[0]) - 9
        w, h = _SEAM_EXTENTS[t]
        x, y = f(a[1] * f(w)) - f(0.5), f(a[2] * f(h)) - f(0.5)
        x0, y0 = np.floor(x), np.floor(y)
        ax, ay = np.float64(f(x - x0)), np.float64(f(y - y0))
        i0, j0 = int(np.mod(np.int64(x0), w)), int(np.mod(np.int64(y0), h))
        i1, j1 = (i0 + 1) % w, (j0 + 1) % h
        tex = lambda i, j: np.array([i, j, i + 1000 * j, 1.0])
        out[r] = (tex(i0, j0) * (1 - ax) + tex(i1, j0) * ax) * (1 - ay) + (tex(i0, j1) * (1 - ax) + tex(i1, j1) * ax) * ay
    return out


def test_oracle_repeat_seams_of_non_power_of_two_extents(pkg, orc):
    ss = _SeamScene(pkg)
    osc = orc.OracleScene(ss.desc, build_bvh=False)
    inp, _ = _seam_inputs()
    got = osc.test_texture(inp, implicit_lod=True).view(np.float32)
    want = _expected_texels(inp)
    err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    assert err.max() < 1e-5, (float(err.max()), inp[int(err.max(axis=1).argmax())])
    # the review's own cases: x0 = n exactly (u = 1 + 1/(2n)) must fetch texel 0 and its neighbour 1, not n - 1
    for t, (w, h) in enumerate(_SEAM_EXTENTS[:3]):
        u = np.float32(1.0 + 0.5 / w)
        one = _inputs(9 + t, np.float32([u]), np.float32([0.5 / h]))
        x = np.float32(u * np.float32(w)) - np.float32(0.5)
        if np.floor(x) == w:
            r = osc.test_texture(one, implicit_lod=True).view(np.float32)[0]
            assert 0.0 <= r[0] <= 1.0, (w, r)


@pytest.mark.gpu
def test_sampler_repeat_seams_match_oracle_bitexact(pkg, orc):
    import torch  # noqa: F401

    ss = _SeamScene(pkg)
    r = pkg.Renderer(device=0)
    r.upload(ss.desc)
    osc = orc.OracleScene(ss.desc, build_bvh=False)
    inp, _ = _seam_inputs()
    rng = np.random.default_rng(12)
    n = 20000
    rnd = np.zeros((n, 7), np.float32)
    rnd.view(np.uint32)[:, 0] = rng.integers(9, 9 + len(_SEAM_EXTENTS), n)
    rnd[:, 1:3] = rng.uniform(-40, 40, (n, 2))
    rnd[:, 3:7] = (10.0 ** rng.uniform(-5, 0.0, (n, 4)) * rng.choice([-1, 1], (n, 4)))
    rnd[:2000, 3:7] = 0
    for batch in (inp, rnd):
        for implicit in (True, False):
            a = r.test_texture(batch, implicit)
            b = osc.test_texture(batch, implicit)
            assert (a == b).all(), f"{int((a != b).any(axis=1).sum())} samples differ (implicit={implicit})"
    want = _expected_texels(inp)
    got = r.test_texture(inp, True).view(np.float32)
    assert (np.abs(got - want) / np.maximum(1.0, np.abs(want))).max() < 1e-5
    r.close()

"""GPU parity tests proper: the HIP path (through the C-ABI) against the CPU oracle and the
golden vectors.  Integer/bit work is compared bit-exactly; floating point images within the
tolerance north_star states (rel-L2 <= 1e-3) -- in practice the images are bit-identical because
both sides fix the same arithmetic conventions."""
import ctypes as C

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

REL_L2_TOL = 1e-3  # BASELINE.json north_star: "<= 1e-3 relative L2 radiance error"


# ---------------------------------------------------------------------------------------
# function level (mirrors Path-Tracing-Tests ShadingTest.* / BsdfTest.* on the device)
# ---------------------------------------------------------------------------------------
def test_functions_match_reference_golden_bitexact(pkg, gpu_renderer):
    golden = util.load_golden("fixed")
    for name, (fn, inp, exp) in golden.items():
        out = gpu_renderer.test_eval(fn, inp)
        ok = util.bits_equal_or_both_nan(out, exp)
        assert ok.all(), f"{name}: {int((~ok).sum())} of {ok.size} outputs differ from the reference GLSL vectors"


def test_functions_match_oracle_bitexact(pkg, orc, gpu_renderer):
    golden = util.load_golden("fixed")
    for name, (fn, inp, exp) in golden.items():
        out = gpu_renderer.test_eval(fn, inp)
        ref = orc.test_eval(fn, inp, exp.shape[1])
        assert util.bits_equal_or_both_nan(out, ref).all(), name


def test_functions_finite_on_reference_grids(pkg, gpu_renderer):
    # ShadingTest.cpp:31-35 etc.: the reference asserts only not-NaN / not-Inf on its grids
    golden = util.load_golden("fixed")
    grids = {"GGXDistribution": 6, "Lambda": 6, "GGXSmith": 6, "DielectricFresnel": 4, "SchlickFresnel": 2,
             "EvaluateReflection": 54, "EvaluateRefraction": 108, "SampleGGX": 24}
    for name, n in grids.items():
        fn, inp, _ = golden[name]
        out = gpu_renderer.test_eval(fn, inp[:n]).view(np.float32)
        assert np.isfinite(out).all(), name


def test_lobe_pdfs_sum_to_one(pkg, gpu_renderer):
    # BsdfTest.cpp:34-40: ASSERT_FLOAT_EQ(sum, 1) (4 ULP)
    fn, inp, _ = util.load_golden("fixed")["sampleLobePdfs"]
    out = gpu_renderer.test_eval(fn, inp).view(np.float32)
    s = out.sum(axis=1, dtype=np.float32)
    assert np.all(np.abs(s - 1.0) <= 4 * np.finfo(np.float32).eps)


def test_sincos_pow_kernels(pkg, orc, gpu_renderer):
    rng = np.random.default_rng(7)
    x = rng.uniform(-0.8, 6.3, size=(4096, 1)).astype(np.float32)
    out = gpu_renderer.test_eval(pkg.FN["sincos"], x)
    ref = orc.test_eval(pkg.FN["sincos"], x, 2)
    assert (out == ref).all()
    of = out.view(np.float32)
    assert np.abs(of[:, 0] - np.sin(x[:, 0].astype(np.float64))).max() < 3e-7
    assert np.abs(of[:, 1] - np.cos(x[:, 0].astype(np.float64))).max() < 3e-7
    xy = np.stack([rng.uniform(0.0, 1.0, 4096), rng.uniform(0.0, 12.0, 4096)], axis=1).astype(np.float32)
    xy[:8] = [[0, 0], [0, 1], [1, 5], [0.5, 0], [1e-30, 3], [0.9, 1e-30], [1, 0], [0.25, 0.5]]
    out = gpu_renderer.test_eval(pkg.FN["pow"], xy)
    ref = orc.test_eval(pkg.FN["pow"], xy, 1)
    assert (out == ref).all()
    truth = np.power(xy[:, 0].astype(np.float64), xy[:, 1].astype(np.float64))
    got = out.view(np.float32)[:, 0].astype(np.float64)
    assert np.all(np.abs(got - truth) <= 2e-7 * np.maximum(truth, 1e-30) + 1e-45)


def test_specified_reciprocal_on_all_inputs(pkg, orc, gpu_renderer):
    """The division of the shader path is a * rcp(b) (oracle/pt_oracle_math.h): the device's v_rcp_f32 + one Newton step against
    the oracle's definition AND against that definition restated in numpy, bit for bit, on all 2^23 mantissas (both signs), every
    exponent, denormals, infinities, NaNs and both flush boundaries (16.8 M + 2.1 M patterns; the same comparison over ALL 2^32
    patterns is tools/experiments/rcp_sqrt_exhaustive.hip, 0 mismatches: profiles/r05_rcp_sqrt_exhaustive.txt); the correctly
    rounded sqrt beside it."""
    b = util.reciprocal_inputs()
    rng = np.random.default_rng(5)
    a = rng.integers(0, 1 << 32, len(b), dtype=np.uint64).astype(np.uint32)  # any bit pattern as the numerator
    a[: 1 << 23] = np.float32(1.0).view(np.uint32)
    pairs = np.stack([a, b], axis=1)
    spec = util.reciprocal_spec(b.view(np.float32))
    for lo in range(0, len(b), 1 << 22):
        sl = slice(lo, lo + (1 << 22))
        out = gpu_renderer.test_eval(pkg.FN["divide"], pairs[sl])
        ref = orc.test_eval(pkg.FN["divide"], pairs[sl], 2)
        assert util.bits_equal_or_both_nan(out, ref).all(), f"device != oracle in chunk {lo}"
        assert util.bits_equal_or_both_nan(out[:, 0], spec[sl].view(np.uint32)).all(), f"device != numpy definition in chunk {lo}"
    # the corner cases the convention answers differently from IEEE `/` (stated in oracle/pt_oracle_math.h): zero numerators over
    # zero / denormal divisors, infinite numerators over |b| > 2^126, x / x outside the range -- the same bits on the device
    from test_oracle_golden import _division_corner_pairs
    ca, cb = _division_corner_pairs()
    corner = np.stack([ca.view(np.uint32), cb.view(np.uint32)], axis=1)
    assert util.bits_equal_or_both_nan(gpu_renderer.test_eval(pkg.FN["divide"], corner), orc.test_eval(pkg.FN["divide"], corner, 2)).all()
    zeros = np.stack([np.where(rng.integers(0, 2, len(b)) == 1, np.uint32(0x80000000), np.uint32(0)).astype(np.uint32), b], axis=1)[: 1 << 22]
    assert util.bits_equal_or_both_nan(gpu_renderer.test_eval(pkg.FN["divide"], zeros), orc.test_eval(pkg.FN["divide"], zeros, 2)).all()
    # round 6: the specified reciprocal square root of normalize() / inversesqrt() -- both exponent parities x all 2^23 mantissas
    # (1 / sqrt(4^k x) = 2^-k / sqrt(x): these 2^24 classes decide every positive normal input), every exponent with random
    # mantissas, denormals, zeros, infinities, negative numbers, NaNs: device == oracle == (float)(1 / sqrt((double)x))
    rs = np.concatenate([np.uint32(0x3f800000) | np.arange(1 << 23, dtype=np.uint32), np.uint32(0x40000000) | np.arange(1 << 23, dtype=np.uint32),
                         rng.integers(0, 1 << 32, 1 << 21, dtype=np.uint64).astype(np.uint32),
                         np.array([0, 0x80000000, 1, 0x007fffff, 0x80000001, 0x00800000, 0x7f7fffff, 0x7f800000, 0xff800000, 0x7fc00000, 0xbf800000], np.uint32)]).reshape(-1, 1)
    for lo in range(0, len(rs), 1 << 22):
        out = gpu_renderer.test_eval(pkg.FN["rsq"], rs[lo:lo + (1 << 22)])
        ref = orc.test_eval(pkg.FN["rsq"], rs[lo:lo + (1 << 22)], 1)
        assert util.bits_equal_or_both_nan(out, ref).all(), f"rsq chunk {lo}: device != oracle"
        xf = rs[lo:lo + (1 << 22), 0].view(np.float32)
        normal = (xf >= np.float32(1.17549435e-38)) & np.isfinite(xf)
        with np.errstate(all="ignore"):
            want = (1.0 / np.sqrt(xf[normal].astype(np.float64))).astype(np.float32)
        assert (out[normal, 0].view(np.float32) == want).all(), f"rsq chunk {lo}: device != float64 definition"
    x = np.concatenate([np.uint32(0x3f800000) | np.arange(1 << 23, dtype=np.uint32), np.uint32(0x40000000) | np.arange(1 << 23, dtype=np.uint32),
                        rng.integers(0, 1 << 32, 1 << 20, dtype=np.uint64).astype(np.uint32)]).reshape(-1, 1)
    for lo in range(0, len(x), 1 << 22):
        out = gpu_renderer.test_eval(pkg.FN["sqrt"], x[lo:lo + (1 << 22)])
        with np.errstate(all="ignore"):
            want = np.sqrt(x[lo:lo + (1 << 22)].view(np.float32)).view(np.uint32)  # numpy's float32 sqrt is the IEEE one
        assert util.bits_equal_or_both_nan(out, want).all(), f"sqrt chunk {lo}"


# ---------------------------------------------------------------------------------------
# traversal: LBVH closest-hit / any-hit against the oracle's brute force
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,detail,box", [("default", 1.0, 8.0), ("chess_like", 0.06, 6.0), ("roughness_cubes", 1.0, 12.0)])
def test_traversal_matches_bruteforce(pkg, orc, name, detail, box):
    import torch  # noqa: F401

    scene = pkg.Scene(name, detail)
    desc = scene.desc
    r = pkg.Renderer()
    r.upload(scene)
    osc = orc.OracleScene(desc, build_bvh=False)
    rng = np.random.default_rng(11)
    rays = util.random_rays(rng, 20000, -box, box)
    if name == "roughness_cubes":
        rays[:, 0] -= 10.0  # the 6x6 cube grid spans x, z in [-21, 1]
        rays[:, 2] -= 10.0
    hits, ids = r.trace_rays(rays, any_hit=False)
    ref = osc.trace_closest(rays, brute_force=True)
    first = util.pair_first(desc)
    miss = ids[:, 0] == 0xFFFFFFFF
    gid = np.where(miss, 0xFFFFFFFF, first[np.minimum(ids[:, 0], len(first) - 2)] + ids[:, 1]).astype(np.uint32)
    assert (gid == ref["tri"]).all(), f"{int((gid != ref['tri']).sum())} rays hit a different triangle"
    h = ~miss
    assert h.sum() > 1000
    for k, f in enumerate(("t", "u", "v")):
        assert (hits[h, k].view(np.uint32) == ref[f][h].view(np.uint32)).all(), f
    occ_hits, _ = r.trace_rays(rays, any_hit=True)
    occ_ref = osc.trace_any(rays, brute_force=True)
    assert ((occ_hits[:, 3] != 0) == (occ_ref != 0)).all()
    r.close()


def test_traversal_edge_cases(pkg, orc):
    import torch  # noqa: F401

    scene = pkg.Scene("default")
    r = pkg.Renderer()
    r.upload(scene)
    osc = orc.OracleScene(scene.desc, build_bvh=False)
    rays = np.array([
        [3, 1, 0, 1e-5, -1, 0, 0, 1e4],            # camera axis
        [0, 50, 0, 1e-5, 0, 1, 0, 1e4],            # leaves the scene: miss
        [-4.5, 1, 0, 1e-5, 0, 0, 1, 1e4],          # axis-parallel (zero direction components)
        [-4.5, 1, 0, 1e-5, 0, -1, 0, 0.5],         # tmax shorter than the first hit
        [np.nan, 0, 0, 1e-5, 1, 0, 0, 1e4],        # NaN origin must terminate and miss
        [-4.5, 1, 0, 1e-5, 0, 0, 0, 1e4],          # zero direction must terminate and miss
    ], dtype=np.float32)
    hits, ids = r.trace_rays(rays)
    ref = osc.trace_closest(rays, brute_force=True)
    first = util.pair_first(scene.desc)
    miss = ids[:, 0] == 0xFFFFFFFF
    gid = np.where(miss, 0xFFFFFFFF, first[np.minimum(ids[:, 0], len(first) - 2)] + ids[:, 1]).astype(np.uint32)
    assert (gid == ref["tri"]).all()
    assert miss[1] and miss[3] and miss[4] and miss[5] and not miss[0]
    r.close()


def test_empty_scene_renders_sky(pkg):
    import torch  # noqa: F401

    r = pkg.Renderer()
    d = pkg.SceneDesc()
    ident = (C.c_float * 12)(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0)
    d.transforms = C.addressof(ident)
    d.transformCount = 1
    r.upload(d)
    r.resize(40, 24)
    scene = pkg.Scene("default")
    r.render(scene.uniform(40, 24, bounces=4), scene.lights)
    img = r.readback()
    assert np.allclose(img[..., :3], np.float32([0.08, 0.09, 0.1])) and (img[..., 3] == 1).all()
    r.close()


# ---------------------------------------------------------------------------------------
# image level
# ---------------------------------------------------------------------------------------
_render_pair = util.render_pair


@pytest.mark.parametrize("backend", [0, 1])
def test_default_scene_image_matches_oracle(pkg, orc, backend):
    # the reference's own "Test Scenes/Default" (ExampleScenes.cpp:320-545), camera Scene.h:259-260
    img, ref = _render_pair(pkg, orc, "default", 1.0, 192, 108, frames=3, depth=4, backend=backend, brute=True)
    assert np.isfinite(img).all()
    assert util.rel_l2(img, ref) <= REL_L2_TOL
    assert (img.view(np.uint32) == ref.view(np.uint32)).all(), "expected bit-identical accumulation images"


@pytest.mark.parametrize("name,detail,depth", [("attenuation_blob", 0.1, 6), ("chess_like", 0.05, 8), ("temple_like", 0.08, 8),
                                               ("roughness_cubes", 1.0, 4), ("street_like", 0.03, 6)])
def test_standin_scenes_match_oracle(pkg, orc, name, detail, depth):
    img, ref = _render_pair(pkg, orc, name, detail, 128, 72, frames=2, depth=depth)
    assert np.isfinite(img).all()
    err = util.rel_l2(img, ref)
    assert err <= REL_L2_TOL, f"{name}: rel-L2 {err}"
    differing = int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())
    assert differing == 0, f"{name}: {differing} pixels are not bit-identical"


def test_thin_lens_matches_oracle(pkg, orc):
    img, ref = _render_pair(pkg, orc, "chess_like", 0.04, 96, 54, frames=2, depth=4, lens=0.08)
    assert (img.view(np.uint32) == ref.view(np.uint32)).all()


def test_batched_frames_equal_sequential_launches(pkg):
    import torch  # noqa: F401

    scene = pkg.Scene("chess_like", 0.05)
    lights = scene.lights
    W, H = 160, 96
    a = pkg.Renderer()
    a.upload(scene)
    a.resize(W, H)
    for f in range(4):
        a.render(scene.uniform(W, H, bounces=6, sample_count=1, total_samples=f), lights)
    seq = a.readback()
    a.reset()
    a.render_frames(scene.uniform(W, H, bounces=6), lights, 0, 4)
    batched = a.readback()
    a.close()
    assert (seq.view(np.uint32) == batched.view(np.uint32)).all()


@pytest.mark.parametrize("backend", ["wavefront", "megakernel"])
def test_pipelined_readback_paths(pkg, backend):
    """ptx_readback_begin / _end: the snapshot is the image at the time of _begin whatever is rendered next, through the
    one-workgroup copy kernel (page-locked buffer, on the auxiliary stream -- or the copy stream of a backend without one) and
    through the runtime's copy (pageable buffer the device cannot address)."""
    import torch

    scene = pkg.Scene("chess_like", 0.05)
    W, H = 168, 104
    r = pkg.Renderer(backend=pkg.BACKEND_WAVEFRONT if backend == "wavefront" else pkg.BACKEND_MEGAKERNEL)
    r.upload(scene)
    r.resize(W, H)
    u = scene.uniform(W, H, bounces=5)
    r.render_frames(u, scene.lights, 0, 8)
    first = r.readback()
    pinned = torch.empty(W * H * 4, dtype=torch.float32, pin_memory=True)
    pinned.fill_(-1.0)
    pageable = np.full((H, W, 4), -1.0, dtype=np.float32)
    for ptr, view in ((pinned.data_ptr(), lambda: pinned.numpy().reshape(H, W, 4)), (pageable.ctypes.data, lambda: pageable)):
        r.readback_begin(ptr, W * H * 16)
        r.reset()  # the next frame starts at once: it must not show in the snapshot
        r.render_frames(u, scene.lights, 8, 8)
        r.readback_end()
        assert (view().view(np.uint32) == first.view(np.uint32)).all()
        second = r.readback()
        r.reset()
        r.render_frames(u, scene.lights, 0, 8)
    assert not (second.view(np.uint32) == first.view(np.uint32)).all()
    with pytest.raises(pkg.PtxError):
        r.readback_begin(pinned.data_ptr(), W * H * 16 - 16)
    r.close()


def test_multi_sample_launch_matches_oracle(pkg, orc):
    # SampleCount > 1 in ONE launch: the RNG state is carried across samples (raygen.rgen:42-46)
    scene = pkg.Scene("default")
    lights = scene.lights
    W, H = 96, 54
    r = pkg.Renderer()
    r.upload(scene)
    r.resize(W, H)
    u = scene.uniform(W, H, bounces=4, sample_count=3, total_samples=5)
    r.render(u, lights)
    img = r.readback()
    osc = orc.OracleScene(scene.desc)
    ref, _ = osc.render(u, lights, W, H)
    r.close()
    assert (img.view(np.uint32) == ref.view(np.uint32)).all()


@pytest.mark.parametrize("threshold", ["0", "100000000"])
def test_multi_sample_launch_restart_queue(pkg, orc, monkeypatch, threshold):
    # the samples after the first start from the restart queue (shadow kernel -> k_restart -> next bounce loop); with
    # PTX_TAIL_THRESHOLD=0 they go through the wavefront bounce kernels (the path a full-size launch takes), with a
    # huge threshold through k_tail.  Counts and image must match the oracle either way.
    monkeypatch.setenv("PTX_TAIL_THRESHOLD", threshold)
    scene = pkg.Scene("chess_like", 0.05)
    lights = scene.lights
    W, H = 128, 72
    r = pkg.Renderer()
    r.upload(scene)
    r.resize(W, H)
    u = scene.uniform(W, H, bounces=6, sample_count=4, total_samples=0)
    r.render(u, lights)
    st = r.stats()
    img = r.readback()
    r.close()
    osc = orc.OracleScene(scene.desc)
    ref, ost = osc.render(u, lights, W, H)
    assert st.pathSamples == ost.pathSamples == W * H * 4 + st.retries
    assert st.segments == ost.segments and st.shadowRays == ost.shadowRays
    assert (img.view(np.uint32) == ref.view(np.uint32)).all()


def test_tile_shards_compose_to_full_frame(pkg):
    import torch

    scene = pkg.Scene("chess_like", 0.05)
    lights = scene.lights
    W, H, world = 200, 120, 3  # ragged: not a multiple of the tile size
    u = scene.uniform(W, H, bounces=5)
    full = pkg.Renderer()
    full.upload(scene)
    full.resize(W, H)
    full.render_frames(u, lights, 0, 2)
    ref = full.readback()
    gathered = pkg.Renderer()
    gathered.upload(scene)
    gathered.resize(W, H)
    gathered.set_tile_shard(0, world, 32)
    # ptx_unpack_shard_host: the owner of the frame writes every shard into the host's page-locked frame while it unpacks it
    host = torch.full((H * W * 4,), -1.0, dtype=torch.float32).pin_memory()
    with pytest.raises(pkg.PtxError):
        gathered.unpack_shard(0, 4096, host.data_ptr(), 16)  # the host buffer must be the whole frame
    for rank in range(world):
        part = pkg.Renderer()
        part.upload(scene)
        part.resize(W, H)
        part.set_tile_shard(rank, world, 32)
        part.render_frames(u, lights, 0, 2)
        img = part.readback()
        mask = pkg.shard_mask(W, H, rank, world, 32)
        assert (img[mask].view(np.uint32) == ref[mask].view(np.uint32)).all()
        assert (img[~mask] == 0).all()
        nbytes = part.shard_bytes(rank)
        buf = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
        part.pack_shard(buf.data_ptr())
        part.synchronize()
        if rank == 1:
            gathered.unpack_shard(rank, buf.data_ptr())  # the plain form and ...
            gathered.unpack_shard(rank, buf.data_ptr(), host.data_ptr(), host.numel() * 4)  # ... the form that also writes the host's frame
        else:
            gathered.unpack_shard(rank, buf.data_ptr(), host.data_ptr(), host.numel() * 4)
        gathered.synchronize()
        part.close()
    gathered.readback_end()  # waits for the last unpack's stores to the host
    assert (host.numpy().reshape(H, W, 4).view(np.uint32) == ref.view(np.uint32)).all()
    out = gathered.readback()
    assert (out.view(np.uint32) == ref.view(np.uint32)).all()
    full.close()
    gathered.close()


def test_full_size_properties(pkg):
    """BASELINE configs[1] as bench.py runs it (1920x1080, 8 spp, depth 8, the scene at detail 1.0: all 1,999,000 triangles)
    through size-independent properties: determinism, additivity of accumulation, finite output, alpha == 1."""
    import torch  # noqa: F401

    scene = pkg.Scene("chess_like", 1.0)
    lights = scene.lights
    W, H = 1920, 1080
    r = pkg.Renderer()
    r.upload(scene)
    r.resize(W, H)
    u = scene.uniform(W, H, bounces=8)
    r.render_frames(u, lights, 0, 8)
    a = r.readback()
    st = r.stats()
    assert st.pathSamples == W * H * 8 + st.retries
    assert np.isfinite(a).all() and (a[..., 3] == 1).all()
    r.reset()
    r.render_frames(u, lights, 0, 8)
    b = r.readback()
    assert (a.view(np.uint32) == b.view(np.uint32)).all(), "two identical launches must be bit-identical"
    r.reset()
    r.render_frames(u, lights, 0, 5)
    r.render_frames(u, lights, 5, 3)
    c = r.readback()
    assert (a.view(np.uint32) == c.view(np.uint32)).all(), "5 + 3 frames must equal 8 frames"
    r.close()


@pytest.mark.parametrize("tag,name,spp,depth,shard", [
    ("configs[2]: 64 spp, depth 8", "temple_like", 64, 8, None),
    ("configs[3]: 256 spp, depth 12, a rank of 4", "atrium_like", 256, 12, (1, 4, 32)),
    ("configs[4]: 1024 spp, depth 16, a rank of 8", "street_like", 1024, 16, (5, 8, 16)),
])
def test_long_sample_schedules_match_oracle(pkg, orc, tag, name, spp, depth, shard):
    """The sample counts and depths of BASELINE configs[2..4] run in full -- every frame of the schedule, in batches of 64
    frames per launch as a production run would issue them -- on a small image, against the oracle's frame-by-frame sum.
    (The full-size frames of the same scenes are compared elsewhere; this covers what only a long schedule reaches: the RNG
    frame index up to 1023, depth-16 paths, accumulation over a thousand additions in frame order.)"""
    import torch  # noqa: F401

    scene = pkg.Scene(name, 0.2)
    lights = scene.lights
    W, H = 64, 48
    r = pkg.Renderer()
    r.upload(scene)
    r.resize(W, H)
    sh = None
    if shard:
        r.set_tile_shard(*shard)
        sh = pkg.TileShard(*shard)
    batch = 64
    for first in range(0, spp, batch):
        r.render_frames(scene.uniform(W, H, bounces=depth), lights, first, min(batch, spp - first))
    img = r.readback()
    r.close()
    osc = orc.OracleScene(scene.desc)
    ref = np.zeros((H, W, 4), np.float32)
    for f in range(spp):
        osc.render(scene.uniform(W, H, bounces=depth, total_samples=f), lights, W, H, accum=ref, shard=sh)
    assert (img.view(np.uint32) == ref.view(np.uint32)).all(), tag
    owned = ref[..., 3] == 1
    assert owned.any() and (shard is None) == bool(owned.all()) and np.isfinite(img).all()


@pytest.mark.parametrize("name,frames", [("chess_like", 2), ("atrium_like", 1), ("temple_like", 1), ("street_like", 1), ("alpha_test", 1), ("texture_test", 1),
                                         ("roughness_cubes", 1), ("reuse_mesh_cubes", 1), ("attenuation_blob", 1), ("default", 2), ("animated_test", 1), ("materials_test", 1)])
def test_full_size_frame_matches_oracle(pkg, orc, name, frames):
    """The benchmark workload itself (chess_like at full detail, 1,999,000 triangles, 1920x1080, depth 8) and the Sponza
    and Sun Temple stand-ins (textures + any-hit; point lights and deep paths): frames of the batch against the oracle,
    bit for bit.  At this size a launch takes every path the small cases skip: several wavefront bounces above the tail
    threshold, 130 K traversal chunks per launch, grids capped at the resident block count, block-wide queue appends
    over 65 K blocks."""
    import torch  # noqa: F401

    scene = pkg.Scene(name, 1.0)
    lights = scene.lights
    W, H = 1920, 1080
    r = pkg.Renderer()
    r.upload(scene)
    r.resize(W, H)
    u = scene.uniform(W, H, bounces=8)
    r.render_frames(u, lights, 0, frames)
    img = r.readback()
    st = r.stats()
    r.close()
    osc = orc.OracleScene(scene.desc)
    ref = np.zeros((H, W, 4), np.float32)
    seg = shadow = 0
    for f in range(frames):
        _, ost = osc.render(scene.uniform(W, H, bounces=8, total_samples=f), lights, W, H, accum=ref)
        seg += ost.segments
        shadow += ost.shadowRays
    assert (st.segments, st.shadowRays) == (seg, shadow)
    differing = int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())
    assert differing == 0, f"{differing} of {W * H} pixels are not bit-identical (rel-L2 {util.rel_l2(img, ref)})"


@pytest.mark.parametrize("tag,name,W,H,frames,depth,lens,sample_count,shard,backend", [
    ("4K depth 16", "street_like", 3840, 2160, 1, 16, 0.0, 1, None, 0),  # BASELINE configs[4] size, 64 point lights
    ("thin lens", "chess_like", 1920, 1080, 1, 8, 0.05, 1, None, 0),
    ("SampleCount 4 in one launch", "chess_like", 1920, 1080, 1, 8, 0.0, 4, None, 0),  # restart queue above the tail threshold
    ("rank 3 of 8, 8 spp", "atrium_like", 1920, 1080, 2, 12, 0.0, 1, (3, 8, 32), 0),  # the shape of a weak-scaling rank
    ("megakernel", "temple_like", 1920, 1080, 1, 8, 0.0, 1, None, 1),
])
def test_full_size_variants_match_oracle(pkg, orc, tag, name, W, H, frames, depth, lens, sample_count, shard, backend):
    import torch  # noqa: F401

    scene = pkg.Scene(name, 1.0)
    lights = scene.lights
    r = pkg.Renderer(backend=backend)
    r.upload(scene)
    r.resize(W, H)
    oshard = None
    if shard:
        r.set_tile_shard(*shard)
        oshard = pkg.TileShard(*shard)
    osc = orc.OracleScene(scene.desc)
    ref = np.zeros((H, W, 4), np.float32)
    seg = gseg = 0
    if sample_count == 1:
        r.render_frames(scene.uniform(W, H, bounces=depth, lens_radius=lens, focal_distance=6.0), lights, 0, frames)
        gseg = r.stats().segments
    for f in range(frames):
        u = scene.uniform(W, H, bounces=depth, sample_count=sample_count, total_samples=f * sample_count, lens_radius=lens, focal_distance=6.0)
        if sample_count > 1:
            r.render(u, lights)
            gseg += r.stats().segments
        _, ost = osc.render(u, lights, W, H, accum=ref, shard=oshard)
        seg += ost.segments
    img = r.readback()
    r.close()
    assert gseg == seg
    if shard:  # pixels of other ranks' tiles are untouched on both sides
        assert (img[..., :3] != 0).any()
    differing = int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())
    assert differing == 0, f"{tag}: {differing} of {W * H} pixels are not bit-identical (rel-L2 {util.rel_l2(img, ref)})"


@pytest.mark.parametrize("name", ["street_like", "atrium_like"])
def test_known_answer_rays_match_bruteforce(pkg, orc, name):
    # the rays of util.known_answer_rays: the HIP tree against the oracle's brute force, bit for bit
    scene = pkg.Scene(name, 1.0)
    r = pkg.Renderer()
    r.upload(scene)
    rays = util.known_answer_ray_array(name)
    hits, _ = r.trace_rays(rays, any_hit=False)
    r.close()
    b = orc.OracleScene(scene.desc, build_bvh=False).trace_closest(rays, brute_force=True)
    for k, f in enumerate(("t", "u", "v")):
        assert (hits[:, k].view(np.uint32) == b[f].view(np.uint32)).all(), f


@pytest.mark.parametrize("backend,threshold", [(0, "0"), (0, "100000000"), (0, None), (1, None)])
def test_nan_inf_samples_restart_like_the_reference(pkg, orc, monkeypatch, backend, threshold):
    """raygen.rgen:99-112: a sample whose radiance is NaN / Inf is thrown away and ALL samples of the launch restart with the
    RNG carried on.  A point light of infinite colour makes that happen whenever it is the light picked (1 of 12 + the
    directional one: probability 1/13 per bounce) and not occluded.  The canonical launch finishes its restarts on the device (k_finish_restarts), the
    multi-sample launch through the restart queue round by round: both must give the oracle's image and counters."""
    import torch  # noqa: F401

    if threshold is not None:
        monkeypatch.setenv("PTX_TAIL_THRESHOLD", threshold)
    scene = pkg.Scene("default")
    lights = scene.lights
    lights.LightCount = 12  # eleven dark ones keep the restart rate low enough for three finite samples in a row
    for i in range(12):
        for k in range(3):
            lights.Lights[i].Color[k] = float("inf") if i == 0 else 0.0
        lights.Lights[i].Position[0], lights.Lights[i].Position[1], lights.Lights[i].Position[2] = 1.0, -2.0, 0.5
        lights.Lights[i].AttenuationConstant = 1.0
    W, H = 96, 54
    r = pkg.Renderer(backend=backend)
    r.upload(scene)
    r.resize(W, H)
    osc = orc.OracleScene(scene.desc)
    ref = np.zeros((H, W, 4), np.float32)
    for f, sc_ in enumerate((1, 1, 3)):
        u = scene.uniform(W, H, bounces=4, sample_count=sc_, total_samples=f)
        r.render(u, lights)
        st = r.stats()
        _, ost = osc.render(u, lights, W, H, accum=ref)
        assert ost.retries > 0, "the scene must provoke restarts"
        assert (st.retries, st.pathSamples, st.segments, st.shadowRays) == (ost.retries, ost.pathSamples, ost.segments, ost.shadowRays)
    img = r.readback()
    r.close()
    assert np.isfinite(img).all()
    assert (img.view(np.uint32) == ref.view(np.uint32)).all()


@pytest.mark.parametrize("backend", [0, 1])
def test_zero_bounces_is_a_black_frame(pkg, orc, backend):
    # raygen.rgen:62: `for (bounce = 0; bounce < BounceCount; ...)` never runs -> radiance 0, alpha 1, nothing traced
    import torch  # noqa: F401

    scene = pkg.Scene("default")
    W, H = 64, 40
    r = pkg.Renderer(backend=backend)
    r.upload(scene)
    r.resize(W, H)
    u = scene.uniform(W, H, bounces=0, sample_count=2)
    r.render(u, scene.lights)
    img, st = r.readback(), r.stats()
    r.render_frames(scene.uniform(W, H, bounces=0), scene.lights, 2, 3)
    img2 = r.readback()
    r.close()
    ref, ost = orc.OracleScene(scene.desc).render(u, scene.lights, W, H)
    assert (img.view(np.uint32) == ref.view(np.uint32)).all() and (img[..., :3] == 0).all() and (img[..., 3] == 1).all()
    assert (st.segments, st.shadowRays, st.pathSamples) == (ost.segments, ost.shadowRays, ost.pathSamples) == (0, 0, W * H * 2)
    assert (img2 == img).all()


def test_destroy_releases_device_memory(pkg):
    """ptx_destroy frees everything a renderer allocated -- tree-build state kept for refits, skinning buffers, the
    output stage -- so create / destroy cycles do not eat HBM."""
    import torch

    scene = pkg.Scene("animated_test", 1.0)
    W, H = 256, 144

    def cycle():
        r = pkg.Renderer()
        r.upload(scene)
        r.resize(W, H)
        scene.update(0.25)
        it, bn = scene.animation_state()
        r.update_animation(it, bn, rebuild=False)  # keeps the build state alive
        r.render_frames(scene.uniform(W, H, bounces=3), scene.lights, 0, 2)
        r.postprocess(2)
        r.read_output()
        r.trace_rays(util.random_rays(np.random.default_rng(0), 64, -2.0, 2.0))
        r.close()

    cycle()  # first use also pays one-off runtime allocations
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(6):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, f"{(free0 - free1) / 2**20:.1f} MiB lost over 6 create / destroy cycles"


def test_error_behaviour(pkg):
    import torch  # noqa: F401

    r = pkg.Renderer()
    scene = pkg.Scene("default")
    with pytest.raises(pkg.PtxError):
        r.render(scene.uniform(8, 8), scene.lights)  # nothing uploaded
    r.upload(scene)
    with pytest.raises(pkg.PtxError):
        r.resize(0, 10)
    r.resize(16, 16)
    u = scene.uniform(16, 16)
    u.SampleCount = 0
    with pytest.raises(pkg.PtxError):
        r.render(u, scene.lights)
    bad = scene.desc
    bad.meshCount = 0  # models now point past the mesh table
    with pytest.raises(pkg.PtxError):
        r.upload(bad)
    r.close()


def test_cpp_host_adapter_end_to_end(pkg, tmp_path):
    """The C++ side of the boundary: examples/render_scene.cpp drives RendererHip (the reference's
    Renderer call order) and must produce the same accumulation as the ctypes path."""
    import subprocess
    import torch  # noqa: F401

    host = pkg.PKG_DIR + "/host"
    exe = str(tmp_path / "render_scene")
    cmd = ["g++", "-std=c++20", "-O2", pkg.REPO_DIR + "/examples/render_scene.cpp"] + [f"{host}/{f}.cpp" for f in
           ("Scene", "Camera", "ExampleScenes", "OutputSaver", "TextureImporter", "JpegDecoder", "SceneImporter", "SceneDescription", "FbxReader", "ObjReader", "RendererHip")] + [f"-I{host}", f"-L{pkg.PKG_DIR}", "-lptx_hip",
           f"-Wl,-rpath,{pkg.PKG_DIR}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    W, H, spp, depth = 160, 90, 4, 4
    out = subprocess.check_output([exe, "default", str(W), str(H), str(spp), str(depth), str(tmp_path / "o.png")], text=True)
    mean_cpp = float(out.split("mean radiance")[1].split()[0])
    scene = pkg.Scene("default")
    r = pkg.Renderer()
    r.upload(scene)
    r.resize(W, H)
    for f in range(spp):
        r.render(scene.uniform(W, H, bounces=depth, total_samples=f), scene.lights)
    img = r.readback()
    r.close()
    mean_py = float((img[..., :3].astype(np.float32) * np.float32(1.0 / spp)).astype(np.float64).sum() / (3.0 * W * H))
    assert abs(mean_cpp - mean_py) <= 1e-6 * max(abs(mean_py), 1e-12)
    # the PNG it wrote is the output stage applied to the same sum
    import zlib
    data = open(tmp_path / "o.png", "rb").read()
    assert data.startswith(b"\x89PNG")
    idat = data[data.index(b"IDAT") + 4:data.index(b"IEND") - 8]
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(H, W * 4 + 1)[:, 1:].reshape(H, W, 4).astype(np.uint32)
    png = (np.cumsum(rows, axis=1) & 255).astype(np.uint8)
    r = pkg.Renderer()
    r.resize(W, H)
    r.write_accumulation(img)
    r.postprocess(spp)
    assert (png == r.read_output()).all()
    r.close()


def test_tree_choice_does_not_change_images(pkg, monkeypatch):
    """ptx_build_accel builds seven candidate trees (PLOC search radius, compactness weight, cubic Morton cells), re-optimises
    them by parallel reinsertion, collapses them to 4-wide nodes by a cost-driven rule and keeps the one that costs sampled rays
    least (k_sample_tree_cost); switches in the environment fix a radius, turn the reinsertion off or up, select the greedy
    collapse of rounds 1-3 or the depth-first node order.  Closest hits are tree-independent by construction, so every render is
    the same bits."""
    import torch  # noqa: F401

    scene = pkg.Scene("street_like", 0.05)
    W, H = 160, 90
    u = scene.uniform(W, H, bounces=6)
    images = []
    switches = ({}, {"PTX_PLOC_RADIUS": "16"}, {"PTX_PLOC_RADIUS": "32"}, {"PTX_REINSERT": "0"}, {"PTX_REINSERT": "12", "PTX_COLLAPSE": "0"},
                {"PTX_NODE_LAYOUT": "1"}, {"PTX_BUILDER": "lbvh", "PTX_REINSERT": "4"})
    for env in switches:
        for k in ("PTX_PLOC_RADIUS", "PTX_REINSERT", "PTX_COLLAPSE", "PTX_NODE_LAYOUT", "PTX_BUILDER"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r = pkg.Renderer()
        r.upload(scene)
        assert r.stats().lastBuildMs > 0
        r.resize(W, H)
        r.render_frames(u, scene.lights, 0, 3)
        images.append(r.readback())
        r.close()
    for k in range(1, len(images)):
        assert (images[0].view(np.uint32) == images[k].view(np.uint32)).all(), switches[k]


def test_single_stream_handles_render_the_same_image(pkg, orc):
    """PTX_DEVICE_SINGLE_STREAM (include/ptx.h): a handle whose shadow and tail kernels ride on its main stream -- what a rank of an
    N-GPU job creates, sixteen of them in flight -- renders the image of the two-stream handle bit for bit, on
    a textured scene with alpha-tested geometry and on an untextured one, whole frame and tile shard, several in flight."""
    import torch  # noqa: F401

    for name, detail, depth in (("chess_like", 0.05, 8), ("alpha_test", 1.0, 6)):
        scene = pkg.Scene(name, detail)
        W, H = 200, 120
        u = scene.uniform(W, H, bounces=depth)
        two = pkg.Renderer()
        two.upload(scene)
        two.resize(W, H)
        two.render_frames(u, scene.lights, 0, 4)
        ref = two.readback()
        ones = [pkg.Renderer(single_stream=True) for _ in range(3)]
        ones[0].upload(scene)
        for r in ones[1:]:
            r.share_scene(ones[0])
        for r in ones:
            r.resize(W, H)
        for _ in range(2):            # twice: the first launch of a shape is driven from the host, the second is the hinted schedule
            for r in ones:
                r.reset()
                r.render_frames(u, scene.lights, 0, 4)
        for r in ones:
            assert (r.readback().view(np.uint32) == ref.view(np.uint32)).all(), name
        ones[0].set_tile_shard(1, 3, 32)
        ones[0].reset()
        ones[0].render_frames(u, scene.lights, 0, 4)
        mask = pkg.shard_mask(W, H, 1, 3, 32)
        img = ones[0].readback()
        assert (img.view(np.uint32)[mask] == ref.view(np.uint32)[mask]).all() and (img[~mask] == 0).all()
        for r in ones[::-1]:
            r.close()
        two.close()


@pytest.mark.gpu
def test_second_full_build_on_a_handle_does_not_reuse_the_level_lists_of_the_first(pkg, orc, monkeypatch):
    """A handle keeps its build state between ptx_build_accel calls.  With no reinsertion pass (PTX_REINSERT=0) nothing recomputes
    the level lists before the collapse prices the tree, so a second full build -- another scene, another node count -- must start
    without the lists of the first (round-5 advice: stale lists gave the collapse a stale order over a stale node count).  Two
    uploads of different scenes on one handle, each checked against a fresh handle's tree (same wide-node count) and image."""
    import torch  # noqa: F401

    monkeypatch.setenv("PTX_REINSERT", "0")
    monkeypatch.setenv("PTX_PLOC_RADIUS", "16")
    W, H = 128, 72
    sa, sb = pkg.Scene("temple_like", 0.08), pkg.Scene("street_like", 0.05)
    kept = pkg.Renderer()
    kept.upload(sa)
    kept.resize(W, H)
    for scene in (sb, sa, sb):
        kept.upload(scene)              # second, third, fourth full build on the kept state
        fresh = pkg.Renderer()
        fresh.upload(scene)
        fresh.resize(W, H)
        u = scene.uniform(W, H, bounces=6)
        kept.reset(); kept.render_frames(u, scene.lights, 0, 2)
        fresh.render_frames(u, scene.lights, 0, 2)
        a, b = kept.readback(), fresh.readback()
        assert kept.stats().bvhNodes == fresh.stats().bvhNodes, "the kept handle collapsed its tree differently"
        assert (a.view(np.uint32) == b.view(np.uint32)).all()
        fresh.close()
    kept.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name,detail", [("street_like", 0.2), ("atrium_like", 0.1)])
def test_level_lists_build_the_same_tree_as_the_fence_kernels(pkg, monkeypatch, scene_name, detail):
    """The bottom-up passes of the build (boxes after every reinsertion pass, the collapse's pricing) run over level lists -- depths
    by pointer jumping, nodes sorted by depth, one launch per level -- instead of the round-4 kernels that climbed from every leaf
    with two fences and one atomic per node (PTX_FENCE_REFIT=1).  Both compute the same floats in a different order of nodes, so
    the chosen tree is the same: the same number of wide nodes after the cost-driven collapse (a different box anywhere would move
    the candidates' sampled costs and the collapse's decisions), and the same closest hits."""
    import torch  # noqa: F401

    scene = pkg.Scene(scene_name, detail)
    rays = util.random_rays(np.random.default_rng(3), 20000, -8.0, 8.0)
    got = []
    for fence in ("0", "1"):
        monkeypatch.setenv("PTX_FENCE_REFIT", fence)
        r = pkg.Renderer()
        r.upload(scene)
        st = r.stats()
        hits, ids = r.trace_rays(rays)
        got.append((st.bvhNodes, st.treeTriangles, hits.view(np.uint32).copy(), ids.copy()))
        r.close()
    assert got[0][0] == got[1][0] and got[0][1] == got[1][1], (got[0][:2], got[1][:2])
    assert (got[0][2] == got[1][2]).all() and (got[0][3] == got[1][3]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name,detail", [("temple_like", 0.15), ("alpha_test", 1.0), ("chess_like", 0.05)])
def test_presplit_triangles_change_nothing(pkg, orc, monkeypatch, scene_name, detail):
    """Triangle pre-splitting (PTX_SPLIT_BUDGET, TreeParams::splitBudget; off by default): a large triangle hangs from several
    leaves, each with the tight box of its piece.  A closest-hit walk that meets it twice finds the same (t, u, v, id) twice, an
    occlusion walk ends at the first, the any-hit stage's nearest ignored candidate is idempotent -- so closest hits equal
    brute force bit for bit (ids, t, u, v), images and counters equal the oracle's, and only the leaf count differs."""
    import torch  # noqa: F401

    monkeypatch.setenv("PTX_SPLIT_BUDGET", "0.5")
    scene = pkg.Scene(scene_name, detail)
    r = pkg.Renderer()
    r.upload(scene)
    st = r.stats()
    assert st.treeReferences > st.treeTriangles, "the budget must have produced extra leaf references"
    rng = np.random.default_rng(11)
    a = util.desc_arrays(scene.desc)
    lo, hi = a["vertices"][:, :3].min(axis=0) - 1.0, a["vertices"][:, :3].max(axis=0) + 1.0
    rays = util.random_rays(rng, 20000, lo, hi)
    hits, ids = r.trace_rays(rays)
    want = orc.OracleScene(scene.desc, build_bvh=False).trace_closest(rays, brute_force=True)
    first = util.pair_first(scene.desc)
    miss = ids[:, 0] == 0xFFFFFFFF
    gid = np.where(miss, 0xFFFFFFFF, first[np.minimum(ids[:, 0], len(first) - 2)] + ids[:, 1]).astype(np.uint32)
    assert (gid == want["tri"]).all() and (hits[:, 0].view(np.uint32) == want["t"].view(np.uint32)).all()
    r.close()
    img, ref = util.render_pair(pkg, orc, scene_name, detail, 160, 96, frames=2, depth=6)
    assert (img.view(np.uint32) == ref.view(np.uint32)).all()

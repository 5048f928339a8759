"""Row N3: keyframe animation of the scene graph (SceneGraph.cpp), skinning (skinning.comp) and the
acceleration-structure update per frame.  Scene `animated_test`: a spinning platform with a bobbing rider
(child of an animated node), a 4-bone skinned tube, an animated point light."""
import ctypes as C

import numpy as np
import pytest

import util


def _mat(t12):
    return np.asarray(t12, np.float64).reshape(3, 4)


def test_scene_graph_animation(pkg):
    s = pkg.Scene("animated_test", 0.25)
    assert s.update(0.0)  # animated scenes always report a change (Scene.cpp:63)
    it0, bn0 = s.animation_state()
    assert it0.shape == (4, 12) and bn0.shape == (4, 12)
    # bind pose: every bone matrix is node * inverse(bind) = identity at tick 0 only for joints that have not
    # rotated yet -- the swing keys start at a non-zero angle, so compare against the platform instead:
    # instance 1 = platform (spin about y, 90 degrees every 30 ticks = 1 s), instance 2 = rider (child node)
    assert np.allclose(_mat(it0[1])[:, :3], np.eye(3), atol=1e-6) and np.allclose(_mat(it0[1])[:, 3], [-2, 0, 0.5])
    s.update(1.0)
    it1, _ = s.animation_state()
    r = _mat(it1[1])[:, :3]
    assert np.allclose(r, [[0, 0, 1], [0, 1, 0], [-1, 0, 0]], atol=1e-5)  # rotation by +90 degrees about y
    # the rider is a child: its world transform = platform * local (translate (0.9, ~0.95, 0), scaled, rotated about x)
    rider = np.vstack([_mat(it1[2]), [0, 0, 0, 1]])
    plat = np.vstack([_mat(it1[1]), [0, 0, 0, 1]])
    local = np.linalg.inv(plat) @ rider
    assert np.allclose(local[:3, 3], [0.9, 0.6 + 0.7 * 0.5, 0.0], atol=1e-5)
    # slerp keeps the rotation part orthogonal after removing the scale
    sc = np.linalg.norm(local[:3, :3], axis=0)
    assert np.allclose(sc, [1.2, 0.85, 1.2], atol=1e-5)
    q = local[:3, :3] / sc
    assert np.allclose(q @ q.T, np.eye(3), atol=1e-5)
    # the animation loops: 3 more seconds bring every transform back (4 s = 120 ticks)
    s.update(3.0)
    it4, bn4 = s.animation_state()
    assert np.allclose(it4, it0, atol=2e-5) and np.allclose(bn4, bn0, atol=2e-5)
    # lights follow their node (Scene.cpp:73-75): the point light slides along x
    s2 = pkg.Scene("animated_test", 0.25)
    s2.update(0.0)
    x0 = s2.lights.Lights[0].Position[0]
    s2.update(2.0)
    assert abs(x0 - (-1.0)) < 1e-6 and abs(s2.lights.Lights[0].Position[0] - 2.5) < 1e-5


def test_keyframe_cursors_do_not_depend_on_the_step_size(pkg):
    """A track's key cursor only moves forward between two loops of its clip (Scene.h, AnimationNode::Sequence::Update): many
    small steps, one large step, steps that land exactly on keys (whole seconds: every 30 ticks) and steps across the loop (4 s)
    must all sample the same pose for the same clock."""
    def state(steps):
        sc = pkg.Scene("animated_test", 0.25)
        sc.update(0.0)
        for dt in steps:
            sc.update(float(dt))
        it, bn = sc.animation_state()
        lp = [sc.lights.Lights[0].Position[k] for k in range(3)]
        return np.concatenate([it.ravel(), bn.ravel(), lp])

    rng = np.random.default_rng(4)
    small = rng.uniform(0.01, 0.12, 40)
    small *= 2.35 / small.sum()
    assert np.allclose(state(small), state([2.35]), atol=3e-5)
    assert np.allclose(state([1.0, 1.0, 1.0]), state([3.0]), atol=3e-5)            # every step ends ON a key
    assert np.allclose(state([0.5] * 6), state([1.0, 1.0, 0.25, 0.75]), atol=3e-5)
    assert np.allclose(state([1.7, 1.7, 1.7]), state([1.1]), atol=5e-5)            # 5.1 s = one loop + 1.1 s
    assert np.allclose(state([9.3]), state([1.3]), atol=1e-4)                       # two loops in one step
    assert not np.allclose(state([1.3]), state([1.1]), atol=1e-3)


def test_oracle_skinning_and_posed_scene(pkg, orc):
    s = pkg.Scene("animated_test", 0.2)
    s.update(0.0)
    desc = s.desc
    assert desc.animatedVertexCount > 0 and desc.animatedIndexCount > 0
    n_tri = orc.OracleScene(desc, build_bvh=False).triangle_count
    rays = util.random_rays(np.random.default_rng(5), 3000, -3.0, 3.0)
    rays[:, 1] = np.abs(rays[:, 1]) + 0.1
    # identity bones reproduce the bind pose (Renderer.cpp:296-303) up to the rounding of w0 * p + w1 * p
    ident = np.tile(np.float32([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]), (4, 1))
    a = orc.OracleScene(desc, build_bvh=False).trace_closest(rays)
    b = orc.OracleScene(desc, build_bvh=False, bones=ident).trace_closest(rays)
    same = a["tri"] == b["tri"]
    assert same.mean() > 0.999 and np.abs(a["t"][same] - b["t"][same]).max() < 1e-4
    # one rigid motion on every bone moves the tube rigidly: hits of rays moved the same way keep their triangle
    ang = 0.4
    rot = np.float32([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    shift = np.float32([0.5, 0.2, -0.3])
    rigid = np.tile(np.hstack([rot, shift[:, None]]).reshape(-1), (4, 1)).astype(np.float32)
    tube_first = util.pair_first(desc)[3]
    on_tube = a["tri"] >= tube_first
    on_tube &= a["tri"] != 0xFFFFFFFF
    assert on_tube.sum() > 30
    moved = rays[on_tube].copy()
    moved[:, 0:3] = moved[:, 0:3] @ rot.T + shift
    moved[:, 4:7] = moved[:, 4:7] @ rot.T
    c = orc.OracleScene(desc, build_bvh=False, bones=rigid).trace_closest(moved)
    same = c["tri"] == a["tri"][on_tube]
    assert same.mean() > 0.97 and np.abs(c["t"][same] - a["t"][on_tube][same]).max() < 1e-3
    # posed instance transforms == a desc whose instance records were edited by hand
    s.update(0.8)
    it, bn = s.animation_state()
    posed = orc.OracleScene(s.desc, instance_transforms=it, bones=bn).trace_closest(rays, brute_force=False)
    brute = orc.OracleScene(s.desc, build_bvh=False, instance_transforms=it, bones=bn).trace_closest(rays, brute_force=True)
    assert (posed["tri"] == brute["tri"]).all() and orc.OracleScene(s.desc).triangle_count == n_tri


@pytest.mark.gpu
def test_animated_frames_refit_rebuild_and_oracle_agree(pkg, orc):
    scene = pkg.Scene("animated_test", 0.5)
    W, H = 144, 81
    r = pkg.Renderer()
    r.upload(scene)
    r.resize(W, H)
    # bind pose straight after the upload
    scene.update(0.0)
    u = scene.uniform(W, H, bounces=4, sample_count=2)
    r.render(u, scene.lights)
    ref, _ = orc.OracleScene(scene.desc).render(u, scene.lights, W, H)
    assert (r.readback().view(np.uint32) == ref.view(np.uint32)).all()
    build_ms = {}
    for frame, dt in enumerate((0.0, 0.45, 0.45, 1.3, 2.9)):
        scene.update(dt)
        it, bn = scene.animation_state()
        u = scene.uniform(W, H, bounces=4, sample_count=2, total_samples=0)
        osc = orc.OracleScene(scene.desc, instance_transforms=it, bones=bn)
        ref, ost = osc.render(u, scene.lights, W, H)
        imgs = {}
        for mode in ("refit", "rebuild"):
            r.update_animation(it, bn, rebuild=(mode == "rebuild"))
            build_ms.setdefault(mode, []).append(r.stats().lastBuildMs)
            r.reset()
            r.render(u, scene.lights)
            st = r.stats()
            assert st.segments == ost.segments and st.shadowRays == ost.shadowRays
            imgs[mode] = r.readback()
        assert (imgs["refit"].view(np.uint32) == imgs["rebuild"].view(np.uint32)).all(), frame
        assert (imgs["refit"].view(np.uint32) == ref.view(np.uint32)).all(), frame
        # traversal after a refit against the oracle's brute force
        rays = util.random_rays(np.random.default_rng(frame), 4000, -3.0, 3.0)
        rays[:, 1] = np.abs(rays[:, 1]) + 0.05
        r.update_animation(it, bn)
        hits, ids = r.trace_rays(rays)
        want = osc.trace_closest(rays, brute_force=True)
        first = util.pair_first(scene.desc)
        miss = ids[:, 0] == 0xFFFFFFFF
        gid = np.where(miss, 0xFFFFFFFF, first[np.minimum(ids[:, 0], len(first) - 2)] + ids[:, 1]).astype(np.uint32)
        assert (gid == want["tri"]).all() and (hits[~miss, 0].view(np.uint32) == want["t"][~miss].view(np.uint32)).all()
    print("LBVH ms per frame: refit", np.round(build_ms["refit"], 3), "rebuild", np.round(build_ms["rebuild"], 3))
    # partial updates and argument checks
    r.update_animation(None, bn)
    r.update_animation(it, None)
    with pytest.raises(pkg.PtxError):
        r.update_animation(it[:2], None)
    r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("detail,limit_ms", [(0.5, None), (1.0, 5.0)])
def test_refit_of_a_large_static_scene_is_cheaper_than_a_rebuild(pkg, detail, limit_ms):
    """ptx_update_animation(ACCEL_REFIT) keeps the topology of the last full build (the reference refits BLAS + TLAS every animated
    frame: AccelerationStructure.cpp:48-57, Renderer.cpp:1750-1754) and answers rays like a rebuild.  At full size (2 M triangles)
    the refit must stay below 5 ms -- cheaper than a rendered frame: the bottom-up passes run over level lists, one launch per
    level, instead of one fence and one atomic per node (round 4: 23 ms; PTX_FENCE_REFIT=1 brings those kernels back)."""
    scene = pkg.Scene("chess_like", detail)
    r = pkg.Renderer()
    r.upload(scene)
    it = np.frombuffer(C.string_at(scene.desc.instances, scene.desc.instanceCount * 52), np.uint8).reshape(-1, 52)[:, 4:].copy().view(np.float32)
    r.update_animation(it, None, rebuild=True)
    rebuild = r.stats().lastBuildMs
    it2 = it.copy()
    it2[:, 3] += 0.25  # every instance shifts along x
    r.update_animation(it2, None)
    refit = r.stats().lastBuildMs
    r.update_animation(it2, None, rebuild=True)
    rays = util.random_rays(np.random.default_rng(2), 20000, -6.0, 6.0)
    a_hits, a_ids = r.trace_rays(rays)
    r.update_animation(it, None, rebuild=True)
    r.update_animation(it2, None)  # refit from the OLD topology
    refit = min(refit, r.stats().lastBuildMs)
    b_hits, b_ids = r.trace_rays(rays)
    assert (a_ids == b_ids).all() and (a_hits.view(np.uint32) == b_hits.view(np.uint32)).all()
    print(f"chess_like x{detail}: {scene.triangle_count} triangles, rebuild {rebuild:.2f} ms, refit {refit:.2f} ms")
    assert refit < rebuild
    if limit_ms is not None:
        assert scene.triangle_count >= 1_900_000 and refit <= limit_ms, f"refit of {scene.triangle_count} triangles took {refit:.2f} ms"
    r.close()


@pytest.mark.gpu
def test_triangles_left_out_of_the_tree_come_back(pkg, orc):
    """Zero-area triangles take no part in the tree (the full build sorts them behind the rest).  An instance collapsed
    to a point by its transform is all such triangles; when a later frame gives it a size again a REFIT cannot use the
    kept topology (they are not in it) and falls back to a full build -- results equal a fresh renderer's and the oracle's."""
    scene = pkg.Scene("animated_test", 0.4)
    scene.update(0.3)
    it, bn = scene.animation_state()
    flat = it.copy()
    flat[1] = 0.0  # the spinning platform: linear part and translation zero -> every triangle of it degenerates to a point
    r = pkg.Renderer()
    r.upload(scene)
    n_all = r.stats().treeTriangles
    assert n_all == r.stats().triangles
    r.update_animation(flat, bn, rebuild=True)
    n_flat = r.stats().treeTriangles
    assert n_flat < n_all, "collapsed triangles must leave the tree"
    rays = util.random_rays(np.random.default_rng(9), 8000, -3.0, 3.0)
    rays[:, 1] = np.abs(rays[:, 1]) + 0.05
    want_flat = orc.OracleScene(scene.desc, build_bvh=False, instance_transforms=flat, bones=bn).trace_closest(rays, brute_force=True)
    hits, ids = r.trace_rays(rays)
    assert (hits[:, 0].view(np.uint32) == want_flat["t"].view(np.uint32)).all()
    r.update_animation(it, bn, rebuild=False)  # refit requested; the revived platform forces the rebuild
    fresh = pkg.Renderer()
    fresh.upload(scene)
    fresh.update_animation(it, bn, rebuild=True)
    assert r.stats().bvhNodes == fresh.stats().bvhNodes and r.stats().treeTriangles == fresh.stats().treeTriangles == n_all
    fresh.close()
    hits, ids = r.trace_rays(rays)
    want = orc.OracleScene(scene.desc, build_bvh=False, instance_transforms=it, bones=bn).trace_closest(rays, brute_force=True)
    first = util.pair_first(scene.desc)
    miss = ids[:, 0] == 0xFFFFFFFF
    gid = np.where(miss, 0xFFFFFFFF, first[np.minimum(ids[:, 0], len(first) - 2)] + ids[:, 1]).astype(np.uint32)
    assert (gid == want["tri"]).all() and (hits[:, 0].view(np.uint32) == want["t"].view(np.uint32)).all()
    # and the other way round: valid -> collapsed under a refit stays in the tree, unhittable
    r.update_animation(flat, bn, rebuild=False)
    hits, _ = r.trace_rays(rays)
    assert (hits[:, 0].view(np.uint32) == want_flat["t"].view(np.uint32)).all()
    r.close()

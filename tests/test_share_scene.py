"""ptx_share_scene: frames in flight render ONE scene and one tree (the reference keeps per-frame rendering resources over
one set of scene buffers, Renderer.cpp:238-439, 1454-1460)."""
import ctypes as C
import os

import numpy as np
import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("name,detail", [("chess_like", 0.05), ("alpha_test", 1.0), ("materials_test", 1.0)])
def test_borrower_renders_the_owners_scene_bit_for_bit(pkg, name, detail):
    import torch  # noqa: F401

    scene = pkg.Scene(name, detail)
    W, H = 160, 96
    u = scene.uniform(W, H, bounces=5)
    owner = pkg.Renderer()
    owner.upload(scene)
    owner.resize(W, H)
    owner.render_frames(u, scene.lights, 0, 3)
    ref = owner.readback()
    borrowers = [pkg.Renderer() for _ in range(3)]
    for k, b in enumerate(borrowers):
        if k == 1:  # one that had a scene of its own: released by the call
            b.upload(pkg.Scene("default"))
        b.share_scene(owner)
        b.resize(W, H)
    for b in borrowers:  # all in flight together with the owner
        b.render_frames(u, scene.lights, 0, 3)
    owner.reset()
    owner.render_frames(u, scene.lights, 0, 3)
    for b in borrowers:
        img = b.readback()
        assert (img.view(np.uint32) == ref.view(np.uint32)).all()
        assert b.stats().triangles == owner.stats().triangles
    assert (owner.readback().view(np.uint32) == ref.view(np.uint32)).all()
    # own lights per borrower: a launch with other lights does not disturb the others
    dark = pkg.LightsUbo()
    borrowers[0].reset()
    borrowers[0].render_frames(u, dark, 0, 1)
    borrowers[2].reset()
    borrowers[2].render_frames(u, scene.lights, 0, 3)
    assert (borrowers[2].readback().view(np.uint32) == ref.view(np.uint32)).all()
    for r in borrowers + [owner]:
        r.close()


@pytest.mark.gpu
def test_share_scene_lifetime_and_errors(pkg):
    import torch  # noqa: F401

    scene = pkg.Scene("default")
    W, H = 64, 48
    u = scene.uniform(W, H, bounces=3)
    owner, b, c = pkg.Renderer(), pkg.Renderer(), pkg.Renderer()
    with pytest.raises(pkg.PtxError):  # nothing to share yet
        b.share_scene(owner)
    with pytest.raises(pkg.PtxError):
        b.share_scene(b)
    owner.upload(scene)
    b.share_scene(owner)
    with pytest.raises(pkg.PtxError):  # a borrower is no owner, an owner with borrowers does not borrow
        c.share_scene(b)
    c.upload(scene)
    with pytest.raises(pkg.PtxError):
        owner.share_scene(c)
    with pytest.raises(pkg.PtxError):  # the tree is the owner's business
        b._check(b.lib.ptx_build_accel(b.handle))
    b.resize(W, H)
    b.render_frames(u, scene.lights, 0, 1)
    first = b.readback()
    # the owner changes its scene while the borrower has a frame in flight: waited for, and the borrower renders the new one
    b.reset()
    b.render_frames(u, scene.lights, 0, 1)
    other = pkg.Scene("roughness_cubes")
    owner.upload(other)
    uo = other.uniform(W, H, bounces=3)
    b.reset()
    b.render_frames(uo, other.lights, 0, 1)
    c.upload(other)
    c.resize(W, H)
    c.render_frames(uo, other.lights, 0, 1)
    assert (b.readback().view(np.uint32) == c.readback().view(np.uint32)).all()
    # between the owner's upload and its build the owner's tree describes the OLD triangles: the borrower must not trace
    owner._check(owner.lib.ptx_scene_upload(owner.handle, C.byref(scene.desc)))
    rays = np.array([[0, 1, 5, 0, 0, 0, -1, 100]], np.float32)
    with pytest.raises(pkg.PtxError):
        b.render_frames(uo, other.lights, 0, 1)
    with pytest.raises(pkg.PtxError):
        b.trace_rays(rays)
    owner._check(owner.lib.ptx_build_accel(owner.handle))
    b.trace_rays(rays)
    owner.upload(other)
    # an upload the borrower's handle REFUSES (an instance names a model that does not exist) leaves it borrowing
    bad = type(other.desc)()
    C.memmove(C.byref(bad), C.byref(other.desc), C.sizeof(bad))
    one = np.zeros(13, np.uint32)  # PtxModelInstance: ModelIndex + 3 x 4 floats
    one[0] = 10 ** 6
    bad.instances, bad.instanceCount = one.ctypes.data, 1
    with pytest.raises(pkg.PtxError):
        b._check(b.lib.ptx_scene_upload(b.handle, C.byref(bad)))
    b.reset()
    b.render_frames(uo, other.lights, 0, 1)
    assert (b.readback().view(np.uint32) == c.readback().view(np.uint32)).all()
    assert b.stats().hardwareQueues == int(os.environ["GPU_MAX_HW_QUEUES"])
    # a borrower that uploads gets its own scene back
    b.upload(scene)
    b.reset()
    b.render_frames(u, scene.lights, 0, 1)
    assert (b.readback().view(np.uint32) == first.view(np.uint32)).all()
    # the owner goes first: the borrower is left without a scene, not with dangling pointers
    b.share_scene(owner)
    owner.close()
    with pytest.raises(pkg.PtxError):
        b.render_frames(uo, other.lights, 0, 1)
    b.close()
    c.close()

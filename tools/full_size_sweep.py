#!/usr/bin/env python3
"""Bit-for-bit comparison of the HIP path with the oracle at BASELINE's full sizes, over more frames and variants than
the test-suite carries (tests/test_gpu_parity.py::test_full_size_*).  Needs an MI355X; the oracle runs on the host cores
(about a minute in total).  Prints one line per case with the number of differing pixels (expected: 0) and their
coordinates, which PTO_DEBUG_PIXEL=x,y / pto_debug_path (oracle/pt_oracle.c) then help to bisect -- this is the sweep
that found the oracle's two tree-walk culling bugs in round 1.

Usage: python tools/full_size_sweep.py [--quick | --scale K]   (K multiplies the frame counts of the batch cases; default 4)"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as graft  # noqa: E402


def main():
    import torch  # noqa: F401  (one HIP runtime in the process: torch first)

    pkg, orc = graft.load_package(), graft.load_oracle()
    quick = "--quick" in sys.argv

    def compare(tag, name, W, H, frames, depth, lens=0.0, sample_count=1, shard=None, backend=0):
        t0 = time.time()
        scene = pkg.Scene(name, 1.0)
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
            r.render_frames(scene.uniform(W, H, bounces=depth, lens_radius=lens, focal_distance=6.0), scene.lights, 0, frames)
            gseg = r.stats().segments
        for f in range(frames):
            u = scene.uniform(W, H, bounces=depth, sample_count=sample_count, total_samples=f * sample_count, lens_radius=lens, focal_distance=6.0)
            if sample_count > 1:
                r.render(u, scene.lights)
                gseg += r.stats().segments
            _, ost = osc.render(u, scene.lights, W, H, accum=ref, shard=oshard)
            seg += ost.segments
        img = r.readback()
        r.close()
        ys, xs = np.nonzero((img.view(np.uint32) != ref.view(np.uint32)).any(axis=-1))
        print(f"{tag:14s} {name:18s} {W}x{H} x{frames} depth {depth}: segments {gseg} / {seg}, differing pixels {len(ys)} "
              f"{list(zip(xs.tolist(), ys.tolist()))[:6]}  ({time.time() - t0:.1f} s)", flush=True)
        return len(ys)

    def animated(frames):
        # animated_test posed at a few times: skinning + refit on the device against the oracle's posed scene
        t0 = time.time()
        W, H = 1920, 1080
        scene = pkg.Scene("animated_test", 1.0)
        r = pkg.Renderer()
        r.upload(scene)
        r.resize(W, H)
        bad = 0
        for step, dt in enumerate((0.0, 0.45, 1.3, 2.9)):
            scene.update(dt)
            it, bn = scene.animation_state()
            r.update_animation(it, bn, rebuild=(step == 2))
            r.reset()
            u = scene.uniform(W, H, bounces=6)
            r.render_frames(u, scene.lights, 0, frames)
            img = r.readback()
            osc = orc.OracleScene(scene.desc, instance_transforms=it, bones=bn)
            ref = np.zeros((H, W, 4), np.float32)
            for f in range(frames):
                osc.render(scene.uniform(W, H, bounces=6, total_samples=f), scene.lights, W, H, accum=ref)
            bad += int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())
        r.close()
        print(f"{'animated':14s} {'animated_test':18s} {W}x{H} x{frames} x 4 poses: differing pixels {bad}  ({time.time() - t0:.1f} s)", flush=True)
        return bad

    k = 1 if quick else 4
    if "--scale" in sys.argv:
        k = int(sys.argv[sys.argv.index("--scale") + 1])
    bad = 0
    bad += compare("batch", "chess_like", 1920, 1080, 2 * k, 8)
    bad += compare("batch", "street_like", 1920, 1080, 2 * k, 8)
    bad += compare("batch", "temple_like", 1920, 1080, max(1, 3 * k // 2), 8)
    bad += compare("batch", "atrium_like", 1920, 1080, k, 12)
    bad += compare("batch", "attenuation_blob", 1920, 1080, 2 * k, 8)
    bad += compare("4K depth 16", "street_like", 3840, 2160, 2, 16)
    bad += compare("4K", "chess_like", 3840, 2160, 2, 8)
    bad += compare("thin lens", "chess_like", 1920, 1080, 3, 8, lens=0.05)
    bad += compare("SampleCount 4", "chess_like", 1920, 1080, 1, 8, sample_count=4)
    bad += compare("rank 3 of 8", "atrium_like", 1920, 1080, 2 * k, 12, shard=(3, 8, 32))
    bad += compare("megakernel", "temple_like", 1920, 1080, 2, 8, backend=1)
    for name in ("default", "roughness_cubes", "reuse_mesh_cubes", "texture_test", "alpha_test", "materials_test"):  # skyboxes, sampler, any-hit, decals, SG / Phong / DX normals
        bad += compare("batch", name, 1920, 1080, 4 * k, 8)
    bad += animated(k)
    print("TOTAL differing pixels:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

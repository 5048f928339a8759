#!/usr/bin/env python3
"""Traversal microbenchmark (GPU): primary rays and diffuse-bounce rays of a scene through
ptx_trace_rays, with node-visit / triangle-test statistics.  Experiment tool, not a test."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import __graft_entry__ as graft

pkg = graft.load_package(); orc = graft.load_oracle()
name = sys.argv[1] if len(sys.argv) > 1 else "chess_like"
detail = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
W, H = 1920, 1080
scene = pkg.Scene(name, detail)
r = pkg.Renderer(); r.upload(scene)
st = r.stats(); print(f"{name}: {scene.triangle_count} tris, build {st.lastBuildMs:.1f} ms")
u = scene.uniform(W, H, bounces=8)
# primary rays in the renderer's slot order (8x8 blocks inside 32x32 tiles)
ys, xs = np.mgrid[0:H, 0:W]
tile = 32
tid = (ys // tile) * ((W + tile - 1) // tile) + xs // tile
inx, iny = xs % tile, ys % tile
order = np.lexsort(((inx % 8).ravel(), (iny % 8).ravel(), (inx // 8).ravel(), (iny // 8).ravel(), tid.ravel()))
px, py = xs.ravel()[order].astype(np.uint32), ys.ravel()[order].astype(np.uint32)
n = px.size
inp = np.zeros((n, 38), np.float32)
iv = inp.view(np.uint32)
iv[:, 0] = px; iv[:, 1] = py; iv[:, 2] = W; iv[:, 3] = H
inp[:, 4:6] = 0.5
inp[:, 6:22] = np.array(u.ViewInverse, np.float32); inp[:, 22:38] = np.array(u.ProjInverse, np.float32)
od = orc.test_eval(pkg.FN["constructPrimaryRay"], inp, 18).view(np.float32)  # the ray and its two offset rays
rays = np.zeros((n, 8), np.float32)
rays[:, 0:3] = od[:, 0:3]; rays[:, 3] = 1e-5; rays[:, 4:7] = od[:, 3:6]; rays[:, 7] = 1e4

PROFILE = os.environ.get("EXP_PROFILE")  # one launch per ray type, no stats pass


def bench(label, rays, any_hit=False):
    if PROFILE:
        hits, ids = r.trace_rays(rays, any_hit)
        print(f"{label:28s} {rays.shape[0]/r.stats().lastTraceMs/1e6:8.3f} Grays/s")
        return hits, ids
    r.trace_rays(rays[:1000], any_hit)
    best = 1e9
    for _ in range(3):
        hits, ids = r.trace_rays(rays, any_hit)
        best = min(best, r.stats().lastTraceMs)
    hs, st = r.trace_rays(rays, 2)
    print(f"{label:28s} {rays.shape[0]/best/1e6:8.3f} Grays/s  ({best:.3f} ms)  hit {hits[:,3].mean():.2f}  "
          f"nodes/ray {st[:,0].mean():.1f} (max {st[:,0].max()})  tris/ray {st[:,1].mean():.2f}")
    return hits, ids

hits, ids = bench("primary closest", rays)
# diffuse bounce: from hit points, cosine-ish random directions in the upper hemisphere of +y / random
rng = np.random.default_rng(1)
h = hits[:, 3] != 0
P = rays[h, 0:3] + rays[h, 4:7] * hits[h, 0:1]
d = rng.normal(size=P.shape).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
d[:, 1] = np.abs(d[:, 1])
b = np.zeros((P.shape[0], 8), np.float32)
b[:, 0:3] = P + 1e-3 * d; b[:, 3] = 1e-5; b[:, 4:7] = d; b[:, 7] = 1e4
bench("bounce closest (incoherent)", b)
# shadow rays towards the directional light
L = np.array(scene.lights.Directional.Direction, np.float32); L = -L / np.linalg.norm(L)
s = b.copy(); s[:, 4:7] = L; s[:, 7] = 1e5
bench("shadow any-hit (coherent)", s, True)
sh = b.copy(); sh[:, 7] = 1e5
bench("shadow any-hit (incoherent)", sh, True)

if os.environ.get("EXP_HEATMAP"):
    hs, stt = r.trace_rays(rays, 2)
    img = np.zeros((H, W), np.float32)
    img[py, px] = stt[:, 0]
    np.save("gpurun_out/visits.npy", img[::2, ::2].astype(np.uint16))
    v = stt[:, 0]
    print("visit percentiles 50/90/99/99.9/99.99/max:", [int(np.percentile(v, q)) for q in (50, 90, 99, 99.9, 99.99)], int(v.max()))
    print("share of all visits spent in rays with > 60 visits:", float(v[v > 60].sum() / v.sum()), "ray share", float((v > 60).mean()))

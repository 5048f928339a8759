#!/usr/bin/env python3
"""Node visits per ray of the two queue traversal kernels over whole frames (GPU, instrumented build).

Usage: tools/visit_histogram.py [scene ...]   (builds tools/scratch/libptx_visits.so from a scratch copy of the sources with tools/experiments/visit_stats.patch applied)
One frame of 1920x1080, 8 spp, depth 8 per scene; prints rays, mean / max visits and the histogram in bins of 16 visits,
for k_trace_closest and k_trace_shadow, plus the depth-1 frame (primary rays and their shadow queries only).
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tools", "scratch", "libptx_visits.so")
PATCH = os.path.join(ROOT, "tools", "experiments", "visit_stats.patch")
import glob
SRC = [os.path.join(ROOT, "path-tracing_amd", "csrc", "ptx_capi.hip")] + sorted(glob.glob(os.path.join(ROOT, "path-tracing_amd", "csrc", "*.hpp")))
os.environ["PTX_HIP_LIB"] = LIB  # before the package is imported: it reads the variable once
if not os.path.exists(LIB) or any(os.path.getmtime(f) > os.path.getmtime(LIB) for f in SRC + [PATCH]):
    # the instrumentation lives in a patch, not in the kernels: apply it to a scratch copy of the sources and build that
    import shutil
    import tempfile
    sys.path.insert(0, ROOT)
    import __graft_entry__ as _graft
    _pkg = _graft.load_package()
    with tempfile.TemporaryDirectory(prefix="ptvisits_") as tmp:
        shutil.copytree(os.path.join(ROOT, "path-tracing_amd", "csrc"), os.path.join(tmp, "path-tracing_amd", "csrc"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
        subprocess.check_call(["patch", "-p1", "-s", "-i", PATCH], cwd=tmp)
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + _pkg.HIPCC_FLAGS + ["-DPT_VISIT_STATS", "-o", LIB,
                               os.path.join(tmp, "path-tracing_amd", "csrc", "ptx_capi.hip")])
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (device runtime)
import __graft_entry__ as graft

pkg = graft.load_package()
dbg = ctypes.CDLL(LIB).ptx_debug_visit_stats
dbg.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]


def stats():
    out, rounds = np.zeros((2, 68), np.uint32), np.zeros((2, 8), np.uint64)
    assert dbg(out.ctypes.data, rounds.ctypes.data, 1) == 0
    return out, rounds


def show_rounds(rs):
    rounds, refills, refilled, nsteps, nlanes, leafs, llanes, waves = (float(x) for x in rs)
    if not rounds:
        return
    print(f"    wave loop: {rounds / waves:8.1f} rounds per wave-launch; a round runs the refill {100 * refills / rounds:.0f} % of the time "
          f"({refilled / max(refills, 1):.1f} lanes), {nsteps / rounds:.2f} node steps ({nlanes / max(nsteps, 1):.1f} of 64 lanes in each), "
          f"the leaf phase {100 * leafs / rounds:.0f} % ({llanes / max(leafs, 1):.1f} lanes)")


def show(label, row):
    mx, total, rays = int(row[0]), int(row[1]), int(row[2])
    if not rays:
        print(f"  {label}: no rays")
        return
    hist = row[4:].astype(np.float64) / rays
    cum = np.cumsum(hist)
    p50, p90, p99 = (int(np.searchsorted(cum, q)) * 16 + 16 for q in (0.5, 0.9, 0.99))
    approx = float((hist * (np.arange(64) * 16 + 8)).sum())  # from bin midpoints; the exact sum is a 32-bit counter
    exact = total / rays if abs(total / rays - approx) < 9 else float("nan")
    print(f"  {label}: {rays / 1e6:7.2f} M rays, visits/ray mean {exact:6.1f} (from bins {approx:6.1f}), max {mx}, "
          f"p50 <= {p50}, p90 <= {p90}, p99 <= {p99}")
    print("    bins of 16 visits, % of rays: " + " ".join(f"{100 * h:.1f}" for h in hist[: max(4, int(np.searchsorted(cum, 0.9995)) + 1)]))


W, H = 1920, 1080
for name in sys.argv[1:] or ["chess_like", "temple_like", "atrium_like", "street_like"]:
    scene = pkg.Scene(name, 1.0)
    r = pkg.Renderer()
    r.upload(scene)
    r.resize(W, H)
    print(f"{name}: {scene.triangle_count} triangles")
    for depth in (1, 8):
        stats()
        r.reset()
        u = scene.uniform(W, H, bounces=depth)
        r.render_frames(u, scene.lights, 0, 8)
        r.synchronize()
        s, rs = stats()
        print(f" depth {depth}, 8 spp:")
        show("k_trace_closest", s[0])
        show_rounds(rs[0])
        show("k_trace_shadow ", s[1])
        show_rounds(rs[1])

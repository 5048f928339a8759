#!/usr/bin/env python3
"""Time ptx_update_animation(ACCEL_REFIT) and the full tree build on a static stand-in scene at full size (task 4).
Usage: tools/refit_timing.py [scene ...]   PTX_FENCE_REFIT=1 selects the round-4 fence-and-atomic kernels for comparison."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    import torch  # noqa: F401  (one HIP runtime in the process)

    pkg = graft.load_package()
    for name in (sys.argv[1:] or ["chess_like", "atrium_like"]):
        scene = pkg.Scene(name, 1.0)
        r = pkg.Renderer()
        t0 = time.time()
        r.upload(scene)
        r.synchronize()
        st = r.stats()
        print(f"{name}: {scene.triangle_count} triangles, upload + best-tree build {time.time() - t0:.2f} s, lastBuildMs {st.lastBuildMs:.1f}, nodes {st.bvhNodes}")
        it = np.frombuffer(C.string_at(scene.desc.instances, scene.desc.instanceCount * 52), np.uint8).reshape(-1, 52)[:, 4:].copy().view(np.float32)
        r.update_animation(it, None, rebuild=True)  # keeps the build state for refits
        print(f"  rebuild (kept state, chosen parameters, 2 reinsertion passes): {r.stats().lastBuildMs:.2f} ms")
        it2 = it.copy()
        it2[:, 3] += 0.25
        for k in range(3):
            t0 = time.time()
            r.update_animation(it2 if k % 2 == 0 else it, None)
            wall = (time.time() - t0) * 1e3
            print(f"  refit {k}: lastBuildMs {r.stats().lastBuildMs:.2f} ms (wall of the call {wall:.2f} ms)")
        r.close()
        scene.close()


if __name__ == "__main__":
    main()

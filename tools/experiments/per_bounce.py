#!/usr/bin/env python3
"""Per-bounce ray counts and kernel times of one frame batch alone on the machine (PTX_VERBOSE's lines), for a scene."""
import os
import sys

os.environ["PTX_VERBOSE"] = "1"
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    import torch  # noqa: F401

    pkg = graft.load_package()
    name = sys.argv[1] if len(sys.argv) > 1 else "chess_like"
    scene = pkg.Scene(name, 1.0)
    r = pkg.Renderer()
    r.upload(scene)
    r.resize(1920, 1080)
    u = scene.uniform(1920, 1080, bounces=8)
    for k in range(3):
        r.reset()
        print(f"--- batch {k}", file=sys.stderr)
        r.render_frames(u, scene.lights, 0, 8)
        r.synchronize()
        r.stats()
    r.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Strong scaling of the named frame, emulated on ONE GPU (the 8-GPU run itself is the driver's).

Usage (on the GPU box): tools/scaling_emulation.py [--scene chess_like] [--world 8] [--gather] [bench.py args ...]
Renders the whole frame (bench.py --emulate-shard 0/1) and then every rank's pixel-tile shard of a WORLD-GPU job in turn
(--emulate-shard r/WORLD: the same ring of frames in flight), each in a child process, and prints one JSON object:

  scaling_emulated = {frame_ms, shard_ms[WORLD], gather_ms_model, speedup_model = frame_ms / (max(shard_ms) + gather_ms_model)}

--gather: every shard run does ITS RANK's whole share of the job inside the timed loop (bench.py --force-gather --dist-backend nccl).
Round 6 (the default): the owner of step k's frame is rank k % WORLD -- every step the rank accumulates straight into the gather's
message and makes the collective call (RCCL on a one-rank group: its launch and a local copy); on the steps it OWNS (one in
WORLD) it also takes the WORLD - 1 pieces that would arrive over xGMI (one device copy), composes the frame with ONE
ptx_unpack_shards launch and stores it to the host's page-locked frame.  `--root rank0 --gather-unpack per-rank
--shard-accumulation packed` (passed through to bench.py) gives round 5's step: rank 0 owns every frame.  The whole frame (0/1)
reads back too.  Only the link time itself stays the model below.

gather_ms_model: the one gather of the step.  Every rank contributes W*H*16/WORLD bytes (plus the padding of ragged tiles) and
sends them to the owner over ITS OWN xGMI link (point-to-point: the WORLD - 1 sends proceed in parallel), so the owner has the frame
after one piece's time at the per-link rate of /opt/skills/guides (153 GB/s); `--collective all_gather` (rounds 1-5: RCCL's ring
all_gather, WORLD - 1 pieces over each link one after the other) is priced as that.  The streams of the frames in flight are
asynchronous, so on hardware this time overlaps the next frame's rendering; the model ADDS it to the step (pessimistic).  Ranks
render different tiles, so the slowest shard sets the step.
What the emulation cannot show: the ranks' launch overheads overlap on real hardware exactly as here (one process per GPU), but the
gather's interaction with the frames in flight is modelled, not measured."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
XGMI_LINK_GBS = 153.0


def main():
    args = sys.argv[1:]
    scene, world, W, H, gather = "chess_like", 8, 1920, 1080, False
    rest = []
    i = 0
    while i < len(args):
        if args[i] == "--gather":
            gather = True; i += 1
        elif args[i] == "--scene":
            scene = args[i + 1]; i += 2
        elif args[i] == "--world":
            world = int(args[i + 1]); i += 2
        else:
            if args[i] == "--width":
                W = int(args[i + 1])
            if args[i] == "--height":
                H = int(args[i + 1])
            rest.append(args[i]); i += 1

    def run(shard):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--scene", scene, "--emulate-shard", shard, "--no-cpu-baseline"] + rest
        env = dict(os.environ)
        if gather and shard == "0/1":
            cmd += ["--emulate-readback", "on"]  # the one-GPU step: read-back, no gather
        elif gather:
            cmd += ["--force-gather", "--dist-backend", "nccl"]
            env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + (hash(shard) % 300)),
                       HSA_ENABLE_IPC_MODE_LEGACY="0")
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        if p.returncode != 0:
            raise SystemExit(p.stderr[-2000:])
        return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])["ms_per_step"]

    frame = run("0/1")
    shards = [run(f"{r}/{world}") for r in range(world)]
    piece = W * H * 16 / world
    gather_in_loop = gather
    ring = "all_gather" in " ".join(rest)
    pieces_per_link = (world - 1) if ring else 1
    gather = pieces_per_link * piece / (XGMI_LINK_GBS * 1e9) * 1e3
    out = {"scene": scene, "world": world, "frame_ms": frame, "shard_ms": shards, "gather_ms_model": gather, "gather_in_loop": gather_in_loop,
           "owner": "rank0" if "rank0" in rest else "rotating (k % world)",
           "gather_model": (f"ring all-gather, {world - 1} pieces of {piece / 1e6:.2f} MB per link at {XGMI_LINK_GBS:.0f} GB/s" if ring else
                            f"gather to the owner, one piece of {piece / 1e6:.2f} MB per link at {XGMI_LINK_GBS:.0f} GB/s, {world - 1} links in parallel"),
           "speedup_model": frame / (max(shards) + gather), "speedup_without_gather": frame / max(shards)}
    print(json.dumps({"scaling_emulated": out}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Strong scaling of the named frame, emulated on ONE GPU (the 8-GPU run itself is the driver's).

Usage (on the GPU box): tools/scaling_emulation.py [--scene chess_like] [--world 8] [--gather] [bench.py args ...]
Renders the whole frame (bench.py --emulate-shard 0/1) and then every rank's pixel-tile shard of a WORLD-GPU job in turn
(--emulate-shard r/WORLD: the same ring of frames in flight), each in a child process, and prints one JSON object:

  scaling_emulated = {frame_ms, shard_ms[WORLD], gather_ms_model, speedup_model = frame_ms / (max(shard_ms) + gather_ms_model)}

--gather (round 5): every shard run does RANK 0's whole gather duty inside the timed loop (bench.py --force-gather --dist-backend
nccl: ptx_pack_shard, the RCCL all_gather_into_tensor call on a one-rank group, one device copy standing in for the WORLD - 1 pieces
that arrive over xGMI, WORLD x ptx_unpack_shard, the pipelined read-back), and the whole frame reads back too; only the link time
itself stays the model below.

gather_ms_model: the one all_gather of the step.  Every rank contributes W*H*16/WORLD bytes (plus the padding of ragged tiles);
RCCL's ring all-gather over point-to-point xGMI moves (WORLD - 1) such pieces over each link, one after the other, at the per-link
rate of /opt/skills/guides (153 GB/s) -- per-link bound, the pessimistic schedule (each peer writing straight to rank 0 over its own
link would take one piece's time).  Ranks render different tiles, so the slowest shard sets the step.
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
    gather = (world - 1) * piece / (XGMI_LINK_GBS * 1e9) * 1e3
    out = {"scene": scene, "world": world, "frame_ms": frame, "shard_ms": shards, "gather_ms_model": gather, "gather_in_loop": gather_in_loop,
           "gather_model": f"ring all-gather, {world - 1} pieces of {piece / 1e6:.2f} MB per link at {XGMI_LINK_GBS:.0f} GB/s",
           "speedup_model": frame / (max(shards) + gather), "speedup_without_gather": frame / max(shards)}
    print(json.dumps({"scaling_emulated": out}))


if __name__ == "__main__":
    main()

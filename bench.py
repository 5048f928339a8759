#!/usr/bin/env python3
"""bench.py -- headline benchmark of the path-tracing hot path on MI355X.

Metric (BASELINE.json): Msamples/s = W*H*spp / wall-seconds / 1e6 at 1920x1080, 8 spp, depth 8.
A "step" is one pass of the hot path over one frame batch: reset the accumulation image, add
`spp` samples per pixel (canonical schedule: one sample per launch, RNG frame = launch index,
SURVEY.md 8a) and -- with N > 1 GPUs -- gather the pixel-tile shards to rank 0 (the single RCCL
exchange of SURVEY.md 8e).  Scene upload and the LBVH build are outside the timed region (the
reference builds its acceleration structures in UpdateSceneData, not in Render).  Inputs are
resident in HBM when the timed region starts.

Workload at N = 1: BASELINE configs[1] "ABeautifulGame, 1920x1080, 8 spp, depth 8" through its
procedural stand-in `chess_like` (the glTF assets are downloaded at CMake time by the reference
and do not exist offline).

N > 1: the frame is cut into 32x32 pixel tiles dealt round-robin to the ranks; no collective inside the
data path, one all_gather of the tile shards per step.  Default `--scaling weak`: per-GPU work is fixed --
every GPU adds `spp` x (W*H) path samples per step, i.e. the N-GPU job is the same frame at N*spp samples
per pixel, each rank rendering its 1/N of the tiles at N*spp (what BASELINE configs[3] and [4] do: more GPUs
come with more samples, 256 spp on 4 and 1024 spp on 8).  `--scaling strong` splits the fixed `spp` frame
instead; a step of 16.6 M samples is then 2 M samples per rank at N = 8 and its duration is dominated by the
latency floor of a bounce sequence, which is why it is not the default.

Launch:  python bench.py --gpus N --steps K --warmup W           (N = 1)
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
import __graft_entry__ as graft  # noqa: E402

STAND_IN = {"chess_like": "configs[1] 'Khronos ABeautifulGame'", "temple_like": "configs[2] 'UE4 Sun Temple'",
            "atrium_like": "configs[3] 'Intel Sponza (MAIN+CURTAINS+IVY)'", "street_like": "configs[4] 'Amazon Bistro night'",
            "attenuation_blob": "configs[0] 'Khronos DragonAttenuation'"}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes_per_closest_ray(n_tris: int) -> int:
    """DESIGN.md 'Roofline': k_trace_closest per ray = L(N)*B_node + B_tri + the
    state the kernel must move: queue index 4 B + ray 32 B read + hit record 20 B written."""
    L = max(1, math.ceil(math.log2(max(n_tris, 2))))
    return 32 * L + 36 + 4 + 32 + 20


def effective_cores() -> int:
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (the
    GPU box exposes 256 logical CPUs but grants a 16-CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(orc, pkg, scene, width, height, depth, seconds):
    """The oracle (a scalar C port of the shader path: OpenMP over 16x16 pixel tiles, binned-SAH BVH)
    on the host cores, on a bounded sample of the same workload: the same full frame, one sample
    per launch, as many frames as fit in `seconds` (a rate, so comparable with the 8-spp GPU run)."""
    desc = scene.desc
    lights = scene.lights
    t0 = time.time()
    osc = orc.OracleScene(desc, build_bvh=True)
    build_s = time.time() - t0
    acc = np.zeros((height, width, 4), np.float32)
    cores = effective_cores()
    frames = 0
    t0 = time.time()
    while True:
        u = scene.uniform(width, height, bounces=depth, sample_count=1, total_samples=frames)
        osc.render(u, lights, width, height, accum=acc, threads=cores)
        frames += 1
        el = time.time() - t0
        if el >= seconds or frames >= 64:
            break
    return {
        "value": width * height * frames / el / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": f"oracle (C, OpenMP dynamic over 16x16 tiles, SAH BVH) on the same {width}x{height} frame, "
                  f"{frames} spp of the same RNG schedule, depth {depth}, {el:.1f} s; BVH build {build_s:.1f} s excluded",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="chess_like")
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--backend", default="wavefront", choices=["wavefront", "megakernel"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every GPU adds spp samples per pixel-equivalent (job = N*spp spp, tile-sharded); "
                         "strong = the fixed spp frame is split over the GPUs")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --single-device lets the N > 1 code path be exercised on a 1-GPU box (testing only)")
    ap.add_argument("--single-device", action="store_true", help="testing only: every rank uses cuda:0")
    ap.add_argument("--emulate-shard", default=None, metavar="R/N",
                    help="experiments only: one process renders the tile shard of rank R of N (no gather) to see what a rank of an "
                         "N-GPU run costs; the printed line is marked and is not a benchmark result")
    ap.add_argument("--traffic-json", default=None,
                    help="per-kernel HBM bytes per launch from tools/pmc_traffic.py (default: newest profiles/r*_traffic.json)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch through torch.distributed.run",
                  file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    pkg = graft.load_package()  # after torch: one HIP runtime in the process
    W, H = args.width, args.height
    scene = pkg.Scene(args.scene, args.detail)
    lights = scene.lights
    backend = pkg.BACKEND_WAVEFRONT if args.backend == "wavefront" else pkg.BACKEND_MEGAKERNEL
    r = pkg.Renderer(device=local_rank, backend=backend)
    t0 = time.time()
    r.upload(scene)
    r.synchronize()
    upload_build_s = time.time() - t0
    r.resize(W, H)
    emu_rank, emu_world = (int(x) for x in args.emulate_shard.split("/")) if args.emulate_shard else (rank, world)
    r.set_tile_shard(emu_rank, emu_world, args.tile)
    u = scene.uniform(W, H, bounces=args.depth)
    # samples per pixel of the whole job; a rank renders its tiles at this many frames
    job_spp = args.spp * (emu_world if args.scaling == "weak" else 1)
    build_ms = r.stats().lastBuildMs
    n_tris = scene.triangle_count

    # gather plumbing (N > 1): equal-size padded shard buffers, one all_gather
    if world > 1:
        shard_floats = max(r.shard_bytes(k) for k in range(world)) // 4
        send = torch.zeros(shard_floats, dtype=torch.float32, device="cuda")
        recv = torch.zeros(world * shard_floats, dtype=torch.float32, device="cuda") if True else None

    def step():
        r.reset()
        r.render_frames(u, lights, 0, job_spp)
        if world > 1:
            r.pack_shard(send.data_ptr())
            r.synchronize()
            if args.dist_backend == "nccl":
                dist.all_gather_into_tensor(recv, send)  # RCCL: every shard straight over its own xGMI link
            else:  # gloo (testing): staged through the host
                parts = [torch.empty(shard_floats) for _ in range(world)]
                dist.all_gather(parts, send.cpu())
                recv.copy_(torch.cat(parts))
            if rank == 0:
                torch.cuda.current_stream().synchronize()
                for k in range(world):
                    r.unpack_shard(k, recv.data_ptr() + k * shard_floats * 4)
        r.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        r.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    trace_ms = shade_ms = shadow_ms = tail_ms = 0.0
    trace_launches = 0
    closest_rays = 0
    segments = shadow = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        st = r.stats()
        trace_ms += st.lastTraceMs
        shade_ms += st.lastShadeMs
        shadow_ms += st.lastShadowMs
        tail_ms += st.lastTailMs
        trace_launches += st.traceLaunches // 2
        closest_rays += st.tracedRays
        segments, shadow = st.segments, st.shadowRays
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    checksum = None
    if rank == 0 and os.environ.get("BENCH_CHECKSUM"):
        img = r.readback()
        checksum = [float(img[..., :3].astype(np.float64).sum()), int(np.isfinite(img).all()), int((img[..., 3] == 1).all())]
    if rank == 0:
        samples = W * H * job_spp * args.steps
        value = samples / elapsed / 1e6
        out = {
            "metric": "Msamples/s (paths*spp/s) at 1920x1080, 8spp, depth 8",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling,  # weak: W*H*spp samples per GPU and step; strong: one fixed spp frame split over the GPUs
            "vs_baseline": None,   # the reference publishes no number for this metric
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{args.scene} (procedural stand-in for BASELINE {STAND_IN.get(args.scene, 'scenes')}), "
                            f"{W}x{H}, {args.spp} spp" + (f" per GPU = {job_spp} spp" if job_spp != args.spp else "") + f", depth {args.depth}",
                "triangles": n_tris, "backend": args.backend, "tile": args.tile,
                "parallelism": f"pixel-tile shard x{world}" + (", 1 RCCL all_gather" if world > 1 else ""),
                "segments_per_sample": segments / (W * H * job_spp / world) if world else None,
                "lbvh_build_ms": build_ms, "upload_plus_build_s": upload_build_s,
                "kernel_ms_per_step": {"k_trace_closest": trace_ms / args.steps, "k_shade": shade_ms / args.steps,
                                       "k_trace_shadow": shadow_ms / args.steps, "k_tail": tail_ms / args.steps},
            },
        }
        if checksum is not None:
            out["config"]["frame_checksum"] = checksum
        if args.emulate_shard:
            out["emulated_shard"] = args.emulate_shard
            out["value"] = value / emu_world  # samples of this shard only
            out["config"]["parallelism"] = f"EMULATION of rank {emu_rank} of {emu_world} (tile shard, no gather)"
        if args.backend == "wavefront" and trace_ms > 0:
            bpr = algorithmic_bytes_per_closest_ray(n_tris)
            achieved = closest_rays * bpr / (trace_ms * 1e-3) / 1e9
            out["roofline"] = {
                "bound": "hbm", "kernel": "k_trace_closest", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "bytes_per_ray": bpr, "rays_per_launch": closest_rays / max(trace_launches, 1),
                "avg_launch_ms": trace_ms / max(trace_launches, 1), "launches": trace_launches,
                "grays_per_s": closest_rays / (trace_ms * 1e-3) / 1e9,
            }
            # HBM traffic comes from separate rocprofv3 --pmc passes of this same command (PMC counters
            # cannot be read from inside the process); the committed summary is attached when the
            # workload is the default one it was collected on.
            import glob
            tj = args.traffic_json or (sorted(glob.glob(os.path.join(REPO, "profiles", "r*_traffic.json"))) or [None])[-1]
            default_workload = (args.scene, args.detail, W, H, args.spp, args.depth, world) == ("chess_like", 1.0, 1920, 1080, 8, 8, 1)
            if tj and os.path.exists(tj) and default_workload:
                t = json.load(open(tj)).get("k_trace_closest")
                if t:
                    out["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = os.path.relpath(tj, REPO) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, (2*FETCH+WRITE)*1024)"
                    out["roofline"]["algorithmic_bytes_per_launch"] = bpr * closest_rays / max(trace_launches, 1)
        if world == 1 and not args.no_cpu_baseline:
            orc = graft.load_oracle()
            out["cpu_baseline"] = cpu_baseline(orc, pkg, scene, W, H, args.depth, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    r.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

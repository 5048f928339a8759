#!/usr/bin/env python3
"""bench.py -- headline benchmark of the path-tracing hot path on MI355X.

Metric (BASELINE.json): Msamples/s = W*H*spp / wall-seconds / 1e6 at 1920x1080, 8 spp, depth 8, "including the final
readback / gather" (SURVEY.md 8d).
A "step" is one pass of the hot path over one frame batch: reset the accumulation image, add `spp` samples per pixel
(canonical schedule: one sample per launch, RNG frame = launch index), with N > 1 GPUs gather the
pixel-tile shards to rank 0 (the single RCCL exchange of SURVEY.md 8e), and read the RGBA32F sum back to page-locked
host memory.  The read-back of step k overlaps the rendering of the steps behind it (ptx_readback_begin: device-side snapshot,
then a one-workgroup copy to the page-locked buffer on the renderer's auxiliary stream -- the reference reads its output back
a frame late too, OutputSaver.cpp:120-199); the
timed region ends when the last image is on the host.  `value` is that read-back-inclusive rate; the rate without any
read-back is reported beside it (`no_readback`).  Scene upload and the tree build are outside the timed region (the
reference builds its acceleration structures in UpdateSceneData, not in Render).  Inputs are resident in HBM when the
timed region starts.

Workload at N = 1: BASELINE configs[1] "ABeautifulGame, 1920x1080, 8 spp, depth 8" through its procedural stand-in
`chess_like` (the glTF assets are downloaded at CMake time by the reference and do not exist offline).
The same line for the stand-ins of configs[3] (north_star's target scene), [2] and [4] rides in `configs`, each measured
as `bench.py --scene NAME` in a child process with the same K.

N > 1: the frame is cut into 32x32 pixel tiles dealt round-robin to the ranks; no collective inside the data path, one
all_gather of the tile shards per step.  The reported metric is the NAMED frame split over the GPUs (`"scaling":
"strong"`: each rank renders its tiles of the same 8-spp frame); the weak-scaling figure (every GPU adds 8 spp: the job
is the same frame at 8 N spp, which is how BASELINE configs[3] and [4] are posed) is measured in the same run and
printed in `weak`.

Frames in flight (`--in-flight`, default 8 per GPU: two streams each = the 16 hardware queues; more share queues and lose --
a rank's 1/8 tile shard takes 1.15 / 1.24 / 1.19 ms per step with 8 / 12 / 16 in flight):
consecutive steps run on renderers that take turns, each on its own stream
with its own path state, as the reference keeps frames in flight (Renderer.cpp:1454-1460): ptx_render only enqueues a
frame -- the bounce loop is driven from the device -- so the latency-bound end of one frame overlaps the head of the
next.  Every step is still one complete frame: reset, 8 spp, gather, read-back.

Order of a run: scene and tree build on the first renderer; at N = 1 the launch-alone leg on it (Job.alone: the same frames one
at a time before any other renderer exists -- the top level of `roofline`, `one_in_flight_*`); the other renderers of the ring
and one frame on each so that its buffers exist and its bounce schedule has been learnt (preparation per renderer, like the
build); W warm-up steps; then the timed region of exactly K steps.  The region is repeated (`--repeats`, default: until >= 2 s have been timed) and the line reports the MEDIAN
region; min / max ride in `spread`.

Launch:  python bench.py --gpus N --steps K --warmup W           (N = 1)
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)
Output: the LAST stdout line is ONE compact JSON object (< 4 KB: compact_line()) -- metric, value, ms_per_step, config, `roofline`,
`cpu_baseline` and one flat entry per BASELINE config.  Everything else the run measured (overlapped launch durations, per-kernel
counter bytes, one-frame-alone figures, spreads, the children's full records) is written to bench_detail.json (and to
gpurun_out/bench_detail.json where that directory exists); nothing but that one line reaches stdout -- file descriptor 1 is pointed
at stderr for the life of the process (RCCL and the HIP runtime print to stdout) and the line is written to the saved descriptor.
"""
from __future__ import annotations

import argparse
import glob
import hashlib
import json
import math
import os
import sys
import time

# Before anything initialises HIP: one hardware queue per stream of the process.  The frames in flight take two streams each (16
# for the default 8); a job that gathers also runs torch's default stream (the gather's buffers, the staging copies) and the
# collective library's own stream -- 17+ streams on 16 queues make two of them take turns, which cost a 1 / 8 shard step of
# chess_like 0.14 ms of its 0.95 (profiles/r06_rccl_presence.txt: 16 queues 1.09 ms, 20 / 24 / 32 queues 0.98 / 0.97 / 0.99).
# The HOST owns this variable: the library only reports what it finds.
_GATHERS = "--force-gather" in sys.argv or any(a == "--gpus" and sys.argv[i + 1:i + 2] not in ([], ["1"]) for i, a in enumerate(sys.argv)) \
    or int(os.environ.get("WORLD_SIZE", "1")) > 1
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24" if _GATHERS else "16")

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
import __graft_entry__ as graft  # noqa: E402

STAND_IN = {"chess_like": "configs[1] 'Khronos ABeautifulGame'", "temple_like": "configs[2] 'UE4 Sun Temple'",
            "atrium_like": "configs[3] 'Intel Sponza (MAIN+CURTAINS+IVY)'", "street_like": "configs[4] 'Amazon Bistro night'",
            "attenuation_blob": "configs[0] 'Khronos DragonAttenuation'"}
EXTRA_SCENES = ("atrium_like", "temple_like", "street_like")
NORTH_STAR_SCENE = "atrium_like"  # BASELINE.json north_star: "Intel Sponza at 1080p / 8 spp on 1 MI355X" through its stand-in
# BASELINE.json `configs` at their own definitions (child processes of the default run).  Samples per pixel of the multi-GPU
# jobs are bounded (a rate: 64 of configs[3]'s 256, 128 of configs[4]'s 1024) so that the default run stays within minutes.
CHILD_CPU_SECONDS = 3.0  # CPU leg of a `configs` child (one frame at least); the headline's own leg is --cpu-seconds
BASELINE_CONFIGS = (
    ("configs[0] Khronos DragonAttenuation, 512x512, 1 spp, depth 4 -- CPU reference path (stand-in attenuation_blob)",
     ["--scene", "attenuation_blob", "--width", "512", "--height", "512", "--spp", "1", "--depth", "4", "--steps", "50", "--warmup", "5"]),
    ("configs[0] the reference's own default scene at the same shape (CreateDefaultScene, ExampleScenes.cpp:320-545)",
     ["--scene", "default", "--width", "512", "--height", "512", "--spp", "1", "--depth", "4", "--steps", "50", "--warmup", "5"]),
    ("configs[2] UE4 Sun Temple, 1920x1080, 64 spp, depth 8 -- 1xMI355X: ONE renderer, 64-frame batches",
     ["--scene", "temple_like", "--spp", "64", "--depth", "8", "--in-flight", "1", "--steps", "4", "--warmup", "1"]),
    ("configs[2] the same with two 64-frame batches in flight (--detail-run only)",
     ["--scene", "temple_like", "--spp", "64", "--depth", "8", "--in-flight", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"]),
    ("configs[3] Intel Sponza, 1920x1080, 256 spp, depth 12 -- 4xMI355X: rank 0's tiles, 64 of the 256 spp per step",
     ["--scene", "atrium_like", "--spp", "64", "--depth", "12", "--shard", "0/4", "--in-flight", "2", "--steps", "4", "--warmup", "2"]),
    ("configs[4] Amazon Bistro night, 3840x2160, 1024 spp, depth 16 -- 8xMI355X: rank 0's tiles, 128 of the 1024 spp per step",
     ["--scene", "street_like", "--width", "3840", "--height", "2160", "--spp", "128", "--depth", "16", "--shard", "0/8", "--in-flight", "2", "--steps", "3",
      "--warmup", "2"]),
)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# What the chip delivers for THIS path's access pattern when nothing depends on anything: every lane fetching its own 64-byte record
# (four dwordx4) at a permuted index of a 1 GiB buffer -- tools/experiments/fetch_size_calibration.hip, profiles/r05_fetch_size_calibration.txt:
# 3.14 TB/s (the coalesced float4 stream of the same run: 6.74 TB/s).  The traversal's counter bytes are priced against it as well.
GATHER_PEAK_GBS = 3140.0


CLOSEST_STATE_BYTES = 4 + 32 + 20  # what k_trace_closest must move besides the tree: queue index 4 B + ray 32 B read + hit record 20 B written


def algorithmic_bytes_per_closest_ray(n_tris: int) -> int:
    """SURVEY.md 8(d), the contract figure `roofline.frac` is priced on: B_trace_closest = L(N) * B_node + B_tri = 32 L(N) + 36
    bytes per ray, L = ceil(log2 N).  (`frac_with_state` adds CLOSEST_STATE_BYTES: the kernel's own queue / ray / hit records.)"""
    L = max(1, math.ceil(math.log2(max(n_tris, 2))))
    return 32 * L + 36


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


def source_digest(pkg) -> str:
    """Identity of the kernels being measured: the HIP sources the library is built from."""
    h = hashlib.sha256()
    for f in pkg.hip_sources():
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(orc, scene, width, height, depth, seconds):
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
    osc.close()
    return {
        "value": width * height * frames / el / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": f"oracle (C, OpenMP dynamic over 16x16 tiles, SAH BVH) on the same {width}x{height} frame, "
                  f"{frames} spp of the same RNG schedule, depth {depth}, {el:.1f} s; BVH build {build_s:.1f} s excluded",
    }


class FrameStore:
    """The host frames of a job: `count` page-locked RGBA32F frames, frame j at ptr(j).

    One process (N = 1, emulations): torch's page-locked allocator.  N > 1: ONE POSIX shared-memory segment that every rank of the
    node maps; a rank registers with the HIP runtime (hipHostRegister) the frames it OWNS -- its gather kernel stores straight into
    them -- and any process of the node (rank 0 here; an OutputSaver thread in a host application, OutputSaver.cpp:120-199) reads
    every frame whichever rank composed it.  The name is unlinked as soon as every rank has mapped the segment: nothing stays in
    /dev/shm behind a job, however it ends."""

    def __init__(self, torch, dist, count, nbytes, rank, world, owned, register=True):
        self.torch, self.count, self.nbytes, self.shared = torch, count, nbytes, world > 1
        self.stride = (nbytes + 4095) // 4096 * 4096
        self.registered = []
        if not self.shared:
            self.tensors = [torch.empty(nbytes // 4, dtype=torch.float32, pin_memory=True) for _ in range(count)]
            return
        import ctypes
        import mmap
        path = f"/dev/shm/ptx_frames_{os.environ.get('MASTER_PORT', '0')}_{os.getuid()}"
        size = self.stride * count
        if rank == 0:
            fd = os.open(path, os.O_CREAT | os.O_RDWR | os.O_TRUNC, 0o600)
            os.ftruncate(fd, size)
        dist.barrier()
        if rank != 0:
            fd = os.open(path, os.O_RDWR)
        self.mm = mmap.mmap(fd, size)
        os.close(fd)
        dist.barrier()
        if rank == 0:
            os.unlink(path)
        self.base = ctypes.addressof(ctypes.c_char.from_buffer(self.mm))
        if not register:  # (CPU tests of the segment itself: no HIP runtime to register with)
            return
        rt = torch.cuda.cudart()
        for j in owned:
            err = rt.cudaHostRegister(self.base + j * self.stride, self.stride, 0)
            if int(err) != 0:
                raise RuntimeError(f"hipHostRegister of host frame {j} failed: {err}")
            self.registered.append(self.base + j * self.stride)

    def ptr(self, j):
        return self.base + j * self.stride if self.shared else self.tensors[j].data_ptr()

    def image(self, j, H, W):
        if not self.shared:
            return self.tensors[j].numpy().reshape(H, W, 4)
        return np.frombuffer(self.mm, dtype=np.float32, count=H * W * 4, offset=j * self.stride).reshape(H, W, 4)

    def close(self):
        if self.shared and self.registered:
            rt = self.torch.cuda.cudart()
            for a in self.registered:
                rt.cudaHostUnregister(a)
            self.registered = []


class Job:
    """The scene, `in_flight` renderers on one GPU and the step() of the metric, for any (rank, world).

    Frames in flight: step k runs on renderer k % F, each renderer on its own stream with its own path state, so the
    latency-bound end of one frame (few live paths, every kernel waiting on its longest ray) overlaps the head of the
    next -- the reference keeps frames in flight for the same reason (Swapchain in-flight count, one set of rendering
    resources per frame: Renderer.cpp:1454-1460,1617-1618).  ptx_render only ENQUEUES a frame (the bounce loop is driven
    from the device), so nothing in a step waits for the GPU; a renderer's statistics are read when its turn comes again."""

    def __init__(self, args, pkg, torch, dist, scene_name, rank, world, local_rank, shard=None, alone_steps=0):
        self.args, self.pkg, self.torch, self.dist = args, pkg, torch, dist
        self.rank, self.world = rank, world
        self.W, self.H = args.width, args.height
        self.scene = pkg.Scene(scene_name, args.detail)
        self.scene_has_textures = self.scene.desc.textureCount > 0  # k_shade<true> / k_tail<1,2> run instead of the plain variants
        self.lights = self.scene.lights
        backend = pkg.BACKEND_WAVEFRONT if args.backend == "wavefront" else pkg.BACKEND_MEGAKERNEL
        # A rank of an N-GPU job renders THIN frames (1 / N of the pixels): it keeps 16 of them in flight on ONE stream each, where
        # a whole-frame renderer keeps 8 on two streams each (its shadow and tail kernels beside the next bounce) -- the same 16
        # hardware queues either way.  1 / 8 shard step with the owner's duty in the loop, two streams x 8 -> one stream x 16:
        # chess_like 1.05 -> 0.95 ms, street_like 1.61 -> 1.60, atrium_like 2.69 -> 2.60, temple_like 2.29 -> 2.20; 20 in flight
        # fall off a cliff (3.4 ms); a whole frame loses 2 % with one stream (profiles/r06_single_stream*.txt).
        gathers = world > 1 or args.force_gather
        # ... and so is any frame batch below a few million path slots: BASELINE configs[0] (512 x 512, 1 spp: 262 K slots) 591 -> 809
        # Msamples/s with one stream x 16 frames (604 with one stream x 8, 626 x 18, 496 with two streams x 9)
        slots_per_step = args.width * args.height * args.spp / (shard[1] if shard else max(world, 1))
        thin = slots_per_step < 4.0e6
        self.single_stream = (args.streams_per_frame == 1) if args.streams_per_frame else (gathers or thin)
        self.F = args.in_flight if args.in_flight > 0 else (16 if self.single_stream else 8)
        self.shard_rank, self.shard_world = shard if shard else (rank, world)
        self.u = self.scene.uniform(self.W, self.H, bounces=args.depth)
        self.streams = [torch.cuda.Stream()]
        self.rs = [pkg.Renderer(device=local_rank, backend=backend, stream=self.streams[0].cuda_stream, single_stream=self.single_stream)]
        t0 = time.time()
        self.rs[0].upload(self.scene)
        self.rs[0].synchronize()
        self.upload_build_s = time.time() - t0
        self.rs[0].resize(self.W, self.H)
        self.rs[0].set_tile_shard(self.shard_rank, self.shard_world, args.tile)
        # One frame at a time while this renderer's two streams are the only ones of the process (see alone())
        self.alone_stats = self.alone(args.spp, alone_steps) if alone_steps > 0 else None
        for _ in range(1, self.F):
            self.streams.append(torch.cuda.Stream())
            self.rs.append(pkg.Renderer(device=local_rank, backend=backend, stream=self.streams[-1].cuda_stream, single_stream=self.single_stream))
        for r in self.rs[1:]:  # the frames in flight share one scene and one tree, as the reference's per-frame resources do
            r.share_scene(self.rs[0])
            r.resize(self.W, self.H)
            r.set_tile_shard(self.shard_rank, self.shard_world, args.tile)
        self.build_ms = self.rs[0].stats().lastBuildMs
        self.n_tris = self.scene.triangle_count
        self.nbytes = self.W * self.H * 16
        self.k = 0
        self.host_issue_s = 0.0
        self.issued = [False] * self.F
        self.collect = None
        self.gather = world > 1 or args.force_gather
        # --emulate-shard R/N with --force-gather: one rank's whole share of an N-GPU step on ONE GPU -- the collective call (one
        # rank: RCCL's launch and a local copy) every step and, on the steps whose frame this rank OWNS, the N - 1 pieces that would
        # arrive over xGMI (one device copy), the one gather launch and the frame's way to the host.  The link time itself is the
        # model of tools/scaling_emulation.py; everything else is timed.
        self.pieces = self.shard_world if (args.emulate_shard and args.force_gather) else world
        # The OWNER of step k's frame: the rank that composes it from the gathered shards and hands it to the host.  Rotating
        # (k % N): the RNG is a pure function of (pixel, width, frame) (common.glsl:143-147) and every frame in flight has its own
        # resources (Renderer.cpp:1454-1460), so any rank may own any frame -- no rank carries the read-back of every step.
        self.rotate = args.root == "rotate"
        # host frames: step k lands in frame k % L, L = lcm(ranks, frames in flight) -- (k % N, k % F) name the owner and the ring
        # slot, and a frame is reused only after the step that wrote it has left the ring
        self.L = self.F * self.pieces // math.gcd(self.F, self.pieces) if self.gather else self.F
        me = self.shard_rank if self.pieces != world else rank
        owned = [j for j in range(self.L) if (j % self.pieces if self.rotate else 0) == me] if self.gather else list(range(self.L))
        self.frames = FrameStore(torch, dist, self.L, self.nbytes, rank, world, owned)
        if self.gather:  # gather plumbing: equal-size padded shard buffers per frame in flight, one collective per step
            self.shard_floats = max(self.rs[0].shard_bytes(k) for k in range(self.pieces)) // 4
            self.send = [torch.zeros(self.shard_floats, dtype=torch.float32, device="cuda") for _ in range(self.F)]
            self.recv = [torch.zeros(self.pieces * self.shard_floats, dtype=torch.float32, device="cuda") for _ in range(self.F)]
            if world > 1 and args.dist_backend == "nccl" and args.collective == "gather":
                # one trial gather on a few bytes before anything is timed: a collective library that refuses the grouped send / recv
                # form raises here on every rank alike, and the job falls back to all_gather_into_tensor (rounds 1-5) instead of dying
                try:
                    probe = torch.full((4,), float(rank), device="cuda")
                    parts = [torch.empty(4, device="cuda") for _ in range(world)] if rank == 0 else None
                    dist.gather(probe, parts, dst=0)
                    torch.cuda.synchronize()
                    if rank == 0 and [float(p[0]) for p in parts] != [float(q) for q in range(world)]:
                        raise RuntimeError("dist.gather delivered the wrong pieces")
                except Exception as e:  # noqa: BLE001  (whatever the backend raises: the fallback is the point)
                    print(f"[bench] rank {rank}: dist.gather failed ({e}); using all_gather_into_tensor", file=sys.stderr)
                    args.collective = "all_gather"
                flag = torch.tensor([1 if args.collective == "gather" else 0], dtype=torch.int32, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # every rank takes the same decision
                if int(flag.item()) == 0:
                    args.collective = "all_gather"
            if args.shard_accumulation == "bound":
                # raygen.rgen:115-117's accumulation goes straight into the message of the gather: no ptx_pack_shard pass
                torch.cuda.synchronize()
                for r, send in zip(self.rs, self.send):
                    r.bind_shard_accumulation(send.data_ptr(), self.shard_floats * 4)

    def owner(self, k):
        """The rank that owns the frame of step k (`me` of an emulation is the emulated rank)."""
        return k % self.pieces if self.rotate else 0

    def i_own(self, k):
        return self.owner(k) == (self.shard_rank if self.pieces != self.world else self.rank)

    def _take_stats(self, i):
        if self.issued[i] and self.collect is not None:
            st = self.rs[i].stats()  # waits for that renderer's frame: F steps old by now
            c = self.collect
            c["trace_ms"] += st.lastTraceMs
            c["shade_ms"] += st.lastShadeMs
            c["shadow_ms"] += st.lastShadowMs
            c["tail_ms"] += st.lastTailMs
            c["launches"] += st.traceLaunches // 2
            c["rays"] += st.tracedRays
            c["segments"], c["shadow"] = st.segments, st.shadowRays
        self.issued[i] = False

    def step(self, job_spp, readback=True):
        torch, dist = self.torch, self.dist
        k = self.k
        i = k % self.F
        self.k += 1
        r = self.rs[i]
        self._take_stats(i)
        t_host = time.perf_counter()  # from here on nothing waits for the GPU: what the host spends enqueueing one step
        r.reset()
        r.render_frames(self.u, self.lights, 0, job_spp)  # (bound shard accumulation: k_accumulate writes the message itself)
        self.issued[i] = True
        frame = self.frames.ptr(k % self.L)
        if self.gather:
            send, recv = self.send[i], self.recv[i]
            mine, root = self.i_own(k), self.owner(k)
            if self.args.shard_accumulation != "bound":
                r.pack_shard(send.data_ptr())
            with torch.cuda.stream(self.streams[i]):  # the renderer runs on this torch stream: the collective is ordered behind the frame
                if self.pieces != self.world:  # emulation (see __init__): the call every rank makes, the pieces only the owner receives
                    if self.args.emulate_collective == "on":
                        dist.all_gather_into_tensor(recv[:self.shard_floats], send)
                    if mine:
                        recv[self.shard_floats:].view(self.pieces - 1, self.shard_floats).copy_(send.expand(self.pieces - 1, self.shard_floats))
                elif self.args.dist_backend == "nccl" and self.args.collective == "gather":
                    # the single gather of SURVEY.md 8(e): grouped ncclSend / ncclRecv, every piece over its own xGMI link to the owner
                    dist.gather(send, list(recv.view(self.world, self.shard_floats).unbind(0)) if mine else None, dst=root)
                elif self.args.dist_backend == "nccl":
                    dist.all_gather_into_tensor(recv, send)  # every shard to every rank: N x the bytes, one ring kernel
                else:  # gloo (testing): staged through the host
                    parts = [torch.empty(self.shard_floats) for _ in range(self.world)] if mine else None
                    dist.gather(send.cpu(), parts, dst=root)
                    if mine:
                        recv.copy_(torch.cat(parts))
            if mine:
                if self.args.gather_unpack == "one":
                    # ONE launch composes the frame from all N pieces; with a read-back it stores to the host's frame ONLY -- the
                    # owner hands the frame on (OutputSaver's role) and has no use for a device copy of it
                    if readback:
                        r.unpack_shards(recv.data_ptr(), self.shard_floats * 4, False, frame, self.nbytes)
                    else:
                        r.unpack_shards(recv.data_ptr(), self.shard_floats * 4, True)
                    readback = False
                else:  # rounds 1-5: N launches, device image and (fused) the host's frame
                    fused = readback and self.args.gather_readback == "fused"
                    for q in range(self.pieces):
                        r.unpack_shard(q, recv.data_ptr() + q * self.shard_floats * 4, frame if fused else 0, self.nbytes)
                    if fused:
                        readback = False
            else:
                readback = False
        if readback:
            r.readback_begin(frame, self.nbytes)
        self.host_issue_s += time.perf_counter() - t_host

    def finish(self):
        for i, r in enumerate(self.rs):
            r.readback_end()
            r.synchronize()
            self._take_stats(i)

    def last_image(self):
        """The frame of the last step, from the job's frame store (after finish() + barrier(): whichever rank composed it)."""
        return self.frames.image((self.k - 1) % self.L, self.H, self.W)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()
        for r in self.rs:
            r.synchronize()

    def timed(self, job_spp, steps, readback=True, collect=None):
        """EXACTLY `steps` steps between two barriers; max over ranks.  collect: dict of per-kernel accumulators."""
        self.barrier()
        self.collect = collect
        t0 = time.perf_counter()
        self.host_issue_s = 0.0
        for _ in range(steps):
            self.step(job_spp, readback)
        self.host_enqueue_s = self.host_issue_s  # the host's own share of the region: enqueueing, without the waits for statistics
        self.finish()
        self.barrier()
        elapsed = time.perf_counter() - t0
        self.collect = None
        if self.world > 1:
            t = self.torch.tensor([elapsed], dtype=self.torch.float64, device="cuda" if self.args.dist_backend == "nccl" else "cpu")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    def measure(self, job_spp, steps, warmup, repeats, min_seconds, readback=True):
        """Warm up, then repeat the K-step timed region; returns (median, all regions, kernel stats of the last region).

        Setup before the W warm-up steps: every renderer of the ring renders one frame of the job's shape, so that its
        path-state buffers exist and its bounce schedule has been learnt (the first launch of a shape is driven bounce by
        bounce from the host) -- per-renderer preparation like the BVH build, not a step of the benchmark."""
        for _ in range(self.F):
            self.step(job_spp, readback)
        self.finish()
        for _ in range(warmup):
            self.step(job_spp, readback)
        self.finish()
        regions, stats = [], None
        total = 0.0
        while True:
            c = {"trace_ms": 0.0, "shade_ms": 0.0, "shadow_ms": 0.0, "tail_ms": 0.0, "launches": 0, "rays": 0, "segments": 0, "shadow": 0}
            el = self.timed(job_spp, steps, readback, c)
            regions.append(el)
            stats = c
            total += el
            done = len(regions) >= repeats if repeats > 0 else (total >= min_seconds or len(regions) >= 50)
            if self.world > 1:  # every rank must take the same decision
                flag = self.torch.tensor([1 if done else 0], dtype=self.torch.int32, device="cuda" if self.args.dist_backend == "nccl" else "cpu")
                self.dist.broadcast(flag, 0)
                done = bool(flag.item())
            if done:
                break
        return float(np.median(regions)), regions, stats

    def alone(self, job_spp, steps):
        """The same frames ONE at a time on the first renderer, before the other frames in flight exist: the kernels' durations
        without other frames' kernels sharing the machine, and the latency of one frame batch.  With frames in flight a launch's
        duration counts the time it shares the CUs with the launches of the other frames, so the per-launch figure of the timed
        region falls as the overlap (and the throughput) rises; this one is the kernel's own.  Measured BEFORE the other renderers
        are created because the events that bracket a launch pass through the command processor, which takes longer once more
        hardware queues are mapped, whether they have work or not (the same three steps right after a timed region with 2 / 4 / 8
        frames in flight: 0.78 / 0.83 / 0.87 ms per closest-hit launch, 9.7 / 11.2 / 11.0-11.4 ms per frame; a process with ONE
        frame in flight, which is what `rocprofv3 --kernel-trace` of `--in-flight 1` sees: 0.75-0.77 and 9.6-9.8)."""
        r = self.rs[0]
        for _ in range(2):  # buffers, the bounce schedule (the first launch of a shape is driven from the host)
            r.reset()
            r.render_frames(self.u, self.lights, 0, job_spp)
            r.synchronize()
        c = {"trace_ms": 0.0, "shade_ms": 0.0, "shadow_ms": 0.0, "tail_ms": 0.0, "launches": 0, "rays": 0, "segments": 0, "shadow": 0}
        t0 = time.perf_counter()
        for _ in range(steps):
            r.reset()
            r.render_frames(self.u, self.lights, 0, job_spp)
            r.synchronize()
            st = r.stats()
            c["trace_ms"] += st.lastTraceMs
            c["shade_ms"] += st.lastShadeMs
            c["shadow_ms"] += st.lastShadowMs
            c["tail_ms"] += st.lastTailMs
            c["launches"] += st.traceLaunches // 2
            c["rays"] += st.tracedRays
            c["segments"], c["shadow"] = st.segments, st.shadowRays
        c["wall_s"], c["steps"] = time.perf_counter() - t0, steps
        return c

    def close(self):
        for r in self.rs:
            r.close()
        self.frames.close()
        self.scene.close()


RENDER_KERNELS = ("k_generate", "k_prologue", "k_trace_closest", "k_shade", "k_shade_tex", "k_shade_split", "k_trace_shadow", "k_apply_shadow", "k_tail",
                  "k_finish_restarts", "k_restart", "k_accumulate", "k_copy_out", "k_upload_lights", "__amd_rocclr_copyBuffer", "__amd_rocclr_fillBufferAligned")


def segment_model_bytes(n_tris: int) -> int:
    """B_segment(N) = 2 (32 L(N) + 36) + 796 bytes."""
    L = max(1, math.ceil(math.log2(max(n_tris, 2))))
    return 2 * (32 * L + 36) + 796


def shape_key(W, H, spp, depth, shard):
    """What a counter summary was collected on, beside the scene: image size / samples per pixel / depth / tile shard."""
    r, n = shard if shard else (0, 1)
    return f"{W}x{H}/{spp}spp/d{depth}/shard{r}of{n}"


def counter_doc(job, kind):
    """Per-kernel counters per launch from separate rocprofv3 --pmc passes of this same command on this scene and shape (PMC
    counters cannot be read from inside the process): the newest committed summary, profiles/r*_{traffic,sq}*.json, whose
    `scene` and `shape` match the job (tools/pmc_traffic.py, tools/pmc_sq.py), or --traffic-json."""
    a = job.args
    if a.detail != 1.0 or job.world != 1:
        return None, None
    want = shape_key(job.W, job.H, a.spp, a.depth, (job.shard_rank, job.shard_world))
    if kind == "traffic" and a.traffic_json:
        cands = [a.traffic_json]
    else:
        cands = sorted(glob.glob(os.path.join(REPO, "profiles", f"r*{kind}*.json")), key=os.path.getmtime, reverse=True)
    for f in cands:
        if not os.path.exists(f):
            continue
        doc = json.load(open(f))
        if doc.get("scene", "chess_like") == job.scene.name and doc.get("shape", shape_key(1920, 1080, 8, 8, None)) == want:
            return doc, f
    return None, None


def traffic_doc(job):
    return counter_doc(job, "traffic")


VALU_ISSUE_PER_S = 256 * 4 * 2.4e9 / 2  # wave-instructions per second the chip's 1024 SIMDs can issue: a wave64 VALU instruction
                                         # takes 2 cycles on a SIMD-32 (/opt/skills/guides/MI355X_MICROARCH.md) at 2.4 GHz


def shade_roofline(job, stats_x):
    """k_shade is the largest kernel of the shipped configuration (eight frames in flight) and is bound by instruction issue,
    not by bytes: its figure is VALU wave-instructions per launch (SQ_INSTS_VALU of the committed SQ pass of this scene and
    shape) over what the SIMDs can issue in the launch's own duration (launch alone on the machine, live HIP events)."""
    doc, f = counter_doc(job, "sq")
    if not doc or not stats_x or stats_x["launches"] <= 0:
        return None
    name = "k_shade<true>" if any(k.startswith("k_shade<true>") for k in doc) and job.scene_has_textures else "k_shade<false>"
    k = doc.get(name) or doc.get("k_shade")
    if not k or "SQ_INSTS_VALU" not in k:
        return None
    ms = stats_x["shade_ms"] / stats_x["launches"]
    valu = k["SQ_INSTS_VALU"]
    out = {"kernel": name, "bound": "valu-issue", "valu_wave_insts_per_launch": valu, "ms_alone": ms,
           "issue_ms": valu / VALU_ISSUE_PER_S * 1e3, "frac_valu_issue": valu / VALU_ISSUE_PER_S / (ms * 1e-3),
           "peak": VALU_ISSUE_PER_S / 1e9, "unit": "G wave-instructions/s", "source": os.path.relpath(f, REPO),
           "stale": doc.get("source_digest") != source_digest(job.pkg)}
    if "SQ_WAVE_CYCLES" in k and k["SQ_WAVE_CYCLES"]:
        out["wave_cycles_waiting"] = k.get("SQ_WAIT_ANY", 0.0) / k["SQ_WAVE_CYCLES"]
    if k.get("rocprof_ms_alone"):
        # the kernel's own duration (rocprofv3 --kernel-trace of this command with one frame in flight, committed beside the counters):
        # the live HIP-event bracket (ms_alone) also holds the wait for the previous bounce's k_apply_shadow on the auxiliary stream
        out["ms_alone_rocprof"] = k["rocprof_ms_alone"]
        out["frac_valu_issue_rocprof"] = valu / VALU_ISSUE_PER_S / (k["rocprof_ms_alone"] * 1e-3)
    return out


def kernel_roofline(job, stats, bpr):
    trace_s = stats["trace_ms"] * 1e-3
    launches = max(stats["launches"], 1)
    achieved = stats["rays"] * bpr / trace_s / 1e9
    return {"achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "avg_launch_ms": stats["trace_ms"] / launches, "launches": stats["launches"],
            "rays_per_launch": stats["rays"] / launches, "model_bytes_per_launch": bpr * stats["rays"] / launches,
            "grays_per_s": stats["rays"] / trace_s / 1e9}


def roofline(job, stats, digest, stats_x=None, step_ms=None, segments_per_sample=None):
    """Dominant kernel k_trace_closest: algorithmic bytes (SURVEY.md 8d model) and counter-measured HBM bytes per launch over
    the kernel's launch duration from live HIP events.  The kernel is latency-bound (dependent node fetches), not HBM-bound:
    `bound` says so, `frac` prices the model bytes against the HBM peak as the contract asks, `frac_counter` the bytes that moved.

    Which duration: with frames in flight a launch's duration counts the time it shares the machine with the other frames'
    launches -- six launches of 1.6 ms per 7.7 ms step are more than the step -- so the top-level figures use the launch ALONE on
    the machine (`stats_x`: the same frames one at a time, measured live before the other frames in flight exist), and the timed region's
    overlapped durations ride in `overlapped`.  `step` prices the WHOLE step instead of one kernel: model bytes per sample x samples
    and the sum of every render kernel's counter bytes, over the step time of the timed region."""
    bpr = algorithmic_bytes_per_closest_ray(job.n_tris)
    over = kernel_roofline(job, stats, bpr)
    top = kernel_roofline(job, stats_x, bpr) if stats_x and stats_x["trace_ms"] > 0 else over
    out = {
        "bound": "hbm", "kernel": "k_trace_closest", "achieved": top["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": top["frac"],  # (32 L(N) + 36) B per ray
        "frac_with_state": top["frac"] * (bpr + CLOSEST_STATE_BYTES) / bpr,  # + the kernel's own queue / ray / hit records (rounds 1-5's figure)
        "traffic": None, "achieved_counter": None, "frac_counter": None,
        "measured": ("launch alone on the machine (the same frames one at a time on the first renderer, live HIP events, before the other frames in flight exist)" if top is not over
                     else "launches of the timed region (live HIP events)"),
        "model_bytes_per_ray": bpr, "model_bytes_per_launch": top["model_bytes_per_launch"],
        "rays_per_launch": top["rays_per_launch"], "avg_launch_ms": top["avg_launch_ms"], "launches": top["launches"],
        "grays_per_s": top["grays_per_s"],
        "limiter": "dependent-fetch latency (SQ_WAIT_ANY / SQ_WAVE_CYCLES in profiles/*_sq.txt); priced against HBM as the contract asks",
    }
    if top is not over:
        out["overlapped"] = dict(over, what=f"launches of the timed region, {job.F} frames in flight: a launch's duration counts the time it "
                                            "shares the machine with the other frames' launches (falls as overlap and throughput rise)")
    doc, tj = traffic_doc(job)
    if doc and doc.get("k_trace_closest"):
        traffic = doc["k_trace_closest"]["hbm_bytes_per_launch"]
        out["traffic"] = traffic
        out["achieved_counter"] = traffic / (out["avg_launch_ms"] * 1e-3) / 1e9
        out["frac_counter"] = out["achieved_counter"] / HBM_PEAK_GBS
        out["frac_counter_of_gather_rate"] = out["achieved_counter"] / GATHER_PEAK_GBS  # against the measured rate of independent 64-B gathers
        if "overlapped" in out:
            out["overlapped"]["achieved_counter"] = traffic / (over["avg_launch_ms"] * 1e-3) / 1e9
            out["overlapped"]["frac_counter"] = out["overlapped"]["achieved_counter"] / HBM_PEAK_GBS
        out["traffic_factor"] = doc["k_trace_closest"].get("fetch_factor", 2.0)  # FETCH_SIZE -> bytes, measured per access pattern
        out["traffic_source"] = (os.path.relpath(tj, REPO) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, (factor*FETCH+WRITE)*1024; factor 1 for "
                                 "per-lane 64-B records, 2 for coalesced streams: profiles/r05_fetch_size_calibration.txt)")
        out["traffic_stale"] = doc.get("source_digest") != digest
    if step_ms and segments_per_sample:
        samples = job.W * job.H * job.args.spp / job.shard_world  # what THIS rank renders per step
        model = (segments_per_sample * segment_model_bytes(job.n_tris) + 32) * samples
        step = {"what": "the whole step (every kernel of one frame batch) over ms_per_step of the timed region",
                "model_bytes": model, "frac_model": model / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "model": "SURVEY 8d: (segments_per_sample * B_segment(N) + 32) * samples; counts bytes the caches serve, so it can exceed what HBM delivers",
                "counter_bytes": None, "frac_counter": None}
        if doc and doc.get("k_generate", {}).get("launches"):
            frames = doc["k_generate"]["launches"]  # one k_generate per step of the profiled command
            total = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in doc.items() if isinstance(v, dict) and k in RENDER_KERNELS)
            step["counter_bytes"] = total / frames
            step["frac_counter"] = step["counter_bytes"] / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            upper = sum(v.get("hbm_bytes_per_launch_upper", v["hbm_bytes_per_launch"]) * v["launches"] for k, v in doc.items()
                        if isinstance(v, dict) and k in RENDER_KERNELS)
            step["frac_counter_upper"] = upper / frames / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS  # FETCH_SIZE doubled for every kernel (rounds 1-4)
            step["counter_bytes_by_kernel"] = {k: v["hbm_bytes_per_launch"] * v["launches"] / frames for k, v in doc.items()
                                               if isinstance(v, dict) and k in RENDER_KERNELS and v["hbm_bytes_per_launch"] * v["launches"] / frames > 1e6}
        out["step"] = step
        # the whole-step figures a reader should see first: what HBM moved over the step time; the closed-form model counts
        # bytes the caches serve and stops being a roofline once it exceeds the peak
        out["frac_step_counter"] = step["frac_counter"]
        out["frac_step_model"] = step["frac_model"]
        out["model_valid"] = bool(step["frac_model"] <= 1.0)
    shade = shade_roofline(job, stats_x)
    if shade:
        out["shade"] = shade
    sq, _ = counter_doc(job, "sq")
    if sq:  # rocprofv3's own average of the same launch alone (committed kernel trace, one frame in flight): must agree with avg_launch_ms
        alone = [v for k, v in sq.items() if isinstance(v, dict) and k.startswith("k_trace_closest") and v.get("rocprof_ms_alone")]
        if alone:
            best = max(alone, key=lambda v: v.get("rocprof_calls_alone", 0))
            out["avg_launch_ms_rocprof"] = best["rocprof_ms_alone"]
    return out


LINE_LIMIT = 4096  # bytes of the one stdout line (the driver keeps ~8 KB of stdout tail and parses the last line)


def _r(x, digits=5):
    """Numbers of the compact line: 5 significant digits are more than the run-to-run spread."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if not math.isfinite(x):
            return None
        return float(f"{x:.{digits}g}")
    return x


def _pick(d, keys):
    return {k: _r(d[k]) for k in keys if d is not None and k in d and d[k] is not None}


def config_entry(rec):
    """One flat entry per BASELINE config: {key, value, ms_per_step, frac_step_counter, cpu}."""
    if "is" in rec:
        return {"key": rec["baseline_config"].split(" ")[0] + " = this line"}
    cfg = rec.get("config", {})
    e = {"key": rec["baseline_config"].split(" ")[0] + " " + cfg.get("workload", "").split(" ")[0] + " " + cfg.get("shape", ""),
         "value": _r(rec.get("value")), "ms_per_step": _r(rec.get("ms_per_step")),
         "frac_step_counter": _r((rec.get("roofline") or {}).get("frac_step_counter")),
         "frac": _r((rec.get("roofline") or {}).get("frac")),
         "cpu": _r((rec.get("cpu_baseline") or {}).get("value"))}
    if cfg.get("frames_in_flight") is not None:
        e["frames_in_flight"] = cfg["frames_in_flight"]
    return e


def compact_line(full, detail_path=None):
    """The line the driver parses: the contract's keys, `roofline`, `cpu_baseline`, one flat entry per BASELINE config -- under
    LINE_LIMIT bytes whatever the full record holds (optional parts are dropped, last first, if a string ever grows)."""
    out = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data",
                       "source_digest", "gpu_over_cpu", "one_in_flight_ms_per_step", "one_in_flight_value", "n_ranks_seen"))
    out["vs_baseline"] = full.get("vs_baseline")
    cfg = full.get("config", {})
    out["config"] = _pick(cfg, ("workload", "shape", "triangles", "frames_in_flight", "segments_per_sample", "parallelism", "backend", "tile",
                                "tree_build_ms"))
    if "workload" in out["config"]:
        out["config"]["workload"] = out["config"]["workload"][:200]
    if "no_readback" in full:
        out["no_readback_value"] = _r(full["no_readback"]["value"])
    if "weak" in full:
        out["weak"] = _pick(full["weak"], ("scaling", "value", "ms_per_step", "workload"))
    rf = full.get("roofline")
    if rf:
        o = _pick(rf, ("kernel", "bound", "limiter", "achieved", "peak", "unit", "frac", "frac_with_state", "traffic", "traffic_factor", "frac_counter", "frac_counter_of_gather_rate", "frac_step_counter",
                       "frac_step_model", "model_valid", "avg_launch_ms", "avg_launch_ms_rocprof", "rays_per_launch", "model_bytes_per_ray", "traffic_stale", "measured_on"))
        if rf.get("traffic_source"):
            o["traffic_source"] = rf["traffic_source"].split(" ")[0]
        if rf.get("shade"):
            o["shade"] = _pick(rf["shade"], ("kernel", "frac_valu_issue", "ms_alone", "ms_alone_rocprof", "valu_wave_insts_per_launch"))
        out["roofline"] = o
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind"))
        out["cpu_baseline"]["sample"] = cb.get("sample", "")[:160]
    if full.get("configs"):
        out["configs"] = [config_entry(c) for c in full["configs"]]
    ns = full.get("north_star")
    if ns:  # BASELINE.json north_star's own target: its scene at 1080p / 8 spp / depth 8, whole frame on one GPU, GPU over host CPU >= 10
        out["north_star"] = {"key": ns.get("config", {}).get("workload", "").split(" ")[0] + " " + ns.get("config", {}).get("shape", ""),
                             "value": _r(ns.get("value")), "ms_per_step": _r(ns.get("ms_per_step")),
                             "cpu": _r((ns.get("cpu_baseline") or {}).get("value")), "gpu_over_cpu": _r(ns.get("gpu_over_cpu")),
                             "frac": _r((ns.get("roofline") or {}).get("frac")), "target_gpu_over_cpu": 10}
    if full.get("stand_ins_8spp"):
        out["stand_ins_8spp"] = [{"key": (c.get("config", {}).get("workload", "").split(" ")[0]), "value": _r(c.get("value")),
                                  "ms_per_step": _r(c.get("ms_per_step")), "cpu": _r((c.get("cpu_baseline") or {}).get("value"))}
                                 for c in full["stand_ins_8spp"]]
    if detail_path:
        out["detail"] = detail_path
    for drop in (None, "stand_ins_8spp", "weak", "configs", "north_star"):  # never reached with today's strings; the limit holds by construction
        if drop:
            out.pop(drop, None)
        if len(json.dumps(out)) < LINE_LIMIT:
            break
    return out


_REAL_STDOUT = None


def claim_stdout():
    """Point file descriptor 1 at stderr and keep the real stdout for the one JSON line: RCCL (NCCL_DEBUG), the HIP runtime and
    rocprofv3's tool library print to stdout, not always newline-terminated, and the driver parses the LAST stdout line."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _REAL_STDOUT


def emit(full, compact=True):
    """Write the full record to bench_detail.json, then the compact line as the last (and only) line of the real stdout."""
    detail = None
    if compact:
        for path in (os.path.join(REPO, "bench_detail.json"), os.path.join(REPO, "gpurun_out", "bench_detail.json")):
            if os.path.isdir(os.path.dirname(path)):
                try:
                    with open(path, "w") as f:
                        json.dump(full, f, indent=1)
                    detail = detail or os.path.relpath(path, REPO)
                except OSError:
                    pass
    line = json.dumps(compact_line(full, detail) if compact else full)
    if compact:
        assert len(line) < LINE_LIMIT, len(line)
    out = _REAL_STDOUT or sys.stdout
    out.write("\n" + line + "\n")
    out.flush()


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it: start `python -m torch.distributed.run --nproc-per-node N bench.py <the same
    arguments>` as a CHILD process (this process has not touched the GPU and never will), relay the LAST line of its stdout -- the one
    JSON line rank 0 printed -- and hand back its exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this host driver
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    for l in lines[:-1]:
        print(l, file=sys.stderr)
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    return p.returncode


def scene_line(args, pkg, torch, dist, orc, name, rank, world, local_rank, steps, warmup, min_seconds, with_cpu, digest):
    """Measure one scene: the read-back-inclusive rate (value), the rate without read-back, roofline, CPU baseline."""
    shard = tuple(int(x) for x in args.shard.split("/")) if args.shard else None
    job = Job(args, pkg, torch, dist, name, rank, world, local_rank, shard=shard, alone_steps=3 if world == 1 and args.backend == "wavefront" else 0)
    W, H = job.W, job.H
    spp = args.spp
    share = shard[1] if shard else 1  # a rank's share of a tile-sharded job renders 1 / share of the pixels
    med, regions, stats = job.measure(spp, steps, warmup, args.repeats, min_seconds, readback=True)
    med_nr, regions_nr, _ = job.measure(spp, steps, 0, max(1, min(len(regions), 3)), 0.0, readback=False)
    stats_x = job.alone_stats
    line = None
    if rank == 0:
        samples = W * H * spp * steps / share
        img = job.last_image()
        line = {
            "value": samples / med / 1e6, "unit": "Msamples/s", "ms_per_step": med / steps * 1e3, "steps": steps,
            "no_readback": {"value": samples / med_nr / 1e6, "ms_per_step": med_nr / steps * 1e3},
            "spread": {"regions": len(regions), "timed_s": float(sum(regions)), "min_ms_per_step": min(regions) / steps * 1e3,
                       "max_ms_per_step": max(regions) / steps * 1e3},
            "config": {
                "workload": f"{name} (procedural stand-in for BASELINE {STAND_IN.get(name, 'scenes')}), {W}x{H}, {spp} spp, depth {args.depth}"
                            + (f"; rank {shard[0]}'s share of a {shard[1]}-GPU pixel-tile shard (value = this rank's samples / s; no gather)" if shard else ""),
                "shape": shape_key(W, H, spp, args.depth, shard),
                "triangles": job.n_tris, "backend": args.backend, "tile": args.tile, "frames_in_flight": job.F,
                "segments_per_sample": stats["segments"] / (W * H * spp / world / share),
                "tree_build_ms": job.build_ms, "upload_plus_build_s": job.upload_build_s,
                "kernel_ms_per_step": {"k_trace_closest": stats["trace_ms"] / steps, "k_shade": stats["shade_ms"] / steps,
                                       "k_trace_shadow": stats["shadow_ms"] / steps, "k_tail": stats["tail_ms"] / steps},
                "frame_checksum": [float(img[..., :3].astype(np.float64).sum()), bool(np.isfinite(img).all()), bool((img[..., 3] == 1).all())],
            },
        }
        if args.backend == "wavefront" and stats["trace_ms"] > 0:
            line["roofline"] = roofline(job, stats, digest, stats_x, line["ms_per_step"], line["config"]["segments_per_sample"])
        if stats_x:
            # latency of ONE frame batch (reset -> 8 spp -> done, nothing else on the machine); `value` is the pipelined rate
            line["one_in_flight_ms_per_step"] = stats_x["wall_s"] / stats_x["steps"] * 1e3
            line["one_in_flight_value"] = W * H * spp / share / (stats_x["wall_s"] / stats_x["steps"]) / 1e6
            line["one_in_flight_kernel_ms_per_step"] = {"k_trace_closest": stats_x["trace_ms"] / stats_x["steps"], "k_shade": stats_x["shade_ms"] / stats_x["steps"],
                                                        "k_trace_shadow": stats_x["shadow_ms"] / stats_x["steps"], "k_tail": stats_x["tail_ms"] / stats_x["steps"]}
        if with_cpu:
            line["cpu_baseline"] = cpu_baseline(orc, job.scene, W, H, args.depth, args.cpu_seconds)
            line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
    job.close()
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--repeats", type=int, default=0, help="timed regions of K steps; 0 = until --min-seconds have been timed")
    ap.add_argument("--min-seconds", type=float, default=2.0)
    ap.add_argument("--scene", default="chess_like")
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--in-flight", type=int, default=0,
                    help="frames in flight: renderers (own stream, own path state) taking the steps in turn; 0 = 8 per GPU (16 single-stream ones for a rank of an N-GPU job).  Measured on "
                         "one GPU with 16 hardware queues since the read-back rides on the auxiliary stream (no third stream per frame) and "
                         "leaves through a one-workgroup copy: whole frame 6.91 / 6.76 / 6.63 ms per step with 4 / 6 / 8 in flight "
                         "(chess_like), 19.8 / 19.7 / 19.4 (atrium_like), 12.98 / 12.49 / 11.99 (street_like)")
    ap.add_argument("--streams-per-frame", type=int, default=0, choices=[0, 1, 2],
                    help="HIP streams of a frame in flight: 2 = main + auxiliary (shadow and tail kernels beside the next bounce), 1 = one "
                         "(PTX_DEVICE_SINGLE_STREAM); 0 = 2 for a whole frame on one GPU, 1 for thin frames: a rank's tile shard of an N-GPU job, "
                         "or fewer than 4 M path slots per step")
    ap.add_argument("--backend", default="wavefront", choices=["wavefront", "megakernel"])
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-scenes", action="store_true", help="N = 1: skip the `configs` children (BASELINE's other configs)")
    ap.add_argument("--detail-run", action="store_true",
                    help="N = 1: also measure the other stand-ins at the headline's shape (`stand_ins_8spp`) and the reference's default "
                         "scene at configs[0]'s shape -- minutes more; the default run keeps to one child per BASELINE config")
    ap.add_argument("--child", action="store_true", help="internal: a `configs` child -- print the FULL record as the one line")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --single-device lets the N > 1 code path be exercised on a 1-GPU box (testing only)")
    ap.add_argument("--single-device", action="store_true", help="testing only: every rank uses cuda:0")
    ap.add_argument("--force-gather", action="store_true",
                    help="testing only: with ONE rank, still create the process group and run the N > 1 step (pack, all_gather, unpack, "
                         "read-back) -- the RCCL branch executes on a 1-GPU box, where a 2-rank communicator on one device is refused")
    ap.add_argument("--emulate-shard", default=None, metavar="R/N",
                    help="experiments only: one process renders the tile shard of rank R of N (no gather) to see what a rank of an "
                         "N-GPU run costs; the printed line is marked and is not a benchmark result")
    ap.add_argument("--gather-readback", default="fused", choices=["fused", "separate"],
                    help="N > 1, rank 0: the gathered frame goes to the host inside the unpack kernels (ptx_unpack_shard_host) / through "
                         "ptx_unpack_shard + ptx_readback_begin as a whole-frame renderer's does (rounds 1-4)")
    ap.add_argument("--root", default="rotate", choices=["rotate", "rank0"],
                    help="N > 1: the owner of step k's frame (composes it from the gathered shards, hands it to the host) is rank k %% N / "
                         "always rank 0 (rounds 1-5)")
    ap.add_argument("--collective", default="gather", choices=["gather", "all_gather"],
                    help="N > 1, RCCL: one gather to the frame's owner (grouped send / recv: 1 / N of the bytes) / all_gather_into_tensor "
                         "(every shard to every rank, rounds 1-5)")
    ap.add_argument("--gather-unpack", default="one", choices=["one", "per-rank"],
                    help="the owner composes the frame with ONE launch over all N pieces, host-only stores when it reads back "
                         "(ptx_unpack_shards) / N ptx_unpack_shard[_host] launches (rounds 1-5)")
    ap.add_argument("--shard-accumulation", default="bound", choices=["bound", "packed"],
                    help="N > 1: k_accumulate writes the gather's message itself (ptx_bind_shard_accumulation) / row-major image + ptx_pack_shard")
    ap.add_argument("--launch-check", action="store_true",
                    help="testing: the ranks only join the process group and count themselves (no GPU work); rank 0 prints one JSON line")
    ap.add_argument("--emulate-collective", default="on", choices=["on", "off"],
                    help="experiments only: leave the collective call out of an emulated shard step (what the call itself costs)")
    ap.add_argument("--emulate-scaling", default="strong", choices=["weak", "strong"])
    ap.add_argument("--emulate-readback", default="auto", choices=["auto", "on", "off"],
                    help="experiments only: --emulate-shard steps end with the pipelined read-back (auto: with --force-gather)")
    ap.add_argument("--shard", default=None, metavar="R/N",
                    help="N = 1 only: measure rank R's share of an N-GPU pixel-tile shard of the frame (BASELINE configs[3] and [4] are "
                         "4- and 8-GPU jobs: one GPU renders one rank's tiles; the line says so and `value` is that rank's rate)")
    ap.add_argument("--traffic-json", default=None,
                    help="per-kernel HBM bytes per launch from tools/pmc_traffic.py (default: newest profiles/r*_traffic.json)")
    ap.add_argument("--dump-image", default=None, help="testing: rank 0 saves the last frame (npy)")
    ap.add_argument("--dump-frames", default=None,
                    help="testing, N > 1: rank 0 saves EVERY host frame of the job's frame store after the strong-scaling run (npy, [L, H, W, 4]): "
                         "frame j was composed by rank j %% N")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))  # before anything touches the GPU: the ranks are CHILD processes, never an exec
    claim_stdout()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: the launcher and the flag disagree", file=sys.stderr)
        sys.exit(2)
    if args.single_device:
        local_rank = 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if args.launch_check:  # the launch path alone: rendezvous, one collective, the one line -- runs without a GPU
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            emit({"launch_check": True, "n_gpus": world, "n_ranks_seen": int(t.item())}, compact=False)
        return
    torch.cuda.set_device(local_rank)
    n_ranks_seen = 1
    if world > 1 or args.force_gather:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1, dtype=torch.int64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t)  # every rank counts itself through the collective the job will use
        n_ranks_seen = int(t.item())

    pkg = graft.load_package()  # after torch: one HIP runtime in the process
    orc = graft.load_oracle() if (rank == 0 and not args.no_cpu_baseline) else None
    digest = source_digest(pkg)
    W, H = args.width, args.height
    metric = "Msamples/s (paths*spp/s) at 1920x1080, 8spp, depth 8"
    common = {"metric": metric, "unit": "Msamples/s", "n_gpus": world, "warmup": args.warmup, "higher_is_better": True,
              "vs_baseline": None,  # the reference publishes no number for this metric
              "dtype": "f32", "data": "synthetic", "source_digest": digest}

    if args.emulate_shard:  # experiments: what one rank of an N-GPU job costs, on one GPU (--force-gather: with rank 0's gather duty)
        er, ew = (int(x) for x in args.emulate_shard.split("/"))
        job = Job(args, pkg, torch, dist, args.scene, 0, 1, local_rank, shard=(er, ew))
        job_spp = args.spp * (ew if args.emulate_scaling == "weak" else 1)
        med, regions, stats = job.measure(job_spp, args.steps, args.warmup, args.repeats, args.min_seconds, readback=(args.emulate_readback == "on" or (args.emulate_readback == "auto" and args.force_gather)))
        out = dict(common, emulated_shard=args.emulate_shard, scaling=args.emulate_scaling, steps=args.steps, gather_in_loop=bool(args.force_gather),
                   host_enqueue_ms_per_step=job.host_enqueue_s / args.steps * 1e3,  # of the last region: the host's enqueue calls alone (no waits)
                   value=W * H * job_spp * args.steps / med / 1e6 / ew, ms_per_step=med / args.steps * 1e3,
                   config={"workload": f"EMULATION of rank {er} of {ew} ({args.scene}, tile shard, no gather, {job_spp} spp)",
                           "kernel_ms_per_step": {k: stats[k] / args.steps for k in ("trace_ms", "shade_ms", "shadow_ms", "tail_ms")}})
        emit(out, compact=False)
        job.close()
        return

    if world == 1 and not args.force_gather:
        line = scene_line(args, pkg, torch, dist, orc, args.scene, 0, 1, local_rank, args.steps, args.warmup, args.min_seconds,
                          not args.no_cpu_baseline, digest)
        out = dict(common, scaling="strong", **line)  # N = 1 of the strong-scaling series: the named 8-spp frame
        out["config"]["parallelism"] = "pixel-tile shard x1"
        if not args.no_extra_scenes and args.scene == "chess_like":
            # Every other line is measured in a process of its own, as `bench.py --scene NAME ...` runs it (behind another scene's
            # job in THIS process atrium_like reads 10 % lower: its 460 MB tree lands in memory the first job used and freed).
            #   stand_ins_8spp  the other stand-ins at the headline's shape (1080p, 8 spp, depth 8), the same
            #                   K steps per region (a region starts and ends with an empty ring of frames in flight)
            #   configs         BASELINE.json's five configs AS BASELINE STATES THEM -- resolution, samples per pixel and depth of
            #                   each; configs[1] is this line itself.  The multi-GPU configs run as ONE rank's share of the job on
            #                   this GPU (--shard), with a bounded number of the job's samples per pixel; configs[0] is the config
            #                   BASELINE names "CPU reference path": its cpu_baseline is that path, the GPU figure rides beside it.
            import subprocess
            mine = set(common) | {"scaling"}

            def child(extra, what, cpu=True):
                cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-extra-scenes", "--child", "--detail", str(args.detail),
                       "--tile", str(args.tile), "--backend", args.backend, "--cpu-seconds", str(min(args.cpu_seconds, CHILD_CPU_SECONDS))]
                cmd += (["--no-cpu-baseline"] if args.no_cpu_baseline or not cpu else []) + [a for a in extra if a != "--no-cpu-baseline"]
                p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
                if p.returncode != 0:
                    raise SystemExit(f"[bench] the {what} run failed (exit code {p.returncode}): {' '.join(cmd)}")
                l = json.loads(p.stdout.strip().splitlines()[-1])
                return {k: v for k, v in l.items() if k not in mine}

            if args.detail_run:
                out["stand_ins_8spp"] = []
                for name in EXTRA_SCENES:
                    out["stand_ins_8spp"].append(child(["--scene", name, "--steps", str(args.steps), "--warmup", "2", "--repeats", str(args.repeats),
                                                        "--min-seconds", str(min(args.min_seconds, 1.5)), "--width", str(W), "--height", str(H),
                                                        "--spp", str(args.spp), "--depth", str(args.depth), "--in-flight", str(args.in_flight)],
                                                       f"stand-in {name}"))
            out["configs"] = []
            for key, extra in BASELINE_CONFIGS:
                if ("default scene" in key or "--detail-run only" in key) and not args.detail_run:
                    continue
                l = child(extra + ["--repeats", "2", "--min-seconds", "0"], key, cpu="--no-cpu-baseline" not in extra)
                out["configs"].append(dict({"baseline_config": key}, **l))
            out["configs"].insert(1, {"baseline_config": "configs[1] Khronos ABeautifulGame, 1920x1080, 8 spp, depth 8 -- 1xMI355X",
                                      "is": "this line (value, roofline, cpu_baseline at the top level)"})
            # north_star's own target: ">= 10x the host-CPU Msamples/s on Intel Sponza at 1080p / 8 spp on 1 MI355X" -- its stand-in,
            # the WHOLE frame on this GPU at the headline's shape
            out["north_star"] = child(["--scene", NORTH_STAR_SCENE, "--steps", "10", "--warmup", "2", "--repeats", "2", "--min-seconds", "0",
                                       "--width", "1920", "--height", "1080", "--spp", "8", "--depth", "8"], "north_star")
        emit(out, compact=not args.child)
        return

    # ---- N > 1: strong scaling of the named frame is the metric; weak scaling beside it
    job = Job(args, pkg, torch, dist, args.scene, rank, world, local_rank, alone_steps=3 if args.backend == "wavefront" else 0)
    med_s, regions_s, stats_s = job.measure(args.spp, args.steps, args.warmup, args.repeats, args.min_seconds, readback=True)
    img = job.last_image().copy() if rank == 0 else None  # from the job's frame store: composed by the rank that owned the last step
    if rank == 0 and args.dump_frames:
        np.save(args.dump_frames, np.stack([job.frames.image(j, H, W) for j in range(job.L)]))
    weak_spp = args.spp * world
    med_w, regions_w, stats_w = job.measure(weak_spp, args.steps, 1, args.repeats, args.min_seconds, readback=True)
    if rank == 0:
        samples = W * H * args.spp * args.steps
        out = dict(common, scaling="strong", steps=args.steps, value=samples / med_s / 1e6, ms_per_step=med_s / args.steps * 1e3,
                   spread={"regions": len(regions_s), "timed_s": float(sum(regions_s)), "min_ms_per_step": min(regions_s) / args.steps * 1e3,
                           "max_ms_per_step": max(regions_s) / args.steps * 1e3},
                   weak={"scaling": "weak", "value": W * H * weak_spp * args.steps / med_w / 1e6, "ms_per_step": med_w / args.steps * 1e3,
                         "workload": f"{weak_spp} spp in total = {args.spp} spp per GPU"},
                   config={"workload": f"{args.scene} (procedural stand-in for BASELINE {STAND_IN.get(args.scene, 'scenes')}), {W}x{H}, "
                                       f"{args.spp} spp, depth {args.depth}",
                           "triangles": job.n_tris, "backend": args.backend, "tile": args.tile, "frames_in_flight": job.F,
                           "parallelism": (f"pixel-tile shard x{world}, 1 {args.collective if args.dist_backend == 'nccl' else 'gloo gather'} per step to the frame's owner "
                                           f"({'rank k % N' if job.rotate else 'rank 0'}), {'one' if args.gather_unpack == 'one' else 'N'} unpack launch(es), "
                                           f"host frames in shared memory; {job.F} frames in flight on {'one stream' if job.single_stream else 'two streams'} each"),
                           "rank0_kernel_ms_per_step": {"k_trace_closest": stats_s["trace_ms"] / args.steps, "k_shade": stats_s["shade_ms"] / args.steps,
                                                        "k_trace_shadow": stats_s["shadow_ms"] / args.steps, "k_tail": stats_s["tail_ms"] / args.steps},
                           "frame_checksum": [float(img[..., :3].astype(np.float64).sum()), bool(np.isfinite(img).all()), bool((img[..., 3] == 1).all())]})
        out["n_ranks_seen"] = n_ranks_seen
        if args.backend == "wavefront" and stats_s["trace_ms"] > 0:
            # as at N = 1: the launch ALONE on rank 0's GPU (its tile shard of the frame, before the other frames in flight exist)
            out["roofline"] = roofline(job, stats_s, digest, job.alone_stats)
            out["roofline"]["measured_on"] = f"rank 0's 1 / {world} tile shard of the frame, launch alone on its GPU"
        if args.dump_image:
            np.save(args.dump_image, img)
        if orc is not None:  # the same CPU leg as at N = 1, on rank 0's host cores while the other ranks wait in the barrier below
            out["cpu_baseline"] = cpu_baseline(orc, job.scene, W, H, args.depth, min(args.cpu_seconds, 5.0))
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    job.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:  # after the process group is gone: nothing prints behind the line
        emit(out)


if __name__ == "__main__":
    main()

"""Host-side logic of bench.py that needs no GPU: the byte models of SURVEY.md 8(d), the shape key that ties a counter summary
to a workload, and the lookup of committed summaries by scene and shape."""
import json
import os
import subprocess
import sys
import types

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def test_byte_models_of_the_survey():
    # B_trace_closest = 32 L(N) + 36 with L = ceil(log2 N) is what roofline.frac is priced on (round-5 review);
    # the 56 B of queue entry, ray and hit record ride beside it as frac_with_state
    assert bench.algorithmic_bytes_per_closest_ray(1_999_000) == 32 * 21 + 36 == 708
    assert bench.algorithmic_bytes_per_closest_ray(4_119_368) == 32 * 22 + 36 == 740
    assert bench.CLOSEST_STATE_BYTES == 56
    # worked values of the survey: N = 4 M -> B_segment = 2 * 740 + 796 = 2,276; N = 36 (the default scene) -> 1,252
    assert bench.segment_model_bytes(4_000_000) == 2276
    assert bench.segment_model_bytes(36) == 1252


def test_shape_key_is_what_profile_set_writes():
    """tools/profile_set.sh derives the same string from the bench.py arguments it is given (SHAPE=...)."""
    assert bench.shape_key(1920, 1080, 8, 8, None) == "1920x1080/8spp/d8/shard0of1"
    assert bench.shape_key(3840, 2160, 128, 16, (0, 8)) == "3840x2160/128spp/d16/shard0of8"
    script = open(os.path.join(REPO, "tools", "profile_set.sh")).read()
    start = script.index("TAG=${1:?tag}; shift")
    end = script.index("# bench.py shape_key()")
    snippet = script[start:script.index("\n", end)] + '\necho "$SCENE $SHAPE"\n'
    out = subprocess.run(["bash", "-c", snippet, "x", "tag", "--scene", "street_like", "--width", "3840", "--height", "2160", "--spp", "128", "--depth", "16",
                          "--shard", "0/8", "--in-flight", "2"], capture_output=True, text=True, check=True).stdout.split()
    assert out == ["street_like", bench.shape_key(3840, 2160, 128, 16, (0, 8))]


def test_counter_summaries_are_found_by_scene_and_shape(tmp_path, monkeypatch):
    """bench.py attaches the NEWEST committed summary whose scene and shape match the job; a summary of another shape of the
    same scene (BASELINE configs[2] is temple_like at 64 spp) is not the 8-spp line's."""
    prof = tmp_path / "profiles"
    prof.mkdir()
    docs = {"r09a_traffic.json": {"scene": "temple_like", "shape": bench.shape_key(1920, 1080, 8, 8, None), "k_generate": {"launches": 1, "hbm_bytes_per_launch": 1.0}},
            "r09a_cfg2_traffic.json": {"scene": "temple_like", "shape": bench.shape_key(1920, 1080, 64, 8, None)},
            "r09a_sq.json": {"scene": "temple_like", "shape": bench.shape_key(1920, 1080, 8, 8, None), "k_shade<false>": {"SQ_INSTS_VALU": 2.0e8}}}
    for k, (name, d) in enumerate(docs.items()):
        p = prof / name
        p.write_text(json.dumps(d))
        os.utime(p, (1000 + k, 1000 + k))
    monkeypatch.setattr(bench, "REPO", str(tmp_path))

    def job(spp, shard=None):
        j = types.SimpleNamespace()
        j.args = types.SimpleNamespace(detail=1.0, spp=spp, depth=8, traffic_json=None)
        j.world, j.W, j.H = 1, 1920, 1080
        j.shard_rank, j.shard_world = shard if shard else (0, 1)
        j.scene = types.SimpleNamespace(name="temple_like")
        return j

    doc, path = bench.counter_doc(job(8), "traffic")
    assert os.path.basename(path) == "r09a_traffic.json" and doc["shape"].startswith("1920x1080/8spp")
    doc, path = bench.counter_doc(job(64), "traffic")
    assert os.path.basename(path) == "r09a_cfg2_traffic.json"
    doc, path = bench.counter_doc(job(8), "sq")
    assert os.path.basename(path) == "r09a_sq.json"
    assert bench.counter_doc(job(8, (0, 4)), "traffic") == (None, None)  # a rank's share is another shape


def test_baseline_configs_cover_baseline_json():
    """Every config of BASELINE.json has its line(s) in bench.py's `configs` (configs[1] is the headline itself)."""
    baseline = json.load(open(os.path.join(REPO, "BASELINE.json")))
    keys = [k for k, _ in bench.BASELINE_CONFIGS]
    for i in range(len(baseline["configs"])):
        if i == 1:
            continue
        assert any(k.startswith(f"configs[{i}]") for k in keys), i
    by_key = dict(bench.BASELINE_CONFIGS)
    a = by_key[[k for k in keys if k.startswith("configs[4]")][0]]
    assert a[a.index("--width") + 1] == "3840" and a[a.index("--depth") + 1] == "16" and a[a.index("--shard") + 1] == "0/8"
    a = by_key[[k for k in keys if k.startswith("configs[3]")][0]]
    assert a[a.index("--depth") + 1] == "12" and a[a.index("--shard") + 1] == "0/4"


def _full_record():
    """A full bench record of the size that broke the round-4 driver parse (33 KB: headline + 3 stand-ins + 7 configs)."""
    full = json.load(open(os.path.join(REPO, "profiles", "r04b_bench.json")))
    assert len(json.dumps(full)) > 30000
    return full


def test_compact_line_is_under_4k_and_carries_the_contract():
    """The driver keeps ~8 KB of stdout tail and parses the LAST line: that line must stay under 4 KB and hold value,
    ms_per_step, roofline.frac, cpu_baseline.value and one flat entry per BASELINE config."""
    full = _full_record()
    line = bench.compact_line(full, "bench_detail.json")
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT == 4096, len(text)
    back = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "config"):
        assert k in back, k
    assert "vs_baseline" in back and back["vs_baseline"] is None
    assert back["config"]["workload"].startswith("chess_like") and "model" not in back["config"]
    rf = back["roofline"]
    assert rf["kernel"] == "k_trace_closest" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["traffic"] > 0 and "frac_step_counter" in rf
    assert set(rf["shade"]) >= {"frac_valu_issue", "ms_alone"}
    cb = back["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and cb["sample"]
    keys = [c["key"] for c in back["configs"]]
    for i in range(5):
        assert any(k.startswith(f"configs[{i}]") for k in keys), (i, keys)
    for c in back["configs"]:
        assert c["key"].endswith("this line") or (c["value"] > 0 and c["ms_per_step"] > 0 and set(c) >= {"key", "value", "ms_per_step", "frac_step_counter", "cpu"})
    # nothing nested deeper than roofline.shade: no `overlapped`, no per-kernel byte tables on the line
    assert "overlapped" not in rf and "step" not in rf and "spread" not in back


def test_compact_line_of_a_multi_gpu_record():
    full = _full_record()
    for k in ("stand_ins_8spp", "configs", "no_readback"):
        full.pop(k)
    full.update(n_gpus=8, weak={"scaling": "weak", "value": 1.0e4, "ms_per_step": 13.0, "workload": "64 spp in total = 8 spp per GPU"})
    full["config"]["parallelism"] = "pixel-tile shard x8, 1 gather per step to the frame's owner (rank k % N), one unpack launch(es), host frames in shared memory"
    full["n_ranks_seen"] = 8
    line = bench.compact_line(full)
    assert len(json.dumps(line)) < 2048
    assert line["n_gpus"] == 8 and line["weak"]["scaling"] == "weak" and line["cpu_baseline"]["value"] > 0 and line["roofline"]["frac"] > 0
    assert line["n_ranks_seen"] == 8


def test_compact_line_carries_north_stars_own_target():
    """BASELINE.json north_star: '>= 10x the host-CPU Msamples/s on Intel Sponza at 1080p / 8 spp on 1 MI355X' -- the line has one
    flat entry for exactly that (the stand-in's whole frame on one GPU) and still fits."""
    full = _full_record()
    child = dict(full["stand_ins_8spp"][0])
    assert child["config"]["workload"].startswith(bench.NORTH_STAR_SCENE)
    child["gpu_over_cpu"] = child["value"] / child["cpu_baseline"]["value"]
    full["north_star"] = child
    full.pop("stand_ins_8spp")
    line = bench.compact_line(full, "bench_detail.json")
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    ns = line["north_star"]
    assert ns["key"].startswith("atrium_like") and ns["value"] > 0 and ns["cpu"] > 0 and ns["gpu_over_cpu"] > 10 and ns["target_gpu_over_cpu"] == 10
    assert 0 < ns["frac"] < 1


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it (round-5 review): the process spawns torch.distributed.run as a child
    before anything touches a GPU, the ranks rendezvous on 127.0.0.1, and the parent relays rank 0's one line and the exit code.
    --launch-check stops after the ranks have counted themselves, so this runs without a GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--launch-check"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.splitlines()[-1])
    assert line == {"launch_check": True, "n_gpus": 2, "n_ranks_seen": 2}
    # a failing rank's exit code comes back through the launcher
    bad = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--launch-check", "--no-such-flag"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert bad.returncode != 0


def test_frame_store_owners_cover_every_step():
    """Step k's frame is owned by rank k % N and lives in host frame k % lcm(N, F): every frame index has exactly one owner, and
    a frame is reused only after F more steps (the ring has turned)."""
    import math
    for N, F in ((8, 8), (2, 8), (4, 8), (3, 8), (8, 2), (4, 6)):
        L = F * N // math.gcd(F, N)
        owners = {}
        for k in range(4 * L):
            j = k % L
            assert owners.setdefault(j, k % N) == k % N, (N, F, k)
        assert len(owners) == L and L >= F


def test_the_json_line_ends_stdout_whatever_else_prints(tmp_path):
    """RCCL / the HIP runtime print to file descriptor 1, not always newline-terminated: bench.py points fd 1 at stderr and
    writes the one line to the saved stdout, so `stdout.splitlines()[-1]` parses on its own."""
    script = (
        "import json, os, sys\n"
        f"sys.path.insert(0, {REPO!r})\n"
        "import bench\n"
        f"bench.REPO = {str(tmp_path)!r}\n"
        "bench.claim_stdout()\n"
        "os.write(1, b'NCCL WARN something without a newline')\n"
        "print('a python print')\n"
        f"full = json.load(open(os.path.join({REPO!r}, 'profiles', 'r04b_bench.json')))\n"
        "bench.emit(full)\n"
        "os.write(1, b'late chatter')\n")
    p = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, check=True)
    last = p.stdout.splitlines()[-1]
    assert len(last) < 4096 and len(p.stdout) < 4200
    line = json.loads(last)
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0 and line["detail"] == "bench_detail.json"
    assert "NCCL WARN" in p.stderr and "late chatter" in p.stderr
    detail = json.load(open(tmp_path / "bench_detail.json"))
    assert "overlapped" in detail["roofline"] and len(detail["configs"]) >= 5  # the full record is in the file


def test_fetch_size_factor_follows_the_measured_calibration(tmp_path):
    """tools/pmc_traffic.py: FETCH_SIZE counts half the bytes of a coalesced 16-B-per-lane stream and all the bytes of per-lane
    64-byte records (profiles/r05_fetch_size_calibration.txt), so the factor is per kernel; the all-doubled figure of rounds 1-4
    rides beside it."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import pmc_traffic

    assert pmc_traffic.fetch_factor("k_generate") == 2.0 and pmc_traffic.fetch_factor("k_accumulate") == 2.0
    assert pmc_traffic.fetch_factor("k_trace_closest") == 1.0 and pmc_traffic.fetch_factor("k_trace_shadow") == 1.0 and pmc_traffic.fetch_factor("k_shade") == 1.0
    cal = open(os.path.join(REPO, "profiles", "r05_fetch_size_calibration.txt")).read()
    assert "k_stream" in cal and "counter / known = 0.500" in cal and "k_gather64" in cal and "counter / known = 1.000" in cal
    for d, counter, rows in (("fetch", "FETCH_SIZE", (("void k_trace_closest<false>(ptd::TraceScene)", 1000.0), ("k_generate(LaunchParams, Wavefront)", 10.0))),
                             ("write", "WRITE_SIZE", (("void k_trace_closest<false>(ptd::TraceScene)", 100.0), ("k_generate(LaunchParams, Wavefront)", 2000.0)))):
        (tmp_path / d).mkdir()
        with open(tmp_path / d / "x_counter_collection.csv", "w") as f:
            f.write("Kernel_Name,Counter_Name,Counter_Value\n")
            for name, v in rows:
                f.write(f'"{name}",{counter},{v}\n')
    out = tmp_path / "t.json"
    subprocess.run([sys.executable, os.path.join(REPO, "tools", "pmc_traffic.py"), str(tmp_path / "fetch"), str(tmp_path / "write"), str(out), "chess_like"],
                   check=True, capture_output=True)
    doc = json.load(open(out))
    assert doc["k_trace_closest"]["hbm_bytes_per_launch"] == (1000.0 + 100.0) * 1024 and doc["k_trace_closest"]["hbm_bytes_per_launch_upper"] == 2100.0 * 1024
    assert doc["k_generate"]["hbm_bytes_per_launch"] == (2 * 10.0 + 2000.0) * 1024 and doc["k_generate"]["fetch_factor"] == 2.0

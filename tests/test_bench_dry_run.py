"""`python bench.py --gpus N --dry-run` (VERDICT r4 "next" item 3): the legs of an N-GPU run, the per-rank HBM plan and the expected wall time, printed without a GPU,
without torch.distributed and without launching anything -- what an operator reads before giving an 8-GPU node to the run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plan(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"] + list(extra), capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout)


def test_dry_run_eight_gpus():
    d = _plan("--gpus", "8")
    assert d["dry_run"] is True and d["nothing_was_run"] is True and d["n_gpus"] == 8 and "torch.distributed.run" in d["launch"]
    legs = [l["leg"] for l in d["legs_in_order"]]
    assert legs[0].startswith("headline") and "config4" in legs and "strong" in legs and any(l.startswith("ntt distributed") for l in legs) and any(l.startswith("mctx") for l in legs)
    by = {l["leg"]: l for l in d["legs_in_order"]}
    assert by["config4"]["per_gpu_points"] == 2 ** 23 and by["strong"]["per_gpu_points"] == 2 ** 21 and by["headline (weak)"]["per_gpu_points"] == 2 ** 24
    assert 0 < d["hbm_total_per_rank_gb"] < d["hbm_capacity_gb"] == 288
    m = d["hbm_plan_per_rank_bytes"]["msm pipeline scratch"]
    assert m["window_bits"] == 19 and m["windows"] == 14 and m["entries"] == 14 * 2 ** 24
    assert 5 < d["expected_wall_s"] < 600


def test_dry_run_single_gpu_and_small():
    d = _plan("--gpus", "1", "--log-n", "20", "--ntt-log-n", "20", "--groth16-k", "64")
    assert d["n_gpus"] == 1 and d["launch"] == "python bench.py"
    assert not any(l["leg"] in ("config4", "strong") for l in d["legs_in_order"])
    assert d["hbm_total_per_rank_gb"] < 20


def test_host_cores_respects_affinity_and_cgroup_quota():
    """bench.py's cpu_baseline threads = the CPUs this process may really use (affinity mask capped by the cgroup quota), never the online count alone"""
    sys.path.insert(0, ROOT)
    import bench

    n = bench.host_cores()
    assert 1 <= n <= (os.cpu_count() or 1) and n <= len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            assert n <= max(1, int(q) // int(per))
    except FileNotFoundError:
        pass

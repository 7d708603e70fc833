"""bench.py end to end at small sizes (every leg, every self-check), so that a change that breaks the bench line is caught by the test
suite and not by the driver's measurement run.  Also exercises the N = 2 code path with two gloo ranks sharing the GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(out: str) -> dict:
    rows = [ln for ln in out.strip().splitlines() if ln.startswith("{")]
    assert len(rows) == 1, out[-2000:]
    return json.loads(rows[0])


def test_bench_single_gpu_small():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--log-n", "16", "--ntt-log-n", "14", "--groth16-k", "4", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["unit"] == "points/s" and d["value"] > 0 and d["n_gpus"] == 1 and d["steps"] == 3 and d["higher_is_better"] is True
    assert d["config"]["workload"] == "bls12_381_g1_msm_2^16_per_gpu" and d["config"]["precomputed_table"].startswith("none")
    for key in ("roofline", "cpu_baseline", "pcie_inclusive", "msm_fixed_key", "msm_skewed_scalars", "ntt", "groth16"):
        assert isinstance(d[key], dict), key
    assert "leg_errors" not in d
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["kernel"] == "k_msm_accumulate"
    assert d["cpu_baseline"]["parity_full_size"] is True and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert d["ntt"]["cpu_baseline"]["parity_full_size"] is True and d["groth16"]["cpu_baseline"]["parity_full_size"] is True
    assert d["groth16"]["verified"] is True and d["msm_fixed_key"]["table_build_ms"] > 0


def test_bench_two_ranks_gloo_one_gpu():
    env = dict(os.environ, ZL_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log-n", "16", "--ntt-log-n", "14", "--groth16-k", "4", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["parallelism"] == "shard2"
    assert d["groth16"]["n_gpus"] == 2 and d["groth16"]["verified"] is True

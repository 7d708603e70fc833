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
    for key in ("roofline", "cpu_baseline", "pcie_inclusive", "msm_fixed_key", "msm_skewed_scalars", "ntt", "groth16", "configs"):
        assert isinstance(d[key], dict), key
    # all five BASELINE configurations are on the line
    c = d["configs"]
    assert c["1"]["checked_exactly"] and c["1"]["cpu_baseline"]["parity_full_size"] is True and c["1"]["gpu_single_call_ms"] > 0
    assert c["2"]["checked_exactly"] and c["2"]["single_call_ms"] > 0 and 0 < c["2"]["int_alu_frac_single_call"] < 1
    assert c["4"]["functional"] is True and c["4"]["checked_exactly"] is True
    assert isinstance(c["3"], str) and isinstance(c["5"], str)
    # (the live PMC passes start rocprofv3 child processes: a profiler that cannot start on this box is reported in leg_errors and the committed passes are quoted --
    # that alone does not fail the bench line; everything else must be clean)
    live_failed = [k for k in d.get("leg_errors", {}) if k.startswith("live_traffic")]
    assert set(d.get("leg_errors", {})) == set(live_failed), d.get("leg_errors")
    assert d["scaling_model"] is None  # needs --log-n >= 22 (prefixes 2^21 .. 2^23 of the headline's inputs): covered by test_bench_scaling_model
    ia = d["roofline"]["int_alu"]
    assert 20 < ia["peak"] < 120 and ia["peak_constant_operands"] == 78.6 and ia["frac"] > ia["frac_vs_constant_operand_peak"] * 0.9
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["kernel"] == "k_msm_accumulate"
    assert d["cpu_baseline"]["parity_full_size"] is True and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert d["ntt"]["cpu_baseline"]["parity_full_size"] is True and d["groth16"]["cpu_baseline"]["parity_full_size"] is True
    assert d["groth16"]["verified"] is True and d["msm_fixed_key"]["table_build_ms"] > 0
    # roofline.traffic is a measurement of THIS run (child processes under rocprofv3 --pmc after the timed legs), for both kernels
    rf, nrf = d["roofline"], d["ntt"]["roofline"]
    if live_failed:
        pytest.skip(f"rocprofv3 child processes failed on this box: {d['leg_errors']}")
    assert rf["traffic_detail"]["how"].startswith("measured in this run") and rf["traffic"] >= 0.5 * rf["algorithmic_bytes"] and rf["traffic_detail"]["window_bits"] == d["config"]["window_bits"]
    assert nrf["traffic_detail"]["how"].startswith("measured in this run") and nrf["traffic"] >= 64.0 * (1 << 14)


def test_bench_scaling_model():
    """N = 1 at 2^22: the line carries the single-GPU times of the 2^21 prefix (exact) and the model's fields; the 2^24-specific efficiencies need the
    full size and are exercised by the driver's default run."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--log-n", "22", "--steps", "3", "--warmup", "1", "--no-ntt", "--groth16-k", "0", "--no-cpu", "--no-skew",
                        "--fixed-key", "-1", "--no-live-traffic"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    m = d["scaling_model"]
    assert m["checked_exactly"] is True and m["measured_ms_per_msm_pipelined"]["2^21"] > 0 and m["measured_ms_per_msm_pipelined"]["2^22"] == d["ms_per_step"]
    assert "PREDICTED" in m["note"] and "leg_errors" not in d


def _check_two_rank_line(d):
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["parallelism"] == "shard2"
    assert d["groth16"]["n_gpus"] == 2 and d["groth16"]["verified"] is True
    assert "secondary_legs_error" not in d
    legs = d["scaling_legs"]
    for name in ("weak", "config4", "strong"):
        e = legs[name]
        assert "error" not in e, e
        assert e["checked_exactly"] is True and e["ms_per_step"] > 0 and e["points_per_s"] > 0 and e["interference_ratio"] > 0
    # fixed-shard legs: the solo / all-rank ratio IS the weak-scaling efficiency; the strong leg carries T_1(total) / (N x T_N)
    assert legs["weak"]["weak_scaling_efficiency"] == legs["weak"]["interference_ratio"] and "weak_scaling_efficiency" in legs["config4"]
    st = legs["strong"]
    assert "weak_scaling_efficiency" not in st and st["single_gpu_total_ms_per_step"] > 0
    assert abs(st["strong_scaling_efficiency"] - st["single_gpu_total_ms_per_step"] / (2 * st["ms_per_step"])) < 1e-9
    assert legs["config4"]["points_total"] == 2 ** 18 and legs["strong"]["points_total"] == 2 ** 16
    # the one-process transport, timed by a child process; both test ranks share GPU 0, so the exchange runs on virtual ranks
    m = d["mctx"]
    assert "error" not in m, m
    assert m["n_gpus"] == 2 and m["rccl_ranks"] == 0 and m["msm"]["checked_exactly"] is True and m["ntt"]["forward_ms"] > 0


_SMALL_N2 = ["--gpus", "2", "--log-n", "16", "--ntt-log-n", "14", "--groth16-k", "4", "--steps", "2", "--warmup", "1", "--config4-log-total", "18",
             "--strong-log-total", "16"]


def test_bench_two_ranks_gloo_one_gpu():
    env = dict(os.environ, ZL_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
                        os.path.join(ROOT, "bench.py")] + _SMALL_N2, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    _check_two_rank_line(_line(r.stdout))


def test_bench_bare_launch_self_execs_under_torchrun():
    """`python bench.py --gpus 2` with no torchrun environment (how a user, or a driver without the launcher line, starts it) must re-launch
    itself one rank per GPU instead of exiting."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ZL_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + _SMALL_N2, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "re-launching" in r.stderr
    _check_two_rank_line(_line(r.stdout))


def test_bench_mctx_transport_virtual_ranks():
    """--transport mctx: ONE process, zl_ctx_create_multi + zl_msm_sharded / zl_ntt_sharded; 4 virtual ranks on GPU 0 (RCCL needs distinct devices)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--transport", "mctx", "--gpus", "4", "--mctx-devices", "0,0,0,0", "--log-n", "14",
                        "--ntt-log-n", "12", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 4 and d["value"] > 0 and d["mctx"]["rccl_ranks"] == 0 and d["mctx"]["msm"]["checked_exactly"] is True
    assert d["mctx"]["ntt"]["log_n"] == 14 and d["config"]["parallelism"] == "mctx4"


def test_bench_one_rank_nccl_group_runs_every_collective():
    """VERDICT r4 missing #1: the torch `nccl` (= RCCL) branch of the one-process-per-GPU path had never executed -- the N = 2 tests above use gloo, which takes
    the host-tensor side of openzl_amd/sharded.py.  ZL_FORCE_COLLECTIVE=1 makes the N = 1 run create a ONE-rank nccl process group and send the headline's gate
    and every timed step through all_gather_into_tensor on DEVICE tensors, its barriers / MAX all_reduce through the group, and one NTT through sharded_ntt's
    all_to_all_single with the stream fences -- each result checked exactly as in an N > 1 run.  No hardware scaling claim: first contact with RCCL only."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ZL_DIST_BACKEND")}
    env["ZL_FORCE_COLLECTIVE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--log-n", "18", "--ntt-log-n", "16", "--groth16-k", "0", "--steps", "3", "--warmup", "1", "--no-cpu",
                        "--no-configs", "--no-skew", "--no-pcie", "--fixed-key", "-1", "--no-live-traffic"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    c = d["collective"]
    assert c["backend"] == "nccl" and c["ranks"] == 1 and c["tensors"].startswith("device")
    a2a = c["all_to_all_single"]
    assert "error" not in a2a, a2a
    assert a2a["log_n"] == 16 and a2a["forward_ms"] > 0 and a2a["self_check"].startswith("iNTT(NTT(x)) == x")
    assert d["n_gpus"] == 1 and d["value"] > 0 and "leg_errors" not in d and d["pcie_inclusive"] is None


def test_sharded_helpers_through_one_rank_nccl_group(backend):
    """The same first contact below bench.py: sharded_msm / sharded_msm_batch / sharded_ntt of openzl_amd/sharded.py over a one-rank nccl group, bit for bit
    against the single-device entry points (MSM vs zl_msm_dev's affine result, NTT vs zl_ntt_dev), in THIS process."""
    import numpy as np
    import torch
    import torch.distributed as dist

    import oracle_lib as ol
    from oracle_lib import po
    from openzl_amd import ZL_BLS12_381
    from openzl_amd.sharded import DeviceNttEngine, fold_partials, sharded_msm, sharded_msm_batch, sharded_ntt

    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    old = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "ZL_FORCE_COLLECTIVE")}
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", ZL_FORCE_COLLECTIVE="1")
    dev = torch.device("cuda", 0)
    try:
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
        curve = po.BLS12_381
        n = 1 << 12
        k = ol.random_scalars(curve, n, 11)
        s = ol.random_scalars(curve, n, 12)
        h = backend.bases_upload(curve.cid, ol.oracle_g1_mul_gen(curve, k))
        d_s = torch.from_numpy(s.view(np.int64)).to(dev)
        exp, einf = backend.msm_dev(h, d_s.data_ptr(), n)
        got, inf = sharded_msm(lambda: backend.msm_partial_dev(h, d_s.data_ptr(), n), ZL_BLS12_381, device=dev)
        assert inf == einf and (np.asarray(got) == np.asarray(exp)).all()
        parts = backend.msm_batch_partial_dev(h, [d_s.data_ptr()] * 3, n)
        for xy, i2 in sharded_msm_batch(parts, ZL_BLS12_381, device=dev):
            assert i2 == einf and (np.asarray(xy) == np.asarray(exp)).all()
        backend.bases_free(h)
        x = ol.random_scalars(curve, 1 << 12, 13)
        for inverse in (False, True):
            for coset in (False, True):
                t = torch.from_numpy(x.view(np.int64).copy()).to(dev)
                out = sharded_ntt(DeviceNttEngine(backend, ZL_BLS12_381), t, 12, inverse=inverse, coset=coset, mont=False)
                ref = backend.ntt(curve.cid, x, inverse=inverse, coset=coset)
                assert (out.cpu().numpy().view(np.uint64) == ref).all(), (inverse, coset)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        for kk, v in old.items():
            if v is None:
                os.environ.pop(kk, None)
            else:
                os.environ[kk] = v

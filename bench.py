#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X MSM / NTT backend (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log-n 24] [--no-cpu] [--no-ntt]
    python bench.py --gpus N                       # N > 1 launched bare: re-executes itself under torch.distributed.run (one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus N --transport mctx      # ONE process, zl_ctx_create_multi + zl_msm_sharded / zl_ntt_sharded (RCCL inside the library)

A "step" is one variable-base MSM over 2^log_n random BLS12-381 G1 points and random scalars < r per GPU, bases and
scalars already resident in HBM, through the C ABI (zl_msm_batch_partial_dev / zl_msm_partial_dev), with NO per-key precomputation:
`value` is what multi_scalar_mul(bases, scalars) is.  Every timed step is checked exactly at full size against (sum s_i k_i) G
(the bases have known discrete logs).  With N > 1 every rank owns its own shard of
bases/scalars (weak scaling), the per-rank partial sums (512 B) are all-gathered over RCCL and folded
on every rank (zl_partials_sum).  Rank 0 prints ONE JSON line.  Extra objects on that line:
  roofline      HBM roofline of the dominant kernel (k_msm_accumulate): algorithmic bytes (128 B / point, SURVEY.md
                §8d) / its HIP-event duration on the backend's stream, vs 8 TB/s; .int_alu = the integer-multiply roofline
                that actually binds; .traffic = PMC bytes of the same configuration (profiles/)
  cpu_baseline  the CPU oracle (arkworks-algorithm restatement, NOT the arkworks binary) timed on this box's cores on the same input
                (N = 1 only): all-core (chunk x window) grid on the complete input (must equal the GPU result bit for bit), window-parallel
                (arkworks `parallel`) and single-threaded (the reference's configuration) on bounded samples; value = the faster multi-threaded one
  configs       BASELINE.json's five configurations on this line: "1" 2^16 BN254 (GPU + CPU oracle, equal), "2" 2^20 BLS12-381 (single call /
                pipelined, integer-ALU fraction of the whole call), "3" -> ntt, "4" 2^26 as 8 shards (N = 1: 8 virtual ranks through
                zl_msm_sharded, functional; N > 1: scaling.config4), "5" -> groth16
  scaling_legs  N > 1: weak (2^log_n per GPU = the headline), config4 (2^26 points over the N ranks) and strong (2^24 points over the N ranks),
                each checked exactly against the all-shard dot product and each with interference_ratio = (time of the same per-GPU
                MSM on rank 0 alone, measured in this process) / (time with all ranks) -- which is the weak-scaling efficiency of the weak and config-4
                legs; the strong leg also carries strong_scaling_efficiency = T_1(total) / (N x T_N) with T_1 = the whole 2^24-point MSM on rank 0 alone
  scaling_model N = 1: single-GPU ms per MSM at 2^21 .. 2^24 (exact) and the weak / config-4 / strong efficiencies they imply at N = 2, 4, 8
                (the exchange is 512 B per rank); a prediction, labelled so, not a hardware claim
  mctx          N > 1: the one-process transport (zl_ctx_create_multi + zl_msm_sharded / zl_ntt_sharded) timed by a child process of
                rank 0 after the ranks are done; .rccl_ranks = size of the RCCL communicator the library created
  pcie_inclusive    the same MSM with the scalars coming from host memory (zl_msm); never `value`
  msm_fixed_key     the precomputed-table mode (zl_bases_precompute, c = 22) with its build time, bytes and break-even count; never `value`
  msm_skewed_scalars  the same MSM on Groth16-witness-like scalars (N = 1 only)
  ntt           2^24 BLS12-381 Fr forward+inverse NTT throughput (second half of the BASELINE metric) + its own roofline and cpu_baseline
                (the CPU oracle's transform of the same vector must equal the GPU's bit for bit); N > 1: per-GPU replicas
                and .distributed = ONE 2^(24 + log2 N) transform over all ranks with a single RCCL all-to-all
  groth16       config 5 (Poseidon-chain circuit, 958 465 constraints) prove time / constraints per second, proof verified, + cpu_baseline
                (the CPU oracle's proof from the same key / witness / (r, s) must equal the GPU proof bit for bit);
                N > 1: one independent proof per GPU (replicas)
The secondary legs of an N > 1 run execute after the MSM measurement under a watchdog, so a stall there cannot cost the MSM line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np


ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

R_BLS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
MADS_PER_MIXED_ADD = 3542  # ISA of k_msm_accumulate<BlsG1>: 6 products x 392 + 2 squarings x 301 + one dual product scan x 588 v_mad_u64_u32
MADS_PER_MUL = 392
MULS_PER_MIXED_ADD = MADS_PER_MIXED_ADD / MADS_PER_MUL  # 9.04 multiplication-equivalents
FQ_MUL_PEAK_CONST_G = 78.6  # rounds 2-3's ceiling: the multiplier of zl_field28.h alone at the accumulate kernel's occupancy, 3 waves/SIMD (166 registers; 68.4 / 76.6 / 78.6 / 80.0 / 80.6 G/s at 1..5 waves, tools/fbench28_asm.hip, profiles/r02_fbench28_asm_occupancy.log; rounds 1 and early 2 used 74.3 = two waves on a slower box)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
_FQ_MUL_PEAK_LIVE = {}
_ORACLE = {"cflags": "-O3 -march=x86-64-v3 -fopenmp (portable build)"}


def oracle_native():
    """cpu_baseline legs time the oracle compiled ON THIS BOX with -march=native; the flags travel on the JSON line."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol

    _ORACLE["cflags"] = ol.select_native()


def fq_mul_peak_live(be) -> float:
    """10^9 Fq products / s of the accumulation kernel's multiplier ALONE at the kernel's occupancy (3 waves / SIMD) on per-lane pseudo-random operands,
    measured on this box in this run in STEADY STATE (zl_test_fq_mul_rate: one warm-up launch, then the median of three launches of ~0.2 s each --
    MI355X clocks ramp up over tens of milliseconds after an idle gap, a 12-ms launch on its own under-reads by 10-15 %).  Rounds 2-3 used a constant,
    78.6 G/s, taken by tools/fbench28_asm.hip on hipMemset operands; round 4 measured both kinds of operands side by side in steady state
    (profiles/r04_fbench_f64.log): constant operands run 11 % faster than random field elements (78.0 against 69.9 G/s in that binary; the chip clocks to
    its power budget and identical lanes toggle less), and the boxes of the pool differ by +-4 %: the ceiling is therefore measured where and when the
    kernel is."""
    if "v" not in _FQ_MUL_PEAK_LIVE:
        from openzl_amd.backend import hook_fq_mul_rate

        hook_fq_mul_rate(be, 3, 20000)
        _FQ_MUL_PEAK_LIVE["v"] = float(np.median([hook_fq_mul_rate(be, 3, 50000) for _ in range(3)]))
    return _FQ_MUL_PEAK_LIVE["v"]


def random_scalars_lt_r(n: int, seed: int, r: int = R_BLS, bits: int = 255) -> np.ndarray:
    """n uniform scalars < r as (n,4) uint64 little-endian limbs (vectorised rejection sampling)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    r_l = np.array([(r >> (64 * j)) & (2**64 - 1) for j in range(4)], dtype=np.uint64)
    out = np.zeros((n, 4), dtype=np.uint64)
    todo = np.arange(n)
    top_mask = np.uint64((1 << (bits - 192)) - 1)
    while todo.size:
        cand = rng.integers(0, 1 << 64, size=(todo.size, 4), dtype=np.uint64)
        cand[:, 3] &= top_mask
        lt = np.zeros(todo.size, dtype=bool)
        decided = np.zeros(todo.size, dtype=bool)
        for j in (3, 2, 1, 0):
            less = (cand[:, j] < r_l[j]) & ~decided
            more = (cand[:, j] > r_l[j]) & ~decided
            lt |= less
            decided |= less | more
        out[todo[lt]] = cand[lt]
        todo = todo[~lt]
    return out


def live_pmc_traffic(what: str, log_n: int):
    """HBM traffic of the dominant kernel measured IN THIS RUN: two child processes of the same single-call command the committed PMC files come from
    (tools/msm_one.py / tools/ntt_one.py), each under `rocprofv3 --pmc <one counter> --kernel-trace` -- FETCH_SIZE and WRITE_SIZE in separate passes, no other
    trace domain, as /opt/skills/guides/MI355X_MICROARCH.md prescribes -- folded by tools/pmc_fold.py (counter x 1024 B; 2 x FETCH_SIZE + WRITE_SIZE on gfx950).
    The parent process is idle meanwhile (all timed legs are over).  Returns (bytes per launch | per transform, detail dict); raises on any failure."""
    import glob
    import shutil
    import subprocess
    import tempfile

    if any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ):
        raise RuntimeError("this process runs under rocprofv3 itself")
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        raise RuntimeError("rocprofv3 not found")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_fold

    cmd = [sys.executable, os.path.join(ROOT, "tools", "msm_one.py"), str(log_n), "0", "-1", "1"] if what == "msm" else [sys.executable, os.path.join(ROOT, "tools", "ntt_one.py"), str(log_n), "2"]
    got, child_out = {}, ""
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="zl_pmc_", dir="/tmp")
        try:
            r = subprocess.run([rp, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "t", "-f", "csv", "--"] + cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"},
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
            child_out = r.stdout.decode(errors="replace")
            csvs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode or not csvs:
                raise RuntimeError(f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {child_out[-200:]}")
            got[counter] = pmc_fold.rows(csvs[0], counter)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    f, w = got["FETCH_SIZE"], got["WRITE_SIZE"]
    how = f"measured in this run: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace (separate passes) -- python {' '.join(os.path.relpath(c, ROOT) if os.path.isabs(c) and c.startswith(ROOT) else c for c in cmd[1:])}"
    if what == "msm":
        fm, wm = pmc_fold.per_kernel_max(f), pmc_fold.per_kernel_max(w)
        acc = [k for k in set(fm) | set(wm) if k.startswith("k_msm_accumulate")]
        if not acc:
            raise RuntimeError("no k_msm_accumulate dispatch in the counter file")
        import re
        m = re.findall(r"c=(\d+)", child_out)
        return sum(2.0 * fm.get(k, 0.0) + wm.get(k, 0.0) for k in acc), {"how": how, "window_bits": int(m[-1]) if m else None,
                                                                            "FETCH_SIZE_bytes": sum(fm.get(k, 0.0) for k in acc), "WRITE_SIZE_bytes": sum(wm.get(k, 0.0) for k in acc)}
    P = 1 if log_n <= 10 else min(4, (log_n + 7) // 8)
    fp, wp = [v for _, k, v in f if k.startswith("k_ntt_pass")], [v for _, k, v in w if k.startswith("k_ntt_pass")]
    if len(fp) < 2 * P or len(wp) < 2 * P:
        raise RuntimeError("fewer NTT pass dispatches than one forward + inverse transform in the counter file")
    return 2.0 * sum(fp[-2 * P:-P]) + sum(wp[-2 * P:-P]), {"how": how, "passes": P, "FETCH_SIZE_bytes": sum(fp[-2 * P:-P]), "WRITE_SIZE_bytes": sum(wp[-2 * P:-P])}


def host_cores() -> int:
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a container on a 256-thread host with cpu.max = 16 CPUs runs 256
    OpenMP threads on 16 of them -- what os.cpu_count() cannot see; the round-4 `all_core_grid` leg did exactly that)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:  # noqa: BLE001
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:  # noqa: BLE001
            pass
    return max(1, n)


def limbs_to_int(row) -> int:
    return sum(int(v) << (64 * j) for j, v in enumerate(row))


def cpu_baseline(be, h, k64, s_host, gpu_xy, threads_req: int, full_log_n: int):
    """Time the CPU oracle (oracle/libzl_oracle.so: plain-C restatement of the arkworks 0.3.0 algorithm, 'port') on the host cores of this
    box, on the SAME bases and scalars the GPU was timed on (bases downloaded from the device, canonical affine).  Checker code used as a
    reported baseline only -- never on the product path.  Three configurations:
      all-core (chunk x window) grid (value): the FULL input is cut into chunks so that chunks x windows ~ 4 x threads (threads = host_cores(): affinity capped by the cgroup quota), every (chunk, window)
          pair is one task of ark's window routine, partial sums are added -- uses every core on the same algorithm (not an arkworks
          configuration; NOT necessarily the fastest: on a many-core box the window-parallel sample below can beat it, `value` is the best of the two); its result must equal the GPU's full-size result bit for bit;
      window-parallel: what arkworks' `parallel` feature does (one thread per window), on a 2^22 sample with the window width ark's rule
          gives the full input;
      single thread: the reference's actual configuration (no `parallel` feature), on a 2^18 sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from oracle_lib import po

    curve = po.BLS12_381
    n = s_host.shape[0]
    avail = host_cores()
    threads = max(1, min(threads_req or avail, avail))
    t0 = time.perf_counter()
    bases = be.bases_download(h)  # canonical affine x||y, the very points the GPU used
    t_dl = time.perf_counter() - t0
    xy, inf, sec_all = ol.oracle_msm_g1_timed(curve, bases, s_host, algo=2, threads=threads)
    parity = bool(inf == 0 and (xy == np.asarray(gpu_xy)).all())
    if not parity:
        raise SystemExit("full-size parity check failed: the CPU oracle's MSM of the complete input differs from the GPU result")
    c_full = po.ark_window_bits(n)
    m = min(n, 1 << 22)
    windows = (255 + c_full - 1) // c_full
    wt = max(1, min(threads, windows))
    _, _, sec_win = ol.oracle_msm_g1_timed(curve, bases[:m], s_host[:m], algo=0, threads=wt, c_override=c_full)
    m1 = min(n, 1 << 18)
    _, _, sec_one = ol.oracle_msm_g1_timed(curve, bases[:m1], s_host[:m1], algo=0, threads=1)
    del bases
    grid_rate, win_rate = n / sec_all, m / sec_win
    best_is_grid = grid_rate >= win_rate
    return {
        "value": max(grid_rate, win_rate),  # the faster of the two multi-threaded arrangements measured below
        "unit": "points/s",
        "cores": threads if best_is_grid else wt,
        "kind": "port",
        "cflags": _ORACLE["cflags"],
        "value_is": "all_core_grid" if best_is_grid else "window_parallel",
        "all_core_grid": {"value": grid_rate, "cores": threads},
        "sample": f"the complete 2^{full_log_n} BLS12-381 G1 input of the GPU run (same bases, same scalars) as a (chunk x window) task grid (~4 tasks per thread) over {threads} "
                  f"threads (ark's window routine per task, ark window rule for the chunk length, partials added); arkworks-algorithm "
                  f"restatement in C, not the arkworks binary; {sec_all:.2f} s of wall time",
        "parity_full_size": parity,
        "parity_note": "the CPU result over the complete input equals the GPU result bit for bit (canonical affine coordinates)",
        "window_parallel": {"value": m / sec_win, "cores": wt, "sample": f"2^{m.bit_length() - 1} prefix with the full input's window width c={c_full} "
                            f"({windows} windows, one thread each: what arkworks' `parallel` feature does; the sample over-weights the bucket "
                            f"reduction by ~{100.0 * 2 * (1 << c_full) / m:.0f} % relative to the full input)"},
        "single_thread": {"value": m1 / sec_one, "cores": 1, "sample": f"2^{m1.bit_length() - 1} prefix, ark window rule for that size; 1 thread = the reference's "
                          "actual configuration (no `parallel` feature, plugins/arkworks/Cargo.toml)"},
        "host_cpus": avail, "host_cpus_online": os.cpu_count(),
        "bases_download_s": t_dl,
    }


def cpu_baseline_ntt(x_mont: np.ndarray, threads_req: int):
    """CPU oracle NTT (in-order radix-2 with a tabulated root table, as ark-poly 0.3.0) on the GPU run's input: all cores on the full
    vector (value), one thread (the reference's configuration) on a 2^22 prefix."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from oracle_lib import po

    curve = po.BLS12_381
    n = x_mont.shape[0]
    avail = host_cores()
    threads = max(1, min(threads_req or avail, avail, 64))
    X, sec_f = ol.oracle_ntt_timed(curve, x_mont, inverse=False, threads=threads)
    back, sec_i = ol.oracle_ntt_timed(curve, X, inverse=True, threads=threads)
    if not (back == x_mont).all():
        raise SystemExit("CPU oracle NTT round trip failed")
    m = min(n, 1 << 22)
    _, sec_1 = ol.oracle_ntt_timed(curve, np.ascontiguousarray(x_mont[:m]), inverse=False, threads=1)
    return {"value": 2.0 * n / (sec_f + sec_i), "unit": "elements/s (forward + inverse)", "cores": threads, "kind": "port", "cflags": _ORACLE["cflags"],
            "sample": f"the complete 2^{n.bit_length() - 1} vector of the GPU run, forward then inverse, butterflies of every stage split over {threads} threads",
            "forward_s": sec_f, "inverse_s": sec_i,
            "single_thread": {"value": m / sec_1, "unit": "elements/s (forward)", "cores": 1, "sample": f"2^{m.bit_length() - 1} prefix, forward"}}, X


def cpu_baseline_groth16(be, keys, circ, gpu_proof, r, s, threads_req: int):
    """The CPU oracle's Groth16 prover (witness map + 4 G1 MSMs + 1 G2 MSM, window-parallel MSMs = arkworks `parallel`) on the SAME proving
    key (downloaded from the device), R1CS, witness and blinding scalars: must produce the GPU's proof bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import groth16_util as gu
    from oracle_lib import po
    from openzl_amd import ZL_G2  # noqa: F401

    curve = po.BLS12_381
    avail = host_cores()
    threads = max(1, min(threads_req or avail, avail, 32))
    arrays = circ.arrays()
    pk = keys.pk_dict()
    for name in ("a_query", "b_g1_query", "h_query", "l_query", "b_g2_query"):
        pk[name] = be.bases_download(pk[name])
    t0 = time.perf_counter()
    cpu_proof, _ = gu.oracle_prove(curve, arrays, arrays["assignment"], pk, np.ascontiguousarray(r), np.ascontiguousarray(s), threads=threads)
    sec = time.perf_counter() - t0
    same = all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(cpu_proof, gpu_proof))
    if not same:
        raise SystemExit("full-size parity check failed: the CPU oracle's Groth16 proof differs from the GPU proof")
    n_c = arrays["n_constraints"]
    return {"value": n_c / sec, "unit": "constraints/s", "cores": threads, "kind": "port", "cflags": _ORACLE["cflags"], "prove_s": sec, "parity_full_size": True,
            "sample": f"the complete config-5 circuit ({n_c} constraints), same proving key / witness / (r, s) as the GPU proof; window-parallel MSMs over "
                      f"{threads} threads (arkworks `parallel`), single-threaded NTTs; arkworks-algorithm restatement in C, not the arkworks binary"}


def skewed(s):
    """Groth16-witness-like scalars: 50 % zeros, 25 % ones, the rest unchanged, shuffled."""
    n = s.shape[0]
    s2 = s.copy()
    s2[: n // 2] = 0
    s2[n // 2: 3 * n // 4] = np.array([1, 0, 0, 0], dtype=np.uint64)
    perm = np.random.Generator(np.random.PCG64(11)).permutation(n)
    return np.ascontiguousarray(s2[perm])


def groth16_replica_leg(be, torch, k):
    """One rank's Groth16 prove of the config-5 circuit (k chained Poseidon hashes): -> (constraints, mean prove seconds); verified."""
    from openzl_amd import ZL_BLS12_381, Circuit, Groth16Keys

    circ = Circuit(ZL_BLS12_381, k)
    keys = Groth16Keys(be, circ, seed=0x5EED0006)
    try:
        keys.prove(seed=7)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            proof, _, _ = keys.prove(seed=7)
            ts.append(time.perf_counter() - t0)
        if not keys.verify(proof, circ.arrays()["assignment"][1:2]):
            raise RuntimeError("proof does not verify")
        return circ.shape[0], float(np.mean(ts))
    finally:
        keys.close()
        circ.close()


def distributed_ntt_leg(be, dist, torch, dev, rank, world, log_m):
    """One 2^(log_m + log2 world)-point transform over all ranks (weak scaling: 2^log_m elements per GPU): cross step ->
    RCCL all_to_all_single -> local transform, and back.  Timed per transform with barriers, max over ranks."""
    from openzl_amd import ZL_BLS12_381
    from openzl_amd.sharded import DeviceNttEngine, sharded_ntt

    log_g = world.bit_length() - 1  # (world == 1 under ZL_FORCE_COLLECTIVE=1: log_g = 0, the exchange is a one-rank all_to_all_single)
    log_n = log_m + log_g
    eng = DeviceNttEngine(be, ZL_BLS12_381)
    x = random_scalars_lt_r(1 << log_m, 5000 + rank)  # this rank's block-column slice (any residues < r are valid Montgomery limbs)
    f_ms, i_ms = [], []
    cur = torch.from_numpy(x.view(np.int64)).to(dev)
    for it in range(1 + 3):
        for inverse, acc in ((False, f_ms), (True, i_ms)):
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            cur = sharded_ntt(eng, cur, log_n, inverse=inverse, mont=True)
            torch.cuda.synchronize()
            dist.barrier()
            if it:
                acc.append((time.perf_counter() - t0) * 1e3)
    ok = bool((cur.cpu().numpy().view(np.uint64) == x).all())
    tt = torch.tensor([float(np.mean(f_ms)), float(np.mean(i_ms)), 0.0 if ok else 1.0], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if tt[2].item() != 0.0:
        raise RuntimeError("distributed NTT self-check failed: iNTT(NTT(x)) != x")
    f, i = float(tt[0].item()), float(tt[1].item())
    tot = float(1 << log_n)
    sent = (1 << log_m) * 32.0 * (world - 1) / world
    return {"log_n": log_n, "elements_per_gpu": 1 << log_m, "layout": "coefficients block-column, evaluations cyclic (include/zl_backend.h)",
            "forward_ms": f, "inverse_ms": i, "forward_elems_per_s": tot / (f * 1e-3), "inverse_elems_per_s": tot / (i * 1e-3),
            "exchange": "one RCCL all_to_all_single per transform", "bytes_sent_per_gpu": sent,
            "self_check": "iNTT(NTT(x)) == x on every rank; bit-exact parity of the legs in tests/test_gpu_sharded_ntt.py"}


R_BN = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
_TORCHRUN_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE",
                 "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_RUN_ID",
                 "TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_ERROR_FILE", "TORCH_NCCL_ASYNC_ERROR_HANDLING", "NCCL_ASYNC_ERROR_HANDLING")


def _free_port() -> int:
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _self_launch(n_gpus: int):
    """`python bench.py --gpus N` launched bare (no torchrun environment): replace this process by the one-rank-per-GPU launcher the
    contract names, with the same arguments."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without a torchrun environment: re-launching as %s" % (n_gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


class Run:
    """what every leg needs: the backend of this rank, its device, and the process group (if any)"""

    def __init__(self, args, torch, dist, be, dev, rank, world):
        self.args, self.torch, self.dist, self.be, self.dev, self.rank, self.world = args, torch, dist, be, dev, rank, world
        self.is_nccl = os.environ.get("ZL_DIST_BACKEND", "nccl") == "nccl"
        self.coll_dev = dev if self.is_nccl else None  # tensors of collectives: device memory over RCCL, host memory over gloo
        # ZL_FORCE_COLLECTIVE=1 at N = 1: a ONE-rank process group whose barriers / all_reduce / all_gather / all_to_all really run (first contact with RCCL on a one-GPU box)
        self.collective = world > 1 or (dist.is_available() and dist.is_initialized())

    def barrier(self):
        if self.collective:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, vals):
        if not self.collective:
            return [float(v) for v in vals]
        tt = self.torch.tensor([float(v) for v in vals], dtype=self.torch.float64, device=self.coll_dev)
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in tt.cpu().tolist()]


class MsmInputs:
    """One rank's shard of a synthetic MSM: bases P_i = k_i G with known k_i uniform in [1, r) (so uniform points of the group) from the device generator (resident in HBM), two
    different scalar vectors (consecutive steps of a pipelined batch alternate between them, so a cross-job buffer race in the
    three-stream pipeline cannot hide behind identical inputs), and the exact expected answers (sum s_i k_i mod r) G -- of this shard
    and of all shards together -- from one O(n) dot product per vector."""

    def __init__(self, R: Run, n: int, seed: int, curve=None, r_mod: int = R_BLS, bits: int = 255, local_only: bool = False):
        from openzl_amd import ZL_BLS12_381
        from openzl_amd.selfcheck import dot_mod_r, expected_point

        torch, be = R.torch, R.be
        self.R, self.n, self.curve = R, n, curve or ZL_BLS12_381
        k = random_scalars_lt_r(n, seed * 7919 + 1000 + R.rank, r_mod, bits)  # uniform in [0, r): P_i = k_i G is then a uniform point of the group
        k[~k.any(axis=1), 0] = 1
        t0 = time.perf_counter()
        self.h = be.bases_generate(self.curve, k)
        self.t_generate = time.perf_counter() - t0
        self.vecs = [random_scalars_lt_r(n, seed * 7919 + 2000 + R.rank, r_mod, bits), random_scalars_lt_r(n, seed * 7919 + 4000 + R.rank, r_mod, bits)]
        if R.args.scalars == "skewed":
            self.vecs = [skewed(v) for v in self.vecs]
        self.d_vecs = [torch.from_numpy(v.view(np.int64)).to(R.dev) for v in self.vecs]
        torch.cuda.synchronize()
        self.k64 = k  # (n, 4) limbs: the name is historical (rounds 1-4 drew 63-bit multipliers)
        self.dots = [dot_mod_r(v, k, r_mod) for v in self.vecs]
        self.exp_xy = [expected_point(be, self.curve, d) for d in self.dots]   # this rank's shard
        self.exp_all = self.exp_xy                                              # all ranks together
        if R.world > 1 and not local_only:
            mine = torch.tensor([(d >> (32 * j)) & 0xFFFFFFFF for d in self.dots for j in range(8)], dtype=torch.int64, device=R.coll_dev)
            allv = torch.empty(R.world * 16, dtype=torch.int64, device=mine.device)
            R.dist.all_gather_into_tensor(allv, mine)
            allv = allv.cpu().numpy().reshape(R.world, 2, 8)
            tot = [sum(sum(int(allv[g, j, w]) << (32 * w) for w in range(8)) for g in range(R.world)) % r_mod for j in (0, 1)]
            self.exp_all = [expected_point(be, self.curve, d) for d in tot]

    def free(self):
        if self.h is not None:
            self.R.be.bases_free(self.h)
            self.h = None
        self.d_vecs = []
        self.R.torch.cuda.empty_cache()


def msm_leg(R: Run, inp: MsmInputs, steps: int, warmup: int, pipelined: bool, solo: bool, what: str):
    """Gate (exact, full size, both scalar vectors, single call + pipelined batch + all ranks) -> `warmup` untimed steps -> EXACTLY `steps`
    timed steps between barriers, max over ranks -> every timed step checked exactly.  solo (N > 1): the same `steps` local MSMs on
    rank 0 alone while the other ranks wait: the N = 1 reference of the same per-GPU size for interference_ratio (= the weak-scaling efficiency of a fixed-shard leg)."""
    from openzl_amd.sharded import fold_partials, sharded_msm, sharded_msm_batch

    be, n, h, d_vecs, curve = R.be, inp.n, inp.h, inp.d_vecs, inp.curve

    def fail(msg):
        raise SystemExit(f"MSM self-check failed ({what}; {msg}): result != (sum s_i k_i) G at full size")

    def step(j=0):
        # local Pippenger -> 1 partial sum; N > 1: all_gather over RCCL + fold on every rank (openzl_amd/sharded.py)
        return sharded_msm(lambda: be.msm_partial_dev(h, d_vecs[j].data_ptr(), n), curve, device=R.coll_dev)

    def local_batch(cnt):
        # cnt MSMs as ONE pipelined batch (zl_msm_batch_partial_dev: sort of step i+2 | accumulation of step i+1 | tail of step i on three streams)
        return be.msm_batch_partial_dev(h, [d_vecs[i % 2].data_ptr() for i in range(cnt)], n)

    def steps_pipelined(cnt):
        # every step a complete MSM with its own result; N > 1: one all_gather of the K partials per rank, K folds
        return sharded_msm_batch(local_batch(cnt), curve, device=R.coll_dev)

    # ---- correctness gate before timing: the exact configuration that is timed, at full size, both scalar vectors ----------------------
    for j in (0, 1):
        xy_j, inf_j = fold_partials(curve, be.msm_partial_dev(h, d_vecs[j].data_ptr(), n).reshape(1, -1))
        if inf_j or not (np.asarray(xy_j) == inp.exp_xy[j]).all():
            fail("single call")
    if R.collective:  # the folded all-rank result must be (sum over ALL shards of s_i k_i) G
        for j in (0, 1):
            xy_j, inf_j = step(j)
            if inf_j or not (np.asarray(xy_j) == inp.exp_all[j]).all():
                fail("all ranks, single call")
    pipelined = pipelined and steps > 1
    if pipelined:
        # setup, untimed like the base upload: one 3-deep batch creates the side streams and grows all three buffer sets
        for i, part in enumerate(local_batch(3)):
            xy_i, inf_i = fold_partials(curve, part.reshape(1, -1))
            if inf_i or not (np.asarray(xy_i) == inp.exp_xy[i % 2]).all():
                fail("pipelined batch")
        if warmup:
            steps_pipelined(warmup)
    else:
        for _ in range(warmup):
            step()
    dom_ms, tot_ms = [], []
    R.barrier()
    t0 = time.perf_counter()
    if pipelined:
        results = steps_pipelined(steps)
        tm = be.last_timing()
        dom_ms.append(tm.dominant_ms)   # mean accumulation-kernel duration over the K steps (HIP events on its stream)
        tot_ms.append(tm.total_ms)      # device time per step, pipelined
    else:
        results = []
        for i in range(steps):
            results.append(step(i % 2))
            tm = be.last_timing()
            dom_ms.append(tm.dominant_ms)
            tot_ms.append(tm.total_ms)
    R.barrier()
    elapsed = time.perf_counter() - t0
    tm = be.last_timing()
    for i, (xy_i, inf_i) in enumerate(results):  # every timed step is checked exactly (N > 1: against the sum over all shards)
        if inf_i or not (np.asarray(xy_i) == inp.exp_all[i % 2]).all():
            fail("timed step")
    elapsed = R.max_over_ranks([elapsed])[0]
    # latency of one un-pipelined MSM call, for the record
    R.torch.cuda.synchronize()
    t1 = time.perf_counter()
    be.msm_partial_dev(h, d_vecs[0].data_ptr(), n)
    single_ms = (time.perf_counter() - t1) * 1e3
    single_dev = be.last_timing()
    out = {"elapsed": elapsed, "ms_per_step": elapsed / steps * 1e3, "dom_ms": float(np.mean(dom_ms)), "tot_ms": float(np.mean(tot_ms)),
           "window_bits": int(tm.window_bits), "entries": float(tm.entries), "pipelined": pipelined,
           "single_call_latency_ms": single_ms, "single_call_device_ms": float(single_dev.total_ms)}
    if solo and R.world > 1:
        # the same per-GPU work on ONE GPU with the others idle (rank 0, same process, same inputs, no collective): the N = 1 reference
        R.barrier()
        solo_ms = 0.0
        if R.rank == 0:
            t2 = time.perf_counter()
            if pipelined:
                local_batch(steps)
            else:
                for i in range(steps):
                    be.msm_partial_dev(h, d_vecs[i % 2].data_ptr(), n)
            R.torch.cuda.synchronize()
            solo_ms = (time.perf_counter() - t2) / steps * 1e3
        R.barrier()
        out["solo_ms_per_step"] = R.max_over_ranks([solo_ms])[0]
    return out


def scaling_entry(R: Run, name: str, total_desc: str, n_local: int, leg: dict, steps: int):
    tot = float(n_local) * R.world  # (ragged totals: every rank reports its own n; the legs below use equal shards)
    e = {"points_total": tot, "points_per_gpu": n_local, "what": total_desc, "ms_per_step": leg["ms_per_step"], "points_per_s": tot * steps / leg["elapsed"],
         "window_bits": leg["window_bits"], "kernel_ms": leg["dom_ms"], "checked_exactly": True,
         "result_check": "gate + every timed step equal (sum over ALL shards of s_i k_i) G exactly"}
    if "solo_ms_per_step" in leg:
        e["single_gpu_ms_per_step"] = leg["solo_ms_per_step"]
        e["interference_ratio"] = leg["solo_ms_per_step"] / leg["ms_per_step"]
        e["interference_note"] = ("time of the same per-GPU MSMs on rank 0 alone (same process, same inputs, other ranks idle, no collective) / time with all ranks + "
                                  "all_gather + fold.  For a leg whose per-GPU work is fixed as N grows (weak, config 4) this IS the scaling efficiency; for the strong leg it "
                                  "only measures interference -- see strong_scaling_efficiency")
        if name in ("weak", "config4"):
            e["weak_scaling_efficiency"] = e["interference_ratio"]
    if "t1_total_ms_per_step" in leg:
        e["single_gpu_total_ms_per_step"] = leg["t1_total_ms_per_step"]
        e["strong_scaling_efficiency"] = leg["t1_total_ms_per_step"] / (R.world * leg["ms_per_step"])
        e["strong_scaling_note"] = ("T_1(total) / (N x T_N): T_1 = the WHOLE problem (all N shards' points) as one MSM on rank 0 alone, measured in this process "
                                    "(pipelined like the timed steps, checked exactly); T_N = the timed all-rank step")
    return e


def strong_t1_leg(R: Run, log_total: int, steps: int, warmup: int, pipelined: bool):
    """T_1 of the strong-scaling definition: the whole 2^log_total-point problem on ONE GPU (rank 0, the other ranks wait at the barrier), same process,
    pipelined like the timed steps, every result checked exactly.  Returns ms per MSM (0.0 on the other ranks)."""
    from openzl_amd.sharded import fold_partials

    R.barrier()
    ms = 0.0
    if R.rank == 0:
        li = MsmInputs(R, 1 << log_total, 90 + log_total, local_only=True)
        try:
            be = R.be

            def run(cnt):
                if pipelined and cnt > 1:
                    return be.msm_batch_partial_dev(li.h, [li.d_vecs[i % 2].data_ptr() for i in range(cnt)], li.n)
                return np.stack([be.msm_partial_dev(li.h, li.d_vecs[i % 2].data_ptr(), li.n) for i in range(cnt)])

            run(max(warmup, 3))
            R.torch.cuda.synchronize()
            t0 = time.perf_counter()
            parts = run(steps)
            R.torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            for i in range(steps):
                xy, inf = fold_partials(li.curve, parts[i].reshape(1, -1))
                if inf or not (np.asarray(xy) == li.exp_xy[i % 2]).all():
                    raise SystemExit("MSM self-check failed (strong-scaling T_1 leg): result != (sum s_i k_i) G at full size")
        finally:
            li.free()
    R.barrier()
    return R.max_over_ranks([ms])[0]


VALU_CYCLES_PER_MIXED_ADD = 17028 + 186  # profiles/r05_acc_instruction_budget.txt (tools/isa_budget.py): VALU issue cycles of one loop iteration per wave + the bucket-boundary blocks x 0.63
NOMINAL_GHZ = 2.4


IN_KERNEL_CLOCK_GHZ = 2.09  # profiles/r06_clock_probe_2_24.log (2.088-2.094 over three launches of the c = 20 accumulation on a -DZL_MEASURE build of this round's code; r05: 2.075-2.10 at c = 19): tools/clock_probe.py


def int_alu_clock(acc_clock, head, dom_ms):
    """effective clock of the accumulation and what follows from it: SIMD cycles per wave-level mixed addition against the ISA's VALU issue cycles, and the
    kernel against the same instruction stream at the nominal 2.4 GHz.  The clock is the IN-KERNEL reading (s_memtime / s_memrealtime per wave of the
    clock-reading build of the same kernel), which since round 6 exists in -DZL_MEASURE builds only: this line quotes the committed
    measurement and, beside it, what a sleeping probe wave per XCD reads during one more batch of this run.  During the NTT passes that probe reads 2.0-2.05 GHz;
    during the accumulation it reads 2.40 where the kernel's own waves read 1.93-2.28 each (profiles/r05_clock_probe_2_24.log) and GRBM_GUI_ACTIVE / wall gives 2.05
    (profiles/r05_pmc_clock_msm_2_24.json) -- the disagreement is not explained, so the probe's figure is reported under its own name and nothing is derived from it."""
    f = IN_KERNEL_CLOCK_GHZ
    wave_adds_per_simd = head["entries"] / 64.0 / 1024.0
    cyc = f * 1e9 * dom_ms * 1e-3 / wave_adds_per_simd
    t_nominal_ms = wave_adds_per_simd * VALU_CYCLES_PER_MIXED_ADD / (NOMINAL_GHZ * 1e9) * 1e3
    probe = acc_clock if acc_clock and "error" not in acc_clock else None
    return {"effective_clock_ghz": f,
            "effective_clock_source": "in-kernel s_memtime / s_memrealtime deltas of every wave of the clock-reading build of this kernel (k_msm_accumulate_clk, -DZL_MEASURE builds only): "
                                      "committed measurement profiles/r06_clock_probe_2_24.log (2.088-2.094 GHz), same kernel and same kernel time; NOT re-read by this run",
            "sleeping_probe_clock_ghz": probe["effective_clock_ghz"] if probe else None,
            "sleeping_probe_note": ("one sleeping wave per XCD on its own high-priority stream spanning one more pipelined batch of four steps (accumulation %.2f ms per step during the probe, "
                                    "%.2f ms in the timed steps); disagrees with the in-kernel and the GRBM readings of the same kernel, nothing is derived from it" % (probe["accumulate_ms_during_probe"], dom_ms))
                                   if probe else (acc_clock or {}).get("error", "not measured"),
            "simd_cycles_per_wave_mixed_add": cyc, "isa_valu_issue_cycles_per_wave_mixed_add": VALU_CYCLES_PER_MIXED_ADD,
            "valu_issue_efficiency_at_that_clock": VALU_CYCLES_PER_MIXED_ADD / cyc,
            "frac_vs_nominal_2p4ghz": t_nominal_ms / dom_ms,
            "clock_note": "the chip clocks this body to its power budget (MI355X_MICROARCH.md, DVFS): frac_vs_nominal_2p4ghz = (the ISA's VALU issue cycles at 2.4 GHz) / kernel time = "
                          "valu_issue_efficiency x effective_clock / 2.4; the multiplier chain alone (peak) clocks higher than the accumulation (profiles/r05_clock_probe_2_24.log)"}


def scaling_model_leg(R: Run, inp: MsmInputs, head: dict):
    """N = 1 only: what can be known about multi-GPU scaling from one GPU.  The exchange of the sharded MSM is one all_gather of 512 B per rank and N - 1 group
    additions (microseconds), so the per-GPU time at N ranks is the single-GPU time of the shard: measured here, pipelined, at 2^21 / 2^22 / 2^23 (prefixes of
    the headline's inputs, every result checked exactly against the prefix's own dot product), next to the headline's 2^24.  NO hardware claim: the implied
    efficiencies hold if and only if the ranks do not disturb each other (the driver's SCALE run measures that)."""
    from openzl_amd.selfcheck import dot_mod_r, expected_point
    from openzl_amd.sharded import fold_partials

    be, torch = R.be, R.torch
    full_log = int(np.log2(inp.n))
    ms = {f"2^{full_log}": head["ms_per_step"]}
    for lg in (21, 22, 23):
        if lg >= full_log:
            continue
        m = 1 << lg
        exp = [expected_point(be, inp.curve, dot_mod_r(v[:m], inp.k64[:m], R_BLS)) for v in inp.vecs]
        ptrs = [inp.d_vecs[i % 2].data_ptr() for i in range(8)]
        be.msm_batch_partial_dev(inp.h, ptrs[:3], m)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        parts = be.msm_batch_partial_dev(inp.h, ptrs, m)
        torch.cuda.synchronize()
        ms[f"2^{lg}"] = (time.perf_counter() - t0) / 8 * 1e3
        for i in range(8):
            xy, inf = fold_partials(inp.curve, parts[i].reshape(1, -1))
            if inf or not (np.asarray(xy) == exp[i % 2]).all():
                raise SystemExit(f"MSM self-check failed (scaling model, 2^{lg} prefix)")
    t = lambda lg: ms.get(f"2^{lg}")  # noqa: E731
    model = {"measured_ms_per_msm_pipelined": ms, "checked_exactly": True,
             "exchange": "one all_gather of 512 B per rank + N - 1 group additions on every rank: microseconds, not modelled",
             "weak_2^%d_per_gpu" % full_log: {"N=2": 1.0, "N=4": 1.0, "N=8": 1.0, "note": "per-GPU work does not change with N; only the un-modelled exchange and rank interference remain"},
             "note": "PREDICTED from single-GPU measurements, not measured on N GPUs: efficiency(N) = T_1(total) / (N x T_1(total / N))"}
    if full_log == 24 and all(t(x) for x in (21, 22, 23)):
        model["strong_2^24_total"] = {"N=2": t(24) / (2 * t(23)), "N=4": t(24) / (4 * t(22)), "N=8": t(24) / (8 * t(21)),
                                      "note": "north_star's >= 0.9 per-GPU efficiency at 8 GPUs is a weak-scaling statement for this path: a 2^21-point shard amortises "
                                              "its bucket reduction and sort over 8x fewer points"}
        model["config4_2^26_total"] = {"N=8_per_gpu_rate_vs_2^24_rate": (2 ** 23 / t(23)) / (2 ** 24 / t(24)), "N=4_per_gpu_rate_vs_2^24_rate": 1.0,
                                       "note": "2^23 points per GPU at N = 8 (2^24 at N = 4 = the headline's shard): points/s per GPU relative to the headline's"}
    return model


def config1_leg(R: Run, no_cpu: bool):
    """BASELINE config 1: 2^16 BN254 G1 variable-base MSM -- GPU single call / pipelined, exact against the known discrete logs, and (cpu_baseline
    part) the CPU oracle single-threaded on the same bases and scalars (the reference's CPU path as configured), results equal bit for bit."""
    from openzl_amd import ZL_BN254

    be, torch = R.be, R.torch
    n = 1 << 16
    inp = MsmInputs(R, n, 16, curve=ZL_BN254, r_mod=R_BN, bits=254)
    try:
        d = inp.d_vecs[0]
        for _ in range(3):
            xy, inf = be.msm_dev(inp.h, d.data_ptr(), n)
        if inf or not (np.asarray(xy) == inp.exp_xy[0]).all():
            raise SystemExit("config 1 self-check failed: 2^16 BN254 MSM != (sum s_i k_i) G")
        ts, dv = [], []
        for _ in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            be.msm_dev(inp.h, d.data_ptr(), n)
            ts.append(time.perf_counter() - t0)
            dv.append(be.last_timing().total_ms)
        tm = be.last_timing()
        be.msm_batch_partial_dev(inp.h, [inp.d_vecs[i % 2].data_ptr() for i in range(6)], n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        be.msm_batch_partial_dev(inp.h, [inp.d_vecs[i % 2].data_ptr() for i in range(6)], n)
        batch_ms = (time.perf_counter() - t0) / 6 * 1e3
        out = {"config": "2^16 BN254 G1 variable-base MSM", "gpu_single_call_ms": float(np.min(ts)) * 1e3, "gpu_single_call_median_ms": float(np.median(ts)) * 1e3,
               "gpu_device_ms": float(np.median(dv)), "gpu_pipelined_ms_per_msm": batch_ms, "window_bits": int(tm.window_bits),
               "gpu_points_per_s": n / float(np.min(ts)), "checked_exactly": True}
        if not no_cpu:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as ol
            from oracle_lib import po

            bases = be.bases_download(inp.h)
            cxy, cinf, sec = ol.oracle_msm_g1_timed(po.BN254, bases, inp.vecs[0], algo=0, threads=1)
            if cinf or not (np.asarray(cxy) == np.asarray(xy)).all():
                raise SystemExit("config 1 parity check failed: the CPU oracle's 2^16 BN254 MSM differs from the GPU result")
            out["cpu_baseline"] = {"value": n / sec, "unit": "points/s", "cores": 1, "kind": "port", "ms": sec * 1e3,
                                   "sample": "the complete 2^16 input, single thread, ark window rule (c = 13): the reference's CPU configuration", "parity_full_size": True}
        return out
    finally:
        inp.free()


def config2_leg(R: Run):
    """BASELINE config 2: 2^20 BLS12-381 G1 MSM on one GPU: single call, pipelined batch, integer-ALU fraction of the WHOLE call."""
    be, torch = R.be, R.torch
    n = 1 << 20
    inp = MsmInputs(R, n, 20)
    try:
        leg = msm_leg(R, inp, 20, 2, True, False, "config 2")  # (a 20-step batch like the headline's default driver run: a 6-step batch of 3-ms jobs is one third ramp)
        d = inp.d_vecs[0]
        ts, dv, ac = [], [], []
        for _ in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            be.msm_partial_dev(inp.h, d.data_ptr(), n)
            ts.append(time.perf_counter() - t0)
            tm = be.last_timing()
            dv.append(tm.total_ms)
            ac.append(tm.dominant_ms)
        single = float(np.min(ts)) * 1e3
        mul_eq = float(tm.entries) * MULS_PER_MIXED_ADD
        peak = fq_mul_peak_live(be)
        return {"config": "2^20 BLS12-381 G1 MSM, uniform scalars < r, bases k_i G resident", "single_call_ms": single, "single_call_median_ms": float(np.median(ts)) * 1e3,
                "single_call_device_ms": float(np.median(dv)), "kernel_ms": float(np.median(ac)), "pipelined_ms_per_msm": leg["ms_per_step"],
                "window_bits": int(tm.window_bits), "points_per_s_single": n / (single * 1e-3), "points_per_s_pipelined": n / (leg["ms_per_step"] * 1e-3),
                "int_alu_frac_single_call": mul_eq / (single * 1e-3) / 1e9 / peak, "int_alu_frac_pipelined": mul_eq / (leg["ms_per_step"] * 1e-3) / 1e9 / peak,
                "int_alu_frac_kernel": mul_eq / (float(np.median(ac)) * 1e-3) / 1e9 / peak, "int_alu_peak_live": peak,
                "int_alu_frac_single_call_vs_constant_operand_peak": mul_eq / (single * 1e-3) / 1e9 / FQ_MUL_PEAK_CONST_G, "checked_exactly": True,
                "note": "int_alu_frac_* = (point, window) pairs x 9.04 multiplication-equivalents / time / the multiplier's standalone LIVE-DATA rate measured in this run "
                        "(int_alu_peak_live; rounds 2-3 divided by 78.6 G/s, a constant-operand figure: *_vs_constant_operand_peak keeps that series): whole call wall time, "
                        "pipelined time per MSM, accumulation kernel alone"}
    finally:
        inp.free()


def mctx_run(args, devices, log_n, ntt_log_m, steps, warmup, g16_k=None):
    """ONE process, G devices: zl_ctx_create_multi (RCCL communicator inside the library when the devices are distinct) + zl_msm_sharded
    (complete local Pippenger per device on its own host thread -> ncclAllGather of the partials -> fold) and zl_ntt_sharded (cross step ->
    grouped ncclSend / ncclRecv all-to-all -> local transform).  Returns the result dict; every MSM result is checked exactly."""
    import torch
    from openzl_amd import ZL_BLS12_381
    from openzl_amd.backend import MultiBackend
    from openzl_amd.selfcheck import dot_mod_r, expected_point

    mb = MultiBackend(devices)
    G = mb.size
    n = 1 << log_n
    hs, d_s, dots = [], [], [0, 0]
    try:
        for g in range(G):
            dev = torch.device("cuda", devices[g])
            k = random_scalars_lt_r(n, 77000 + g)
            hs.append(mb.ranks[g].bases_generate(ZL_BLS12_381, k))
            vs = [random_scalars_lt_r(n, 78000 + 2 * g), random_scalars_lt_r(n, 78001 + 2 * g)]
            d_s.append([torch.from_numpy(v.view(np.int64)).to(dev) for v in vs])
            for j in (0, 1):
                dots[j] = (dots[j] + dot_mod_r(vs[j], k, R_BLS)) % R_BLS
            del k, vs
        for g in set(devices):
            torch.cuda.synchronize(g)
        exp = [expected_point(mb.ranks[0], ZL_BLS12_381, d) for d in dots]

        def call(j):
            return mb.msm_sharded(hs, [d_s[g][j].data_ptr() for g in range(G)], [n] * G)

        for j in (0, 1):
            xy, inf = call(j)
            if inf or not (np.asarray(xy) == exp[j]).all():
                raise SystemExit("zl_msm_sharded self-check failed: result != (sum over all shards of s_i k_i) G")
        for i in range(warmup):
            call(i % 2)
        t0 = time.perf_counter()
        res = [call(i % 2) for i in range(steps)]
        el = time.perf_counter() - t0
        for i, (xy, inf) in enumerate(res):
            if inf or not (np.asarray(xy) == exp[i % 2]).all():
                raise SystemExit("zl_msm_sharded self-check failed (timed step)")
        # the same per-GPU MSM on rank 0's ctx alone (single calls, as zl_msm_sharded issues them)
        t1 = time.perf_counter()
        for i in range(steps):
            mb.ranks[0].msm_partial_dev(hs[0], d_s[0][i % 2].data_ptr(), n)
        solo = (time.perf_counter() - t1) / steps
        out = {"transport": "mctx: one process, zl_ctx_create_multi + zl_msm_sharded / zl_ntt_sharded (include/zl_backend.h)", "n_gpus": G, "devices": list(devices),
               "rccl_ranks": G if mb.uses_rccl else 0, "exchange": "RCCL (ncclCommInitAll; ncclAllGather / grouped ncclSend+ncclRecv)" if mb.uses_rccl else
               "virtual ranks sharing a device: device-to-device copies with the same data movement pattern (RCCL refuses duplicate devices)",
               "msm": {"points_per_gpu": n, "points_total": float(n) * G, "steps": steps, "ms_per_step": el / steps * 1e3, "points_per_s": float(n) * G * steps / el,
                       "single_gpu_ms_per_step": solo * 1e3, "interference_ratio": solo / (el / steps), "checked_exactly": True,
                       "issued_as": "separate zl_msm_sharded calls (not pipelined)"}}
        for hh, r in zip(hs, mb.ranks):
            r.bases_free(hh)
        hs, d_s = [], []
        torch.cuda.empty_cache()
        log_g = G.bit_length() - 1
        if ntt_log_m and (1 << log_g) == G and 1 <= log_g <= 4 and 2 * log_g <= ntt_log_m + log_g:
            M = 1 << ntt_log_m
            xs = [random_scalars_lt_r(M, 79000 + g) for g in range(G)]
            dx = [torch.from_numpy(x.view(np.int64)).to(torch.device("cuda", devices[g])) for g, x in enumerate(xs)]
            ptrs = [t.data_ptr() for t in dx]
            f_ms, i_ms = [], []
            for it in range(1 + 3):
                for inverse, acc in ((False, f_ms), (True, i_ms)):
                    for g in set(devices):
                        torch.cuda.synchronize(g)
                    t0 = time.perf_counter()
                    mb.ntt_sharded(ZL_BLS12_381, ptrs, ntt_log_m + log_g, inverse=inverse, mont=True)
                    if it:
                        acc.append((time.perf_counter() - t0) * 1e3)
            ok = all(bool((t.cpu().numpy().view(np.uint64) == x).all()) for t, x in zip(dx, xs))
            if not ok:
                raise SystemExit("zl_ntt_sharded self-check failed: iNTT(NTT(x)) != x")
            tot = float(M) * G
            out["ntt"] = {"log_n": ntt_log_m + log_g, "elements_per_gpu": M, "forward_ms": float(np.mean(f_ms)), "inverse_ms": float(np.mean(i_ms)),
                          "forward_elems_per_s": tot / (float(np.mean(f_ms)) * 1e-3), "inverse_elems_per_s": tot / (float(np.mean(i_ms)) * 1e-3),
                          "self_check": "iNTT(NTT(x)) == x on every rank; bit-exact parity of the entry point in tests/test_gpu_multi.py"}
        k_g16 = g16_k if g16_k is not None else (min(args.groth16_k, 4096 if log_n >= 22 else 64) if getattr(args, "groth16_k", 0) > 0 else 0)
        if k_g16 > 0:
            out["groth16_one_proof_over_ranks"] = mctx_groth16_leg(mb, k_g16)
        return out
    finally:
        for hh, r in zip(hs, mb.ranks):
            try:
                r.bases_free(hh)
            except Exception:  # noqa: BLE001
                pass
        mb.close()


def mctx_groth16_leg(mb, k):
    """ONE proof over the ranks of the mctx (zl_groth16_prove_sharded): the key is compiled on rank 0, every query cut into G contiguous slices and uploaded to the
    ranks, the witness map runs on rank 0, every rank its five partial MSMs, the host folds and assembles.  The proof must equal rank 0's single-device proof for
    the same (r, s) byte for byte.  Latency scaling of one proof (what 'constraints/s at N GPUs' means for a single prover), beside the replicas leg."""
    from openzl_amd import ZL_BLS12_381, Circuit, Groth16Keys

    be0 = mb.ranks[0]
    circ = Circuit(ZL_BLS12_381, k)
    keys = Groth16Keys(be0, circ, seed=0x5EED0006)
    skeys = None
    try:
        arr = circ.arrays()
        n_c = arr["n_constraints"]
        ref, r, s = keys.prove(seed=7)
        for _ in range(2):
            keys.prove(seed=7)
        t1 = []
        for _ in range(5):
            t0 = time.perf_counter()
            keys.prove(seed=7)
            t1.append(time.perf_counter() - t0)
        pk = {name: be0.bases_download(getattr(keys.pk, name)) for name in ("a_query", "b_g1_query", "h_query", "l_query", "b_g2_query")}
        nq = 6
        for name, words in (("alpha_g1", 2 * nq), ("beta_g1", 2 * nq), ("delta_g1", 2 * nq), ("beta_g2", 4 * nq), ("delta_g2", 4 * nq)):
            pk[name] = np.ctypeslib.as_array(getattr(keys.pk, name), shape=(words,)).copy()
        skeys = mb.groth16_shard_keys(ZL_BLS12_381, pk, arr)
        z = arr["assignment"]
        got = skeys.prove(z, r, s)
        if not all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(got, ref)):
            raise SystemExit("zl_groth16_prove_sharded self-check failed: the proof over the ranks differs from the single-device proof")
        for _ in range(2):
            skeys.prove(z, r, s)
        tg = []
        for _ in range(5):
            t0 = time.perf_counter()
            skeys.prove(z, r, s)
            tg.append(time.perf_counter() - t0)
        return {"hashes": k, "constraints": int(n_c), "ranks": mb.size, "prove_ms": float(np.median(tg)) * 1e3, "single_device_prove_ms": float(np.median(t1)) * 1e3,
                "constraints_per_s": n_c / float(np.median(tg)), "equals_single_device_proof": True,
                "note": "functional: the witness map runs on rank 0 alone and the five MSMs of a rank run one after the other through the single-call path; "
                        "exchange = device-to-device copies of the z / h slices + 5 x 512 B of partials per rank"}
    finally:
        if skeys is not None:
            skeys.close()
        keys.close()
        circ.close()


def main_mctx(args):
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    devices = [int(x) for x in args.mctx_devices.split(",")] if args.mctx_devices else list(range(args.gpus))
    res = mctx_run(args, devices, args.log_n, 0 if args.no_ntt else args.ntt_log_n, args.steps, args.warmup)
    m = res["msm"]
    line = {"metric": "MSM points/sec (BLS12-381 G1)", "value": m["points_per_s"], "unit": "points/s", "n_gpus": len(devices), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": m["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"bls12_381_g1_msm_2^{args.log_n}_per_gpu", "points_per_gpu": 1 << args.log_n, "parallelism": f"mctx{len(devices)}",
                       "transport": res["transport"]}, "mctx": res}
    print(json.dumps(line), flush=True)


def mctx_child(args, devices, timeout_s=240):
    """time the one-process transport in a child process (this process may still hold a process group / its GPU); fail-soft"""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in _TORCHRUN_ENV}
    cmd = [sys.executable, os.path.abspath(__file__), "--transport", "mctx", "--gpus", str(len(devices)), "--mctx-devices", ",".join(str(d) for d in devices),
           "--log-n", str(min(args.log_n, 22)), "--ntt-log-n", str(min(args.ntt_log_n, 22)), "--steps", str(max(2, min(args.steps, 6))), "--warmup", "1"]
    # (2^22 points / elements per device: the child generates every device's inputs itself, one after the other; the point of this leg is the
    #  library's own RCCL exchange, whose cost does not depend on the shard size)
    if args.no_ntt:
        cmd.append("--no-ntt")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env, cwd=ROOT)
        rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not rows:
            return {"error": f"child exited with {r.returncode}: {(r.stderr or r.stdout)[-400:]}"}
        return json.loads(rows[-1])["mctx"]
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=24, help="log2 points per GPU of the headline (BASELINE metric: 24)")
    ap.add_argument("--window", type=int, default=0, help="force the Pippenger window width (0 = auto)")
    ap.add_argument("--fixed-key", type=int, default=22,
                    help="window width of the precomputed-table mode reported BESIDE the headline as msm_fixed_key (zl_bases_precompute: "
                         "2^(c w) P_i, W x the base memory, built once per key); -1 = skip that leg")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--no-skew", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs 1 / 2 / 4 legs (N = 1) and the config-4 / strong-scaling legs (N > 1)")
    ap.add_argument("--no-pipeline", action="store_true", help="issue the K steps as K separate calls instead of one pipelined batch")
    ap.add_argument("--scalars", choices=["uniform", "skewed"], default="uniform", help="skewed: 50%% zeros, 25%% ones, rest uniform (profiling aid)")
    ap.add_argument("--ntt-log-n", type=int, default=24)
    ap.add_argument("--groth16-k", type=int, default=4096, help="config 5: chained Poseidon hashes (4096 -> domain 2^20); 0 = skip")
    ap.add_argument("--transport", choices=["ranks", "mctx"], default="ranks",
                    help="ranks: one process per GPU over torch.distributed (RCCL); mctx: ONE process, zl_ctx_create_multi + zl_msm_sharded / zl_ntt_sharded")
    ap.add_argument("--mctx-devices", default="", help="mctx: comma-separated device ids (may repeat: virtual ranks on one GPU); default 0..N-1")
    ap.add_argument("--config4-log-total", type=int, default=26, help="N > 1: total points of the config-4 leg (BASELINE: 2^26 over 8 GPUs)")
    ap.add_argument("--strong-log-total", type=int, default=24, help="N > 1: total points of the strong-scaling leg")
    ap.add_argument("--no-mctx", action="store_true", help="N > 1: skip the child-process run of the one-process transport")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-scalar leg (pcie_inclusive): rocprofv3 runs of the headline keep only the headline's own accumulation launches")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not measure roofline.traffic in this run (four child processes under rocprofv3 --pmc after the timed legs, ~70 s); the committed PMC passes of the same configuration are quoted instead")
    ap.add_argument("--dry-run", action="store_true", help="print the legs a `--gpus N` run executes, the per-rank HBM plan and the expected wall time, and exit (no GPU, no process group)")
    args = ap.parse_args()

    if args.dry_run:
        print(json.dumps(dry_run_plan(args), indent=1))
        return
    if args.transport == "mctx":
        return main_mctx(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_launch(args.gpus)

    import torch
    import torch.distributed as dist
    from openzl_amd import Backend, ZL_BLS12_381
    from openzl_amd.selfcheck import dot_mod_r, expected_point
    from openzl_amd.sharded import fold_partials

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch as `python bench.py --gpus N` or with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    is_nccl = os.environ.get("ZL_DIST_BACKEND", "nccl") == "nccl"
    if not is_nccl:
        local_rank = local_rank % torch.cuda.device_count()  # test mode: ranks may share a GPU
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: --gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from openzl_amd.sharded import forced_collective

    forced = world == 1 and forced_collective()
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if forced:
            # one rank, real backend: every collective of the N > 1 path (barrier, all_reduce, all_gather_into_tensor, all_to_all_single on device tensors) runs
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # nccl = RCCL over xGMI.  ZL_DIST_BACKEND=gloo exists only to exercise the N>1 code path on a single-GPU box.
        if is_nccl:
            dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend=os.environ["ZL_DIST_BACKEND"], rank=rank, world_size=world)

    be = Backend(local_rank)
    be.enable_timing(True)
    if args.window:
        be.set_msm_window(args.window)
    n = 1 << args.log_n
    R = Run(args, torch, dist, be, dev, rank, world)
    if rank == 0 and world == 1:
        oracle_native()  # before the first leg that loads the oracle

    leg_errors = {}

    def _guard(name, fn):
        """A secondary leg must never cost the headline line: an unexpected exception there (allocation failure, a missing tool on the
        box) is recorded in the JSON line instead.  Failed self-checks raise SystemExit and still abort the run; with N > 1 the legs
        contain collectives, so an exception is fatal there as before."""
        try:
            fn()
        except Exception as e:  # noqa: BLE001
            if world > 1:
                raise
            leg_errors[name] = f"{type(e).__name__}: {e}"
            print(f"[bench] leg {name} failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)

    # ---- the headline: synthetic inputs generated per rank, resident in HBM before the timed region; gate; K timed steps ------------------
    def _dbg_c2(where):  # ZL_BENCH_DBG_C2=1 (developer aid, tools/ab/r6_c2_bench.sh): the config-2 leg at several points of the run, to stderr
        if os.environ.get("ZL_BENCH_DBG_C2") == "1" and rank == 0 and world == 1:
            c = config2_leg(R)
            print(f"[dbg c2 @ {where}] single {c['single_call_ms']:.3f} pipelined {c['pipelined_ms_per_msm']:.3f} dev {c['single_call_device_ms']:.3f}", file=sys.stderr, flush=True)

    _dbg_c2("start")
    inp = MsmInputs(R, n, 0)
    h, k64, vecs, d_vecs, exp_xy = inp.h, inp.k64, inp.vecs, inp.d_vecs, inp.exp_xy
    t_generate = inp.t_generate
    head = msm_leg(R, inp, args.steps, args.warmup, not args.no_pipeline, True, "headline")
    elapsed, pipelined, single_ms = head["elapsed"], head["pipelined"], head["single_call_latency_ms"]
    tm = be.last_timing()
    # effective shader clock while the dominant kernel runs: a sleeping one-wave-per-XCD probe on its own stream (include/zl_backend_test.h) spans one more pipelined
    # batch of four steps, ~90 % of which is the accumulation (the rest: the sorts and tails between two accumulations); every step checked like the timed ones.
    # (Round 5 read the clock inside a clock-reading build of the kernel, k_msm_accumulate_clk; that kernel now exists in -DZL_MEASURE builds only -- VERDICT r5
    # item 7 -- and tools/clock_probe.py still uses it there: profiles/r05_clock_probe_2_24.log has both readings side by side.)
    acc_clock = None
    if rank == 0:
        try:
            from openzl_amd.backend import hook_clock_probe_launch, hook_clock_probe_read
            cnt = 4
            torch.cuda.synchronize()
            hook_clock_probe_launch(be, max(100, int(0.92 * cnt * head["ms_per_step"] * 1e3)))
            parts_c = be.msm_batch_partial_dev(h, [d_vecs[i % 2].data_ptr() for i in range(cnt)], n)
            tm_c = be.last_timing()
            torch.cuda.synchronize()
            acc_clock = hook_clock_probe_read(be)
            acc_clock["accumulate_ms_during_probe"] = float(tm_c.dominant_ms)
            for i, part_c in enumerate(parts_c):
                xy_c, inf_c = fold_partials(inp.curve, part_c.reshape(1, -1))
                if inf_c or not (np.asarray(xy_c) == exp_xy[i % 2]).all():
                    raise SystemExit("MSM self-check failed (batch under the clock probe)")
        except SystemExit:
            raise
        except Exception as e:  # noqa: BLE001
            acc_clock = {"error": f"{type(e).__name__}: {e}"}
    _dbg_c2("after headline")
    scaling_model = None
    if rank == 0 and world == 1 and not args.no_configs and args.log_n >= 22:
        def _leg_scaling_model():
            nonlocal scaling_model
            scaling_model = scaling_model_leg(R, inp, head)

        _guard("scaling_model", _leg_scaling_model)

    def check(xy, inf, j, what):
        if inf or not (np.asarray(xy) == exp_xy[j]).all():
            raise SystemExit(f"MSM self-check failed ({what}): result != (sum s_i k_i) G at full size")

    def local_batch(cnt):
        return be.msm_batch_partial_dev(h, [d_vecs[i % 2].data_ptr() for i in range(cnt)], n)

    s_host = vecs[0]
    d_scalars = d_vecs[0]
    # ---- PCIe-inclusive rate (SURVEY.md §8d config 2's timed region: scalars from host memory + kernels + result; never `value`) ---------
    pcie_info = None
    def _leg_pcie_info():
        nonlocal pcie_info
        be.msm(h, s_host)  # warm-up: staging buffer
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            xy_p, inf_p = be.msm(h, s_host)
            ts.append(time.perf_counter() - t1)
        check(xy_p, inf_p, 0, "host scalars")
        pcie_info = {"ms_per_msm": float(np.min(ts)) * 1e3, "points_per_s": n / float(np.min(ts)),
                     "note": "zl_msm: scalars copied from pageable host memory (32 B/point over PCIe) + all kernels + result D2H, bases resident"}

    if rank == 0 and world == 1 and not args.no_pcie:
        _guard("pcie_info", _leg_pcie_info)

    skew_info = None
    def _leg_skew_info():
        nonlocal skew_info
        # SURVEY.md §8d's non-uniform variant: Groth16-witness-like scalars (50 % zeros, 25 % ones, rest uniform), same bases
        s2 = skewed(s_host)
        d2 = torch.from_numpy(s2.view(np.int64)).to(dev)
        torch.cuda.synchronize()
        ts = []
        for it in range(1 + 3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            be.msm_partial_dev(h, d2.data_ptr(), n)
            if it:
                ts.append(time.perf_counter() - t1)
        xy_s, inf_s = fold_partials(ZL_BLS12_381, be.msm_partial_dev(h, d2.data_ptr(), n).reshape(1, -1))
        if inf_s or not (np.asarray(xy_s) == expected_point(be, ZL_BLS12_381, dot_mod_r(s2, k64, R_BLS))).all():
            raise SystemExit("MSM self-check failed (skewed scalars): result != (sum s_i k_i) G at full size")
        skew_info = {"scalars": "50% zeros, 25% ones, 25% uniform < r, shuffled", "ms_per_step": float(np.mean(ts)) * 1e3, "checked_exactly": True,
                     "points_per_s": n / float(np.mean(ts)),
                     "note": "zero digits are dropped by the recoder; scalars equal to 1 bypass the sort (compact list + direct sum, as "
                             "arkworks special-cases them); other repeated values form giant buckets cut into fixed-length chunks and merged "
                             "in two stages; exactness of these paths: tests/test_gpu_msm.py (skewed cases), tests/test_gpu_msm_fuzz.py"}
        del d2

    if rank == 0 and world == 1 and not args.no_skew:
        _guard("skew_info", _leg_skew_info)

    # ---- fixed-key mode, reported beside the headline: table of 2^(c w) P_i built once per key (a Groth16 proving key is static) -------
    fixed_info = None
    def _leg_fixed_info():
        nonlocal fixed_info
        c_fk = args.fixed_key if args.log_n >= 24 else 0  # let the library pick c for small inputs
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        be.bases_precompute(h, c_fk)
        torch.cuda.synchronize()
        build_ms = (time.perf_counter() - t1) * 1e3
        for j in (0, 1):
            xy_j, inf_j = fold_partials(ZL_BLS12_381, be.msm_partial_dev(h, d_vecs[j].data_ptr(), n).reshape(1, -1))
            check(xy_j, inf_j, j, "fixed-key single call")
        tmf1 = be.last_timing()
        local_batch(3)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        parts = local_batch(args.steps)
        torch.cuda.synchronize()
        fk_ms = (time.perf_counter() - t1) * 1e3 / args.steps
        tmf = be.last_timing()
        for i, part in enumerate(parts):
            xy_i, inf_i = fold_partials(ZL_BLS12_381, part.reshape(1, -1))
            check(xy_i, inf_i, i % 2, "fixed-key pipelined step")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        be.msm_partial_dev(h, d_vecs[0].data_ptr(), n)
        fk_single = (time.perf_counter() - t1) * 1e3
        plain_ms = elapsed / args.steps * 1e3
        cw = int(tmf.window_bits)
        W = (256 + cw - 1) // cw
        fixed_info = {
            "points_per_s": n / (fk_ms * 1e-3), "ms_per_msm": fk_ms, "single_call_latency_ms": fk_single, "window_bits": cw, "windows": W,
            "kernel_ms": float(tmf.dominant_ms),
            "table_build_ms": build_ms, "table_bytes": float(W) * n * 128.0,
            "break_even_msms": (build_ms / (plain_ms - fk_ms)) if plain_ms > fk_ms else None,
            "note": "zl_bases_precompute: 2^(c w) P_i for every window beside the bases, so all windows share ONE bucket set and c grows to 22 "
                    "(12 instead of 14 additions per point); the build is per key and pays back only after break_even_msms MSMs on the same "
                    "bases (a Groth16 proving key).  Same exact full-size checks as the headline.  NOT `value`: multi_scalar_mul(bases, scalars) has no per-key state.",
        }

    if args.fixed_key >= 0 and rank == 0 and world == 1 and not args.window:
        _guard("fixed_info", _leg_fixed_info)

    ntt_info = None
    def _leg_ntt_info():
        nonlocal ntt_info
        # second half of the metric.  N > 1: (i) independent replicas, one 2^log_n transform per GPU (Groth16's a/b/c pipelines
        # are independent transforms) -- every rank measures, rank 0 reports max-over-ranks; (ii) further down, ONE
        # 2^(log_n + log2 N) transform spread over all ranks with a single RCCL all-to-all (openzl_amd/sharded.py).
        ln = args.ntt_log_n
        x = random_scalars_lt_r(1 << ln, 3000 + rank)
        dx = torch.from_numpy(x.view(np.int64)).to(dev)
        torch.cuda.synchronize()
        # canonical -> (treated as Montgomery limbs: any residue < r is a valid Montgomery representative)
        fwd, inv = [], []
        for it in range(8 + 8):  # 8 untimed round trips (twiddle tables; the clocks of an idle device take ~6 transforms to settle), 8 timed
            be.ntt_dev(ZL_BLS12_381, dx.data_ptr(), ln, inverse=False, mont=True)
            f = be.last_timing().total_ms
            be.ntt_dev(ZL_BLS12_381, dx.data_ptr(), ln, inverse=True, mont=True)
            i = be.last_timing().total_ms
            if it >= 8:
                fwd.append(f)
                inv.append(i)
        back = dx.cpu().numpy().view(np.uint64)
        if not (back == x).all():
            raise SystemExit("NTT self-check failed: iNTT(NTT(x)) != x")
        # effective clock while the passes run: a sleeping one-wave-per-XCD probe on its own stream spans 8 more round trips (include/zl_backend_test.h)
        ntt_clock = None
        try:
            from openzl_amd.backend import hook_clock_probe_launch, hook_clock_probe_read
            span_us = int(0.9 * 8 * (float(np.mean(fwd)) + float(np.mean(inv))) * 1e3)
            torch.cuda.synchronize()
            hook_clock_probe_launch(be, max(100, span_us))
            for _ in range(8):
                be.ntt_dev(ZL_BLS12_381, dx.data_ptr(), ln, inverse=False, mont=True)
                be.ntt_dev(ZL_BLS12_381, dx.data_ptr(), ln, inverse=True, mont=True)
            torch.cuda.synchronize()
            ntt_clock = hook_clock_probe_read(be)
        except Exception as e:  # noqa: BLE001
            ntt_clock = {"error": f"{type(e).__name__}: {e}"}
        ntt_cpu = None
        if not args.no_cpu and rank == 0 and world == 1:
            # CPU oracle on the same vector; its forward transform must equal the GPU's bit for bit (full size)
            ntt_cpu, X_cpu = cpu_baseline_ntt(x, args.cpu_threads)
            be.ntt_dev(ZL_BLS12_381, dx.data_ptr(), ln, inverse=False, mont=True)
            X_gpu = dx.cpu().numpy().view(np.uint64)
            if not (X_gpu == X_cpu).all():
                raise SystemExit("full-size parity check failed: the CPU oracle's NTT differs from the GPU result")
            ntt_cpu["parity_full_size"] = True
            del X_cpu, X_gpu
        ntt_traffic, ntt_traffic_src = None, None
        for cand in ("r06_pmc_traffic_ntt.json", "r05_pmc_traffic_ntt.json", "r04_pmc_traffic_ntt.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", cand)))
                if ln == int(pmc["log_n"]):
                    ntt_traffic, ntt_traffic_src = pmc["traffic_bytes_per_transform"], cand
                    break
            except Exception:
                continue
        f_ms, i_ms = float(np.mean(fwd)), float(np.mean(inv))
        if world > 1:
            cpu_dev = dev if os.environ.get("ZL_DIST_BACKEND", "nccl") == "nccl" else None
            tt = torch.tensor([f_ms, i_ms], dtype=torch.float64, device=cpu_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            f_ms, i_ms = float(tt[0].item()), float(tt[1].item())
        tot = float(1 << ln) * world
        ntt_info = {
            "metric": "NTT elems/sec (BLS12-381 Fr, radix-2, natural order in/out)", "log_n": ln, "n_gpus": world,
            "scaling": "replicas" if world > 1 else "single",
            "forward_ms": f_ms, "inverse_ms": i_ms,
            "forward_elems_per_s": tot / (f_ms * 1e-3), "inverse_elems_per_s": tot / (i_ms * 1e-3),
            "fwd_plus_inv_elems_per_s": tot / ((f_ms + i_ms) * 1e-3),
            "roofline": {"bound": "hbm", "achieved": 64.0 * (1 << ln) / (f_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": 64.0 * (1 << ln) / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": ntt_traffic,
                         "effective_clock_ghz": (ntt_clock or {}).get("effective_clock_ghz"), "nominal_clock_ghz": NOMINAL_GHZ,
                         "effective_clock_source": "sleeping clock probe (one wave per XCD on its own stream: s_memtime / s_memrealtime) spanning 8 forward + inverse transforms issued back to back"
                                                   if ntt_clock and "error" not in ntt_clock else (ntt_clock or {}).get("error"),
                         "time_at_nominal_clock_ms": (f_ms * ntt_clock["effective_clock_ghz"] / NOMINAL_GHZ) if ntt_clock and ntt_clock.get("effective_clock_ghz") else None,
                         "traffic_unit": f"bytes per transform, all passes, 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction of the guide; profiles/{ntt_traffic_src or 'r04_pmc_traffic_ntt.json'}; null for other sizes)",
                         "note": "per GPU; 64 B/element algorithmic (32 read + 32 written) per transform, all butterfly passes of one transform together; "
                                 "the last pass also streams its combined twiddles (+32 B/element read, one multiplication less)"},
            "cpu_baseline": ntt_cpu,
        }
        del dx

    if not args.no_ntt:
        _guard("ntt_info", _leg_ntt_info)

    g16_info = None
    def _leg_g16_info():
        nonlocal g16_info
        # config 5: Groth16 prove of the Poseidon-hash chain circuit through the C++ host mirror (Groth16<E>::compile /
        # prove, csrc/zl_host.h): matrices + proving key device-resident, only the assignment travels per proof.
        from openzl_amd import Circuit, Groth16Keys

        t0 = time.perf_counter()
        circ = Circuit(ZL_BLS12_381, args.groth16_k)
        t_synth = time.perf_counter() - t0
        n_c, n_i, n_w = circ.shape
        t0 = time.perf_counter()
        keys = Groth16Keys(be, circ, seed=0x5EED0006)
        t_setup = time.perf_counter() - t0
        p0, r_g16, s_g16 = keys.prove(seed=7)  # warm-up (twiddle tables, scratch growth, event pool, persistent host threads) ...
        for _ in range(2):                     # ... and two more: the second and third proof of a key still take ~8 ms longer (tools/g16_lat_dist.py: 58, 8.5, 8.4, then 1.2 ms
            keys.prove(seed=7)                 # at k = 1), and the first launches of a process see the clock ramp (profiles/r04_ntt_drift.log)
        times, devms = [], []
        for _ in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            p1, _, _ = keys.prove(seed=7)
            times.append(time.perf_counter() - t0)
            devms.append(be.last_timing().total_ms)
        if not all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(p0, p1)):
            raise SystemExit("Groth16 self-check failed: proof not reproducible for fixed (r, s)")
        # the proof of the full-size circuit must verify (Groth16::verify with the host pairing), and must not verify for a wrong input
        pub = circ.arrays()["assignment"][1:2]
        t0 = time.perf_counter()
        ok = keys.verify(p1, pub)
        t_verify = time.perf_counter() - t0
        bad = pub.copy()
        bad[0, 0] ^= np.uint64(1)
        if not ok or keys.verify(p1, bad):
            raise SystemExit("Groth16 self-check failed: verify(proof) != (True, False on a wrong public input)")
        # per-proof synthesis once the context holds the matrices: the same circuit code in a witness-only compiler (R1CS::for_witness = ark-relations'
        # SynthesisMode::Prove { construct_matrices: false }); its proof must be the full compiler's, byte for byte
        t0 = time.perf_counter()
        circ_w = Circuit(ZL_BLS12_381, args.groth16_k, witness_only=True)
        t_wsynth = time.perf_counter() - t0
        pw, _, _ = keys.prove(seed=7, circuit=circ_w)
        if not all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(p0, pw)):
            raise SystemExit("Groth16 self-check failed: the proof from the witness-only compiler differs")
        circ_w.close()
        tp = float(np.median(times))  # steady-state latency of one proof: median of 7 after three warm-up proofs (min / mean / every sample beside it)
        g16_info = {
            "metric": "Groth16 prove constraints/sec (BLS12-381, Poseidon arity-2 hash chain, config 5)",
            "hashes": args.groth16_k, "constraints": n_c, "instance_vars": n_i, "witness_vars": n_w,
            "domain_log_n": int(max(1, (n_c + n_i - 1).bit_length())),
            "prove_ms": tp * 1e3, "prove_ms_min": float(np.min(times)) * 1e3, "prove_ms_mean": float(np.mean(times)) * 1e3,
            "prove_ms_samples": [round(t * 1e3, 3) for t in times], "prove_device_ms": float(np.median(devms)), "constraints_per_s": n_c / tp,
            "synthesis_s": t_synth, "setup_s": t_setup, "verify_ms": t_verify * 1e3, "verified": True,
            # BASELINE config 5 says "end-to-end": circuit synthesis (R1CS + full assignment of the Poseidon chain on the host, openzl::R1CS) + prove; the reference's
            # prove() boundary (groth16.rs:445-457) starts AFTER synthesis, which is what constraints_per_s prices
            "end_to_end_s": t_synth + tp, "end_to_end_constraints_per_s": n_c / (t_synth + tp),
            # ... and per proof AFTER the first (the matrices are static and device-resident: only a witness-only synthesis + prove remain)
            "witness_only_synthesis_s": t_wsynth, "end_to_end_per_further_proof_s": t_wsynth + tp, "end_to_end_per_further_proof_constraints_per_s": n_c / (t_wsynth + tp),
            "timing": "prove_ms = median of 7 proofs after three warm-up proofs (prove_ms_mean / _min / _samples beside it)",
            "note": "prove = Groth16<E>::prove: assignment H2D, spmv, 7 NTTs, 4 G1 MSMs + 1 G2 MSM on the device, host assembly; "
                    "the proof is verified here with Groth16::verify (host pairing); bit-exact parity vs the oracle in tests/test_groth16.py, tests/test_host_mirror.py",
        }
        if not args.no_cpu:
            g16_info["cpu_baseline"] = cpu_baseline_groth16(be, keys, circ, p1, r_g16, s_g16, args.cpu_threads)
        lane_jobs = [(g16_info, keys, circ, p1, 10)]  # measured at the END of this leg (the lanes' streams must not change which streams of the proofs timed here share a hardware queue)
        # SURVEY.md §8d config 5 also names the small circuits: k = 1 (N = 2^8) and k = 2^6 (N = 2^14); latency-bound, reported beside
        small = []
        for k_small in (1, 64):
            if k_small >= args.groth16_k:
                continue
            c2 = Circuit(ZL_BLS12_381, k_small)
            k2 = Groth16Keys(be, c2, seed=0x5EED0006)
            for _ in range(4):
                k2.prove(seed=7)
            ts2 = []
            for _ in range(15):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                pr, _, _ = k2.prove(seed=7)
                ts2.append(time.perf_counter() - t0)
            ok2 = k2.verify(pr, c2.arrays()["assignment"][1:2])
            if not ok2:
                raise SystemExit("Groth16 self-check failed on the small circuit")
            small.append({"hashes": k_small, "constraints": c2.shape[0], "prove_ms": float(np.median(ts2)) * 1e3, "prove_ms_min": float(np.min(ts2)) * 1e3,
                          "prove_ms_mean": float(np.mean(ts2)) * 1e3, "prove_ms_samples": [round(t * 1e3, 3) for t in ts2], "constraints_per_s": c2.shape[0] / float(np.median(ts2)), "verified": True,
                          "timing": "median of 15 proofs after four warm-up proofs"})
            lane_jobs.append((small[-1], k2, c2, pr, 40))
        g16_info["small_circuits"] = small
        # the reference's other pairing curve (plugins/arkworks: `bn254` feature): the same circuit over BN254, both groups on the 28-bit lazily reduced
        # fields since round 4; reported beside config 5, not part of it
        from openzl_amd import ZL_BN254
        cb = Circuit(ZL_BN254, args.groth16_k)
        kb = Groth16Keys(be, cb, seed=0x5EED0006)
        for _ in range(3):
            kb.prove(seed=7)
        tb = []
        for _ in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pb, _, _ = kb.prove(seed=7)
            tb.append(time.perf_counter() - t0)
        if not kb.verify(pb, cb.arrays()["assignment"][1:2]):
            raise SystemExit("Groth16 self-check failed on the BN254 circuit")
        g16_info["bn254"] = {"hashes": args.groth16_k, "constraints": cb.shape[0], "prove_ms": float(np.median(tb)) * 1e3, "prove_ms_min": float(np.min(tb)) * 1e3,
                             "constraints_per_s": cb.shape[0] / float(np.median(tb)), "verified": True, "timing": "median of 7 proofs after three warm-up proofs"}
        kb.close()
        cb.close()
        # Throughput of TWO prover lanes (zl_groth16_prove_circuits -> zl_ctx_fork: a second host thread proving over the same device-resident key, as two threads may share the reference's
        # &ProvingContext): proofs per second of a stream of proofs, not the latency of one.  Every proof of both lanes must equal the single-lane proof.
        try:
            for info, kx, cx, ref, cnt in lane_jobs:
                kx.prove_many([7] * 6)  # warm-up of the second lane (its scratch, streams, twiddles)
                runs = []
                for _ in range(5):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    got = kx.prove_many([7] * (2 * cnt))
                    runs.append((time.perf_counter() - t0) / (2 * cnt))
                    if not all(all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(ref, pr_l)) for pr_l in got):
                        raise SystemExit("Groth16 self-check failed: a proof of the two-lane stream differs from the single-lane proof")
                med = float(np.median(runs))
                # (the gain compares like with like -- the MEDIAN per-proof time of five two-lane streams against the MEDIAN single-lane proof)
                info["two_lanes"] = {"ms_per_proof": med * 1e3, "ms_per_proof_min": min(runs) * 1e3, "ms_per_proof_samples": [round(x * 1e3, 3) for x in runs],
                                     "constraints_per_s": cx.shape[0] / med, "gain_vs_one_lane": info["prove_ms"] * 1e-3 / med,
                                     "timing": f"zl_groth16_prove_circuits: a stream of {2 * cnt} proofs over ONE device-resident key on two prover lanes (two host threads inside the library, "
                                               f"zl_ctx_fork), wall / {2 * cnt}, median of 5 streams (gain = median single-lane proof / this); every proof checked byte for byte against the single-lane proof"}
        finally:
            be.L.zl_ctx_drop_lanes(be._ctx)
            for _, kx, cx, _, _ in lane_jobs:
                kx.close()
                cx.close()

    if args.groth16_k > 0 and rank == 0 and world == 1:
        _guard("g16_info", _leg_g16_info)

    cpu = None
    def _leg_cpu():
        nonlocal cpu
        cpu = cpu_baseline(be, h, k64, s_host, exp_xy[0], args.cpu_threads, args.log_n)

    if not args.no_cpu and rank == 0 and world == 1:
        _guard("cpu", _leg_cpu)

    # ---- BASELINE configs 1, 2, 4 on the N = 1 line (3 = ntt, 5 = groth16 above) -------------------------------------------------------------
    configs = {}
    if rank == 0 and world == 1 and not args.no_configs:
        def _leg_c1():
            configs["1"] = config1_leg(R, args.no_cpu)

        def _leg_c2():
            configs["2"] = config2_leg(R)

        def _leg_c4():
            # 2^26 = 8 x 2^23 through the sharded entry point with 8 virtual ranks on this one GPU: functional (exact) check of the whole
            # config-4 data path; its multi-GPU throughput needs N > 1 (scaling.config4 on that line)
            shard_log = 23 if args.log_n >= 24 else max(10, args.log_n - 3)
            r4 = mctx_run(args, [local_rank] * 8, shard_log, 0, 2, 1, g16_k=64 if args.groth16_k > 0 else 0)
            m4 = r4["msm"]
            configs["4"] = {"config": f"2^{shard_log + 3} BLS12-381 G1 MSM as 8 shards of 2^{shard_log} (zl_msm_sharded, 8 virtual ranks on ONE GPU)",
                            "functional": True, "checked_exactly": True, "ms_per_msm_on_one_gpu": m4["ms_per_step"], "points_per_s_on_one_gpu": m4["points_per_s"],
                            "multi_gpu_throughput": "not measured at N = 1 (see scaling.config4 of a --gpus N run)", "exchange": r4["exchange"],
                            "groth16_one_proof_over_8_virtual_ranks": r4.get("groth16_one_proof_over_ranks")}

        _dbg_c2("before config 1")
        _guard("config1", _leg_c1)
        _dbg_c2("after config 1")
        _guard("config2", _leg_c2)
        _guard("config4", _leg_c4)
        configs["3"] = "see ntt (2^24 BLS12-381 Fr forward + inverse)"
        configs["5"] = "see groth16 (Poseidon hash-chain circuit, 958 465 constraints)"

    if rank == 0:
        pts = float(n) * world * args.steps
        value = pts / elapsed
        dom = head["dom_ms"]
        achieved = 128.0 * n / (dom * 1e-3) / 1e9
        # HBM traffic of the dominant kernel from the committed PMC passes (profiles/r02_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE in separate runs of this same command); only valid for the profiled configuration, null otherwise
        traffic = None
        traffic_src = None
        traffic_how = None
        under_prof = any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ)  # (a rocprofv3 run of this file: no profiler inside the profiler)
        if world == 1 and not args.no_live_traffic and not under_prof:
            # measured NOW, after every timed leg: child processes under rocprofv3 --pmc (live_pmc_traffic); any failure falls back to the committed passes below
            try:
                t_live, det = live_pmc_traffic("msm", args.log_n)
                if det.get("window_bits") not in (None, head["window_bits"]):
                    raise RuntimeError(f"the child process picked c = {det['window_bits']}, the timed run c = {head['window_bits']}")
                traffic, traffic_how = t_live, det
            except Exception as e:  # noqa: BLE001
                leg_errors["live_traffic_msm"] = f"{type(e).__name__}: {e}"
            if ntt_info is not None:
                try:
                    t_live, det = live_pmc_traffic("ntt", int(ntt_info["log_n"]))
                    ntt_info["roofline"]["traffic"] = t_live
                    ntt_info["roofline"]["traffic_unit"] = "bytes per transform, all passes, 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction of the guide); " + det["how"]
                    ntt_info["roofline"]["traffic_detail"] = det
                except Exception as e:  # noqa: BLE001
                    leg_errors["live_traffic_ntt"] = f"{type(e).__name__}: {e}"
        for cand in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json"):  # (r03 and earlier hold the un-corrected FETCH + WRITE sum)
            if traffic is not None:
                break
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", cand)))
                if args.log_n == int(pmc["log_n"]) and head["window_bits"] == int(pmc["window_bits"]) and not pmc.get("precomputed_table", False):
                    traffic = pmc["k_msm_accumulate_traffic_bytes"]
                    traffic_src = cand
                    break
            except Exception:
                continue
        cw = head["window_bits"]
        line = {
            "metric": "MSM points/sec (BLS12-381 G1)",
            "value": value,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"bls12_381_g1_msm_2^{args.log_n}_per_gpu", "points_per_gpu": n, "curve": "BLS12-381 G1",
                       "scalars": "uniform < r (255 bit)" if args.scalars == "uniform" else "50% zeros, 25% ones, 25% uniform",
                       "bases": "k_i*G with k_i uniform in [1, r) (uniform points of the group, discrete logs known to the checker only) from a device generator, resident in HBM; NO per-key precomputation (what multi_scalar_mul(bases, scalars) is)",
                       "window_bits": cw, "windows": (256 + cw - 1) // cw,
                       "precomputed_table": "none (the fixed-key table mode is reported separately as msm_fixed_key)",
                       "parallelism": f"shard{world}" if world > 1 else "single",
                       "steps_issued_as": "one pipelined batch (zl_msm_batch_partial_dev): sort | accumulate | tail of consecutive steps overlap on three streams; "
                                          "consecutive steps alternate between two scalar vectors" if pipelined else "separate calls",
                       "single_call_latency_ms": single_ms, "single_call_device_ms": head["single_call_device_ms"],
                       "bases_generate_s": t_generate,
                       "result_check": "every timed step equals (sum s_i k_i) G exactly at full size (known discrete logs); the CPU oracle's MSM of the "
                                       "complete input equals it too (cpu_baseline.parity_full_size)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_unit": ("bytes per launch, 2 x FETCH_SIZE + WRITE_SIZE (the guide's gfx950 correction: FETCH_SIZE tallies 128-B requests at 64 B); " + traffic_how["how"]) if traffic_how else
                                         f"bytes per launch, 2 x FETCH_SIZE + WRITE_SIZE (the guide's gfx950 correction: FETCH_SIZE tallies 128-B requests at 64 B), from the committed PMC passes of this configuration "
                                         f"(profiles/{traffic_src or 'r04_pmc_traffic.json'}; --no-live-traffic, N > 1 or a failed live pass: see leg_errors); null for any other configuration",
                         "traffic_detail": traffic_how,
                         "algorithmic_bytes": 128.0 * n, "kernel": "k_msm_accumulate",
                         "int_alu": {"unit": "G Fq-mul/s", "achieved": head["entries"] * MULS_PER_MIXED_ADD / (dom * 1e-3) / 1e9, "peak": fq_mul_peak_live(be),
                                     "frac": head["entries"] * MULS_PER_MIXED_ADD / (dom * 1e-3) / 1e9 / fq_mul_peak_live(be),
                                     "peak_constant_operands": FQ_MUL_PEAK_CONST_G, "frac_vs_constant_operand_peak": head["entries"] * MULS_PER_MIXED_ADD / (dom * 1e-3) / 1e9 / FQ_MUL_PEAK_CONST_G,
                                     "mads_per_mixed_add": MADS_PER_MIXED_ADD, "muls_per_mixed_add": MULS_PER_MIXED_ADD,
                                     **int_alu_clock(acc_clock, head, dom),
                                     "note": "the roofline that actually binds: (point, window) pairs x 9.04 multiplication-equivalents per mixed add "
                                             "(6M + 2S + one dual product scan with a shared Montgomery reduction = 6 x 392 + 2 x 301 + 588 = 3542 v_mad_u64_u32, counted in the kernel's ISA) "
                                             "/ kernel time, against the standalone rate of the same 14x28-bit Montgomery multiplier at the kernel's occupancy (3 waves/SIMD) on "
                                             "per-lane pseudo-random operands, measured in this run on this box in steady state (zl_test_fq_mul_rate).  peak_constant_operands = "
                                             "rounds 2-3's constant, taken on hipMemset operands (those clock 11 % higher than random field elements: profiles/r04_fbench_f64.log)"},
                         "kernel_ms": dom, "device_total_ms": head["tot_ms"],
                         "note": "algorithmic bytes = 128 B/point (96 B base + 32 B scalar) x points per launch; the kernel is "
                                 "integer-multiply bound (DESIGN.md), so the HBM fraction is small by construction"},
            "cpu_baseline": cpu,
            "configs": configs or None,
            "scaling_model": scaling_model,
            "pcie_inclusive": pcie_info,
            "msm_fixed_key": fixed_info,
            "msm_skewed_scalars": skew_info,
            "ntt": ntt_info,
            "groth16": g16_info,
        }
        if leg_errors:
            line["leg_errors"] = leg_errors
    else:
        line = None
    inp.free()
    if world > 1:
        # Secondary legs at N > 1, last and under a watchdog: if anything stalls, the MSM line above is still printed.
        import threading

        def _give_up():
            if rank == 0:
                line["secondary_legs_error"] = "timed out"
                print(json.dumps(line), flush=True)
            os._exit(0)

        dog = threading.Timer(600.0, _give_up)
        dog.daemon = True
        dog.start()
        if rank == 0:
            line["scaling_legs"] = {"weak": scaling_entry(R, "weak", f"2^{args.log_n} points per GPU (the headline)", n, head, args.steps)}
        if not args.no_configs:
            # BASELINE config 4 (2^26 points over the ranks; 2^23 per GPU at N = 8) and strong scaling (2^24 points over the ranks):
            # own inputs per leg, same gate / exact checks / barriers as the headline, plus the single-GPU reference of the same shard size
            for name, log_total in (("config4", args.config4_log_total), ("strong", args.strong_log_total)):
                n_loc = max(1 << 10, (1 << log_total) // world)
                try:
                    li = MsmInputs(R, n_loc, 40 + log_total)
                    leg = msm_leg(R, li, args.steps, args.warmup, not args.no_pipeline, True, name)
                    li.free()
                    if name == "strong":
                        leg["t1_total_ms_per_step"] = strong_t1_leg(R, log_total, args.steps, args.warmup, not args.no_pipeline)
                    ent, err = scaling_entry(R, name, f"2^{log_total} points in total = {n_loc} per GPU", n_loc, leg, args.steps), 0.0
                except Exception as e:  # noqa: BLE001 -- (failed self-checks are SystemExit and abort)
                    ent, err = {"error": f"{type(e).__name__}: {e}"}, 1.0
                    print(f"[rank {rank}] scaling leg {name} failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
                if R.max_over_ranks([err])[0] != 0.0 and "error" not in ent:
                    ent = {"error": "a rank failed (see stderr)"}
                if rank == 0:
                    line["scaling_legs"][name] = ent
        if args.groth16_k > 0:
            # constraints/s at N GPUs: independent proofs, one per GPU (replicas: a proof does not shard), max-over-ranks time
            try:
                n_c, t_prove = groth16_replica_leg(be, torch, args.groth16_k)
                err = 0.0
            except Exception as e:  # noqa: BLE001
                n_c, t_prove, err = 0, 0.0, 1.0
                print(f"[rank {rank}] groth16 leg failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            tt = R.max_over_ranks([t_prove, err, float(n_c)])
            if rank == 0:
                if tt[1] == 0.0 and tt[0] > 0.0:
                    line["groth16"] = {"metric": "Groth16 prove constraints/sec (BLS12-381, Poseidon arity-2 hash chain, config 5)", "n_gpus": world,
                                       "scaling": "replicas (one independent proof per GPU)", "hashes": args.groth16_k, "constraints": int(tt[2]),
                                       "prove_ms": tt[0] * 1e3, "constraints_per_s": world * tt[2] / tt[0], "verified": True}
                else:
                    line["groth16"] = {"error": "a rank failed (see stderr)"}
        if is_nccl and not args.no_ntt and (world & (world - 1)) == 0 and world <= 16:
            try:
                dinfo = distributed_ntt_leg(be, dist, torch, dev, rank, world, args.ntt_log_n)
            except Exception as e:  # noqa: BLE001 -- reported in the JSON line, the headline number stands
                dinfo = {"error": f"{type(e).__name__}: {e}"}
            if rank == 0 and line.get("ntt") is not None:
                line["ntt"]["distributed"] = dinfo
        # all collectives are done: leave the process group; ranks != 0 exit and release their GPUs
        R.barrier()
        be.close()
        dist.destroy_process_group()
        if rank != 0:
            dog.cancel()
            return
        if not args.no_mctx:
            # the C-ABI transport (one process, zl_ctx_create_multi: ncclCommInitAll inside the library) timed once, in a child process
            devs = list(range(world)) if is_nccl else [g % torch.cuda.device_count() for g in range(world)]
            time.sleep(1.0)
            line["mctx"] = mctx_child(args, devs)
        dog.cancel()
        print(json.dumps(line), flush=True)
        return
    if forced:
        # ZL_FORCE_COLLECTIVE=1: the gates and every timed step above went through all_gather_into_tensor of a one-rank group (msm_leg -> openzl_amd/sharded.py) and
        # were checked exactly; barriers and the MAX all_reduce ran on the same group.  The NTT exchange: one transform through sharded_ntt with G = 1.
        coll = {"backend": dist.get_backend(), "ranks": dist.get_world_size(), "tensors": "device (RCCL)" if is_nccl else "host (gloo)",
                "all_gather_into_tensor": "headline gate + every timed step, exact", "barrier_and_all_reduce_max": "timed region"}
        try:
            if is_nccl:
                coll["all_to_all_single"] = distributed_ntt_leg(be, dist, torch, dev, 0, 1, min(args.ntt_log_n, 20))
        except Exception as e:  # noqa: BLE001
            coll["all_to_all_single"] = {"error": f"{type(e).__name__}: {e}"}
        line["collective"] = coll
        dist.barrier()
        dist.destroy_process_group()
    print(json.dumps(line), flush=True)
    be.close()


# ---- --dry-run: what a `--gpus N` run will do, without a GPU ---------------------------------------------------------------------------------------
def _msm_scratch_bytes(n: int, c: int, sets: int = 3) -> dict:
    """HBM of one rank's MSM pipeline for n points at window c, following MsmJob::plan_as / alloc / sort_tmp_sizes (openzl_amd/csrc/zl_msm_job.h): `sets`
    rotating buffer sets (pipelined batches) + the shared sort temporaries.  An ESTIMATE for planning (grow-only slots add 1/8 of slack)."""
    W = (255 + 1 + c - 1) // c
    H = 1 << (c - 1)
    NB = W * H
    E = n * W
    chunk = 128 if (E >> 7) >= (1 << 20) else 64
    while chunk > 8 and E // chunk < (1 << 18):
        chunk >>= 1
    nchunks = (E + chunk - 1) // chunk
    xyzz = 256  # four 64-byte coordinates
    per_set = 4 * (3 * NB + n + E // 512) + 4 * E + NB * xyzz + 2 * nchunks * xyzz + (5 * W * max(1, H // 8) + 64) * xyzz
    gn = max(1, NB >> 15)
    s5 = 2 * (n * W * 2) + n * W + n * W * 4 + 8 * gn * W * 64
    s6 = n * W * 2 + n * W * 4 + 8 * gn * 128 * 16
    return {"window_bits": c, "windows": W, "buckets": NB, "entries": E, "per_buffer_set": per_set, "buffer_sets": sets, "sort_temporaries": s5 + s6,
            "total": int(1.125 * (sets * per_set + s5 + s6))}


def dry_run_plan(args) -> dict:
    """`python bench.py --gpus N --dry-run`: the legs of that run in order, every rank's HBM plan against 288 GB, and the expected wall time from the last committed
    single-GPU line under profiles/ (nothing is launched; no GPU and no process group needed)."""
    N = args.gpus
    n = 1 << args.log_n
    ref, ref_name = None, None
    for cand in ("r06_bench_final.json", "r05_bench_final.json", "r04_bench_final.json"):
        try:
            ref = json.loads(open(os.path.join(ROOT, "profiles", cand)).read().strip().split("\n")[-1])
            ref_name = cand
            break
        except Exception:
            continue
    ms24 = ref["ms_per_step"] if ref else 36.9
    sm = (ref or {}).get("scaling_model", {}) or {}
    meas = sm.get("measured_ms_per_msm_pipelined", {})
    ntt_ms = (ref["ntt"]["forward_ms"] + ref["ntt"]["inverse_ms"]) if ref and ref.get("ntt") else 4.5
    g16_ms = ref["groth16"]["prove_ms"] if ref and ref.get("groth16") else 18.4

    def msm_ms(log_pts):  # pipelined ms per MSM on one GPU: measured where the reference line holds it, else scaled from the nearest size
        k = f"2^{log_pts}"
        if k in meas:
            return float(meas[k])
        return ms24 * (2.0 ** (log_pts - 24)) * (1.0 + 0.06 * max(0, 24 - log_pts))

    pick_c = lambda lg: 19 if lg >= 24 else (18 if lg >= 23 else (17 if lg >= 21 else 16))  # noqa: E731 -- zl_pick_window's choices at these sizes
    K, Wm = args.steps, args.warmup
    legs, hbm = [], {}
    head_ms = msm_ms(args.log_n)
    legs.append({"leg": "headline (weak)", "per_gpu_points": n, "steps": K, "warmup": Wm, "collectives": "barrier x2, all_gather_into_tensor of K x 512 B per rank, all_reduce MAX" if N > 1 else "none",
                 "setup_s": round(0.11 * n / (1 << 24) * 2 + 0.6, 2), "timed_ms": round(K * head_ms, 1), "gate_and_warmup_ms": round((2 + 3 + Wm + (2 if N > 1 else 0)) * head_ms * 1.06, 1),
                 "solo_reference_on_rank0_ms": round(K * head_ms, 1) if N > 1 else 0})
    hbm["bases (Affine, 128 B/point)"] = 128 * n
    hbm["two scalar vectors (32 B/point each)"] = 64 * n
    hbm["msm pipeline scratch"] = _msm_scratch_bytes(n, args.window or pick_c(args.log_n))
    if N > 1 and not args.no_configs:
        for name, log_total in (("config4", args.config4_log_total), ("strong", args.strong_log_total)):
            lg = max(10, log_total - (N.bit_length() - 1))
            t = msm_ms(lg)
            legs.append({"leg": name, "total_points": 1 << log_total, "per_gpu_points": 1 << lg, "steps": K, "collectives": "as the headline", "timed_ms": round(K * t, 1),
                         "gate_and_warmup_ms": round((7 + Wm) * t * 1.06, 1), "solo_reference_on_rank0_ms": round(K * t, 1),
                         **({"t1_whole_input_on_rank0_ms": round((K + 4) * msm_ms(log_total), 1), "rank0_extra_hbm": 192 * (1 << log_total)} if name == "strong" else {})})
    if N == 1 and not args.no_configs:
        legs.append({"leg": "scaling_model + configs 1 / 2 / 4 (N = 1 only)", "timed_ms": round(8 * (msm_ms(21) + msm_ms(22) + msm_ms(23)) + 400, 1)})
    if N == 1:
        legs.append({"leg": "pcie_inclusive, msm_fixed_key (c = %d table: %d x the base memory), msm_skewed_scalars (N = 1 only)" % (args.fixed_key, (256 + args.fixed_key - 1) // max(1, args.fixed_key)),
                     "timed_ms": round(4 * 44 + 4 * 33 + 3000 + 6 * 40, 1)})
        hbm["fixed-key table (freed after its leg)"] = 128 * n * ((256 + args.fixed_key - 1) // max(1, args.fixed_key)) if args.fixed_key > 0 else 0
    if not args.no_ntt:
        ln = args.ntt_log_n
        legs.append({"leg": "ntt replicas", "log_n": ln, "transforms": 2 * 16 + 16, "timed_ms": round(24 * ntt_ms * 2.0 ** (ln - 24), 1), "collectives": "all_reduce MAX" if N > 1 else "none"})
        hbm["ntt vector + scratch (32 + 40 B/element)"] = 72 << ln
        hbm["ntt last-pass twiddle tables (32 B/element, forward + inverse) + row tables"] = (64 << ln) + (64 << 20)
        if N > 1 and (N & (N - 1)) == 0:
            legs.append({"leg": "ntt distributed", "log_n": ln + N.bit_length() - 1, "exchange": "one all_to_all_single per transform: %d MiB sent per GPU" % ((32 << ln) * (N - 1) // N >> 20),
                         "transforms": 8, "timed_ms": round(8 * (ntt_ms / 2 * 2.0 ** (ln - 24) + (32 << ln) / 50e9 * 1e3 + 0.3), 1)})
            hbm["distributed ntt: exchange buffer"] = 32 << ln
    if args.groth16_k > 0:
        cons = 234 * args.groth16_k + 1  # Poseidon arity-2 chain: 234 constraints per hash (tests/groth16_util.py)
        dom = 1 << max(1, (cons).bit_length())
        legs.append({"leg": "groth16 " + ("replicas (one independent proof per GPU)" if N > 1 else "config 5 + small circuits + BN254"), "constraints": cons, "domain": dom,
                     "host_synthesis_and_setup_s": round(2.0 * cons / 958465 + 0.3, 2), "proofs": 10, "timed_ms": round(10 * g16_ms * cons / 958465 + 5, 1)})
        hbm["groth16: proving key (5 queries + window tables) + R1CS CSR + witness-map vectors"] = int(cons * (128 * 3 * 14 + 256 * 17 + 3 * 3 * 12 + 7 * 40))
    if N > 1 and not args.no_mctx:
        legs.append({"leg": "mctx (child process of rank 0 after the ranks released their GPUs: zl_ctx_create_multi, ncclCommInitAll inside the library)", "timed_ms": round(6 * msm_ms(args.log_n) + 8 * ntt_ms + 3000, 1)})
    total_hbm = sum(v["total"] if isinstance(v, dict) else v for v in hbm.values())
    wall = sum(l.get("timed_ms", 0) + l.get("gate_and_warmup_ms", 0) + l.get("solo_reference_on_rank0_ms", 0) + l.get("t1_whole_input_on_rank0_ms", 0) for l in legs) / 1e3
    wall += sum(l.get("setup_s", 0) + l.get("host_synthesis_and_setup_s", 0) for l in legs)
    startup = 8.0 + (6.0 if N > 1 else 0.0)  # import torch on a fresh box (1-2 minutes the very first time), library load, RCCL communicator
    return {"dry_run": True, "n_gpus": N, "launch": "python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 --master-port P bench.py --gpus %d ..." % (N, N) if N > 1 else "python bench.py",
            "backend": os.environ.get("ZL_DIST_BACKEND", "nccl") + (" (RCCL over xGMI)" if os.environ.get("ZL_DIST_BACKEND", "nccl") == "nccl" else ""),
            "legs_in_order": legs, "hbm_plan_per_rank_bytes": hbm, "hbm_total_per_rank_gb": round(total_hbm / 1e9, 2), "hbm_capacity_gb": 288,
            "hbm_note": "peak is lower: the fixed-key table, the per-leg inputs and the groth16 keys are freed before the next leg; rank 0 of the strong leg also holds the whole 2^%d input" % args.strong_log_total,
            "expected_wall_s": round(wall + startup, 1), "expected_wall_note": "legs from the single-GPU numbers of profiles/%s (ms per pipelined MSM at each shard size, NTT, proof) + %.0f s of process start; "
                                                                                "secondary legs at N > 1 run under a 600-s watchdog" % (ref_name or "(none found: built-in constants)", startup),
            "nothing_was_run": True}


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X MSM / NTT backend (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log-n 24] [--no-cpu] [--no-ntt]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one variable-base MSM over 2^log_n random BLS12-381 G1 points and random scalars < r per GPU, bases and
scalars already resident in HBM, through the C ABI (zl_msm_partial_dev); with N > 1 every rank owns its own shard of
bases/scalars (weak scaling), the per-rank partial sums (512 B) are all-gathered over RCCL and folded
on every rank (zl_partials_sum).  Rank 0 prints ONE JSON line.  Extra objects on that line:
  roofline      HBM roofline of the dominant kernel (k_msm_accumulate): algorithmic bytes (128 B / point, SURVEY.md
                §8d) / its HIP-event duration on the backend's stream, vs 8 TB/s; .int_alu = the integer-multiply roofline
                that actually binds; .traffic = PMC bytes of the same configuration (profiles/)
  cpu_baseline  the CPU oracle (arkworks-algorithm restatement, NOT the arkworks binary) timed on this box's cores (N = 1 only)
  msm_skewed_scalars  the same MSM on Groth16-witness-like scalars (N = 1 only)
  ntt           2^24 BLS12-381 Fr forward+inverse NTT throughput (second half of the BASELINE metric); N > 1: per-GPU replicas
                and .distributed = ONE 2^(24 + log2 N) transform over all ranks with a single RCCL all-to-all
  groth16       config 5 (Poseidon-chain circuit, 958 465 constraints) prove time / constraints per second, proof verified;
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
FQ_MUL_PEAK_G = 74.3  # measured: the multiplier of zl_field28.h alone, 2 waves/SIMD, MI355X (71.6 - 74.3 over two boxes of the pool; the higher one) (tools/fbench28_asm.hip, profiles/r01_fbench_field_mul_asm.log)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


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


def limbs_to_int(row) -> int:
    return sum(int(v) << (64 * j) for j, v in enumerate(row))


def cpu_baseline(log_n_sample: int, threads_req: int):
    """Time the CPU oracle (oracle/libzl_oracle.so, 'port' of the arkworks 0.3.0 algorithm) on a bounded sample of the
    same workload.  Checker code used as a *reported baseline* only -- never on the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from oracle_lib import po

    curve = po.BLS12_381
    n = 1 << log_n_sample
    k = random_scalars_lt_r(n, 901)
    bases = ol.oracle_g1_mul_gen(curve, k)
    s = random_scalars_lt_r(n, 902)
    avail = os.cpu_count() or 1
    c = po.ark_window_bits(n)
    windows = (255 + c - 1) // c
    threads = max(1, min(threads_req or avail, windows))
    t0 = time.perf_counter()
    r1 = ol.oracle_msm_g1(curve, bases, s, algo=0, threads=1)
    t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    rN = ol.oracle_msm_g1(curve, bases, s, algo=0, threads=threads)
    tN = time.perf_counter() - t0
    assert (r1[0] == rN[0]).all()
    return {
        "value": n / tN,
        "unit": "points/s",
        "cores": threads,
        "kind": "port",
        "sample": f"2^{log_n_sample} BLS12-381 G1 points, ark window rule c={c} ({windows} windows), window-parallel over {threads} threads "
                  f"(what arkworks' `parallel` feature does); arkworks-algorithm restatement, not the arkworks binary",
        "single_thread_value": n / t1,
        "single_thread_note": "1 thread = the reference's actual configuration (no `parallel` feature, plugins/arkworks/Cargo.toml)",
        "host_cpus": avail,
    }, (curve, bases, s, r1)


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

    log_g = world.bit_length() - 1
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-n", type=int, default=24, help="log2 points per GPU (BASELINE metric: 24)")
    ap.add_argument("--window", type=int, default=0, help="force the Pippenger window width (0 = auto)")
    ap.add_argument("--precompute", type=int, default=22,
                    help="window width of the precomputed table of 2^(c w) P_i (zl_bases_precompute; W x the base memory, built once "
                         "at upload, untimed like the upload itself); -1 = no table (plain 16-bit windows)")
    ap.add_argument("--cpu-log-n", type=int, default=18)
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--no-skew", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="issue the K steps as K separate calls instead of one pipelined batch")
    ap.add_argument("--scalars", choices=["uniform", "skewed"], default="uniform", help="skewed: 50%% zeros, 25%% ones, rest uniform (profiling aid)")
    ap.add_argument("--ntt-log-n", type=int, default=24)
    ap.add_argument("--groth16-k", type=int, default=4096, help="config 5: chained Poseidon hashes (4096 -> domain 2^20); 0 = skip")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from openzl_amd import Backend, ZL_BLS12_381

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    if os.environ.get("ZL_DIST_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % torch.cuda.device_count()  # test mode: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # nccl = RCCL over xGMI.  ZL_DIST_BACKEND=gloo exists only to exercise the N>1 code path on a single-GPU box.
        backend = os.environ.get("ZL_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    be = Backend(local_rank)
    be.enable_timing(True)
    if args.window:
        be.set_msm_window(args.window)
    n = 1 << args.log_n

    # ---- synthetic inputs, generated per rank, resident in HBM before the timed region ---------------------------
    k = random_scalars_lt_r(n, 1000 + rank)          # discrete logs of the bases: P_i = k_i * G (device generator)
    h = be.bases_generate(ZL_BLS12_381, k)
    pre_c = args.precompute if args.precompute >= 0 and not args.window else -1
    if pre_c >= 0 and args.log_n < 24:
        pre_c = 0  # let the library pick c for small inputs
    if pre_c >= 0:
        be.bases_precompute(h, pre_c)
    s_host = random_scalars_lt_r(n, 2000 + rank)
    if args.scalars == "skewed":
        s_host = skewed(s_host)
    d_scalars = torch.from_numpy(s_host.view(np.int64)).to(dev)
    torch.cuda.synchronize()

    # ---- correctness gate before timing: known-discrete-log check on a 2^14 prefix (full sizes: tests/) -----------
    m = min(n, 1 << 14)
    got, inf = be.msm_dev(h, d_scalars.data_ptr(), m)
    dot = sum(limbs_to_int(a) * limbs_to_int(b) for a, b in zip(k[:m], s_host[:m])) % R_BLS
    kd = np.array([[(dot >> (64 * j)) & (2**64 - 1) for j in range(4)]], dtype=np.uint64)
    hd = be.bases_generate(ZL_BLS12_381, kd)
    exp = be.bases_download(hd)[0]
    be.bases_free(hd)
    if inf or not (got == exp).all():
        raise SystemExit("MSM self-check failed: result != (sum s_i k_i) G")

    from openzl_amd.sharded import sharded_msm, sharded_msm_batch

    gather_dev = dev if os.environ.get("ZL_DIST_BACKEND", "nccl") == "nccl" else None

    def step():
        # local Pippenger -> 1 partial sum; N > 1: all_gather over RCCL + fold on every rank (openzl_amd/sharded.py)
        return sharded_msm(lambda: be.msm_partial_dev(h, d_scalars.data_ptr(), n), ZL_BLS12_381, device=gather_dev)

    def steps_pipelined(k):
        # the K steps as ONE pipelined batch (zl_msm_batch_partial_dev: sort of step i+2 | accumulation of step i+1 | tail of step i on
        # three streams), every step a complete MSM with its own result; N > 1: one all_gather of the K partials per rank, K folds
        parts = be.msm_batch_partial_dev(h, [d_scalars.data_ptr()] * k, n)
        return sharded_msm_batch(parts, ZL_BLS12_381, device=gather_dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ref_xy, ref_inf = step()  # untimed: the full-size result through the plain single-call path, every timed step must reproduce it
    pipelined = not args.no_pipeline and args.steps > 1
    if pipelined:
        # setup, untimed like the base upload: one 3-deep batch creates the side streams and grows all three buffer sets
        for xy_k, inf_k in steps_pipelined(3):
            if inf_k != ref_inf or not (np.asarray(xy_k) == np.asarray(ref_xy)).all():
                raise SystemExit("MSM self-check failed: the pipelined path disagrees with the single-call result")
        if args.warmup:
            steps_pipelined(args.warmup)
    else:
        for _ in range(args.warmup):
            step()
    dom_ms, tot_ms = [], []
    barrier()
    t0 = time.perf_counter()
    if pipelined:
        results = steps_pipelined(args.steps)
        tm = be.last_timing()
        dom_ms.append(tm.dominant_ms)   # mean accumulation-kernel duration over the K steps (HIP events on its stream)
        tot_ms.append(tm.total_ms)      # device time per step, pipelined
    else:
        results = []
        for _ in range(args.steps):
            results.append(step())
            tm = be.last_timing()
            dom_ms.append(tm.dominant_ms)
            tot_ms.append(tm.total_ms)
    barrier()
    elapsed = time.perf_counter() - t0
    tm = be.last_timing()
    for xy_k, inf_k in results:
        if inf_k != ref_inf or not (np.asarray(xy_k) == np.asarray(ref_xy)).all():
            raise SystemExit("MSM self-check failed: a timed step disagrees with the single-call result")
    # latency of one un-pipelined MSM call, for the record
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    be.msm_partial_dev(h, d_scalars.data_ptr(), n)
    single_ms = (time.perf_counter() - t1) * 1e3
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if os.environ.get("ZL_DIST_BACKEND", "nccl") == "nccl" else None)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    skew_info = None
    if rank == 0 and world == 1 and not args.no_skew:
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
        skew_info = {"scalars": "50% zeros, 25% ones, 25% uniform < r, shuffled", "ms_per_step": float(np.mean(ts)) * 1e3,
                     "points_per_s": n / float(np.mean(ts)),
                     "note": "zero digits are dropped by the recoder; scalars equal to 1 bypass the sort (compact list + direct sum, as "
                             "arkworks special-cases them); other repeated values form giant buckets cut into fixed 64-entry chunks and merged "
                             "in two stages; exactness of these paths: tests/test_gpu_msm.py (skewed cases), tests/test_gpu_msm_fuzz.py"}
        del d2

    ntt_info = None
    if not args.no_ntt:
        # second half of the metric.  N > 1: (i) independent replicas, one 2^log_n transform per GPU (Groth16's a/b/c pipelines
        # are independent transforms) -- every rank measures, rank 0 reports max-over-ranks; (ii) further down, ONE
        # 2^(log_n + log2 N) transform spread over all ranks with a single RCCL all-to-all (openzl_amd/sharded.py).
        ln = args.ntt_log_n
        x = random_scalars_lt_r(1 << ln, 3000 + rank)
        dx = torch.from_numpy(x.view(np.int64)).to(dev)
        torch.cuda.synchronize()
        # canonical -> (treated as Montgomery limbs: any residue < r is a valid Montgomery representative)
        fwd, inv = [], []
        for it in range(1 + 3):
            be.ntt_dev(ZL_BLS12_381, dx.data_ptr(), ln, inverse=False, mont=True)
            f = be.last_timing().total_ms
            be.ntt_dev(ZL_BLS12_381, dx.data_ptr(), ln, inverse=True, mont=True)
            i = be.last_timing().total_ms
            if it:
                fwd.append(f)
                inv.append(i)
        back = dx.cpu().numpy().view(np.uint64)
        if not (back == x).all():
            raise SystemExit("NTT self-check failed: iNTT(NTT(x)) != x")
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
                         "frac": 64.0 * (1 << ln) / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "note": "per GPU; 64 B/element algorithmic (32 read + 32 written) per transform; kernel = k_ntt_pass x3 launches"},
        }
        del dx

    g16_info = None
    if args.groth16_k > 0 and rank == 0 and world == 1:
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
        p0, _, _ = keys.prove(seed=7)  # warm-up (twiddle tables, scratch growth)
        times, devms = [], []
        for _ in range(3):
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
        tp = float(np.mean(times))
        g16_info = {
            "metric": "Groth16 prove constraints/sec (BLS12-381, Poseidon arity-2 hash chain, config 5)",
            "hashes": args.groth16_k, "constraints": n_c, "instance_vars": n_i, "witness_vars": n_w,
            "domain_log_n": int(max(1, (n_c + n_i - 1).bit_length())),
            "prove_ms": tp * 1e3, "prove_device_ms": float(np.mean(devms)), "constraints_per_s": n_c / tp,
            "synthesis_s": t_synth, "setup_s": t_setup, "verify_ms": t_verify * 1e3, "verified": True,
            "note": "prove = Groth16<E>::prove: assignment H2D, spmv, 7 NTTs, 4 G1 MSMs + 1 G2 MSM on the device, host assembly; "
                    "the proof is verified here with Groth16::verify (host pairing); bit-exact parity vs the oracle in tests/test_groth16.py, tests/test_host_mirror.py",
        }
        keys.close()
        circ.close()
        # SURVEY.md §8d config 5 also names the small circuits: k = 1 (N = 2^8) and k = 2^6 (N = 2^14); latency-bound, reported beside
        small = []
        for k_small in (1, 64):
            if k_small >= args.groth16_k:
                continue
            c2 = Circuit(ZL_BLS12_381, k_small)
            k2 = Groth16Keys(be, c2, seed=0x5EED0006)
            k2.prove(seed=7)
            ts2 = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                pr, _, _ = k2.prove(seed=7)
                ts2.append(time.perf_counter() - t0)
            ok2 = k2.verify(pr, c2.arrays()["assignment"][1:2])
            if not ok2:
                raise SystemExit("Groth16 self-check failed on the small circuit")
            small.append({"hashes": k_small, "constraints": c2.shape[0], "prove_ms": float(np.mean(ts2)) * 1e3,
                          "constraints_per_s": c2.shape[0] / float(np.mean(ts2)), "verified": True})
            k2.close()
            c2.close()
        g16_info["small_circuits"] = small

    cpu = None
    if not args.no_cpu and rank == 0 and world == 1:
        cpu, _ = cpu_baseline(args.cpu_log_n, args.cpu_threads)

    if rank == 0:
        pts = float(n) * world * args.steps
        value = pts / elapsed
        dom = float(np.mean(dom_ms))
        achieved = 128.0 * n / (dom * 1e-3) / 1e9
        # HBM traffic of the dominant kernel from the committed PMC passes (profiles/r01_pmc_traffic_final.json: rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE in separate runs of this same command); only valid for the profiled configuration
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_final.json")))
            if args.log_n == 24 and int(tm.window_bits) == int(pmc["window_bits"]) and pre_c >= 0:
                traffic = pmc["k_msm_accumulate_traffic_bytes"]
        except Exception:
            traffic = None
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
                       "scalars": "uniform < r (255 bit)" if args.scalars == "uniform" else "50% zeros, 25% ones, 25% uniform", "bases": "k_i*G from a device generator, resident in HBM",
                       "window_bits": int(tm.window_bits),
                       "precomputed_table": (f"2^(c w) P_i for all windows, c={int(tm.window_bits)} (one merged bucket set; built at upload)"
                                             if pre_c >= 0 else "none"),
                       "parallelism": f"shard{world}" if world > 1 else "single",
                       "steps_issued_as": "one pipelined batch (zl_msm_batch_partial_dev): sort | accumulate | tail of consecutive steps overlap on three streams"
                                          if pipelined else "separate calls",
                       "single_call_latency_ms": single_ms,
                       "result_check": "known-discrete-log prefix check passed; bit-exact parity in tests/"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_unit": "bytes per launch (FETCH_SIZE + WRITE_SIZE, profiles/r01_pmc_traffic_final.json)",
                         "algorithmic_bytes": 128.0 * n, "kernel": "k_msm_accumulate",
                         "int_alu": {"unit": "G Fq-mul/s", "achieved": float(tm.entries) * 9.5 / (dom * 1e-3) / 1e9, "peak": FQ_MUL_PEAK_G,
                                     "frac": float(tm.entries) * 9.5 / (dom * 1e-3) / 1e9 / FQ_MUL_PEAK_G,
                                     "note": "the roofline that actually binds: (point, window) pairs x 9.5 multiplication-equivalents per mixed add "
                                             "(8M + 2S = 10 product scans, two of them sharing one Montgomery reduction: 3724 mads = 9.5 x 392) / kernel time, against the standalone rate of the same 14x28-bit Montgomery multiplier "
                                             "at the kernel's occupancy (tools/fbench28_asm.hip; profiles/r01_fbench_field_mul_asm.log)"},
                         "kernel_ms": dom, "device_total_ms": float(np.mean(tot_ms)),
                         "note": "algorithmic bytes = 128 B/point (96 B base + 32 B scalar) x points per launch; the kernel is "
                                 "integer-multiply bound (DESIGN.md), so the HBM fraction is small by construction"},
            "cpu_baseline": cpu,
            "msm_skewed_scalars": skew_info,
            "ntt": ntt_info,
            "groth16": g16_info,
        }
    else:
        line = None
    is_nccl = os.environ.get("ZL_DIST_BACKEND", "nccl") == "nccl"
    if world > 1:
        # Secondary legs at N > 1, last and under a watchdog: if anything stalls, the MSM line above is still printed.
        import threading

        def _give_up():
            if rank == 0:
                line["secondary_legs_error"] = "timed out after 300 s"
                print(json.dumps(line), flush=True)
            os._exit(0)

        dog = threading.Timer(300.0, _give_up)
        dog.daemon = True
        dog.start()
        if args.groth16_k > 0:
            # constraints/s at N GPUs: independent proofs, one per GPU (replicas: a proof does not shard), max-over-ranks time
            try:
                n_c, t_prove = groth16_replica_leg(be, torch, args.groth16_k)
                err = 0.0
            except Exception as e:  # noqa: BLE001
                n_c, t_prove, err = 0, 0.0, 1.0
                print(f"[rank {rank}] groth16 leg failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            tt = torch.tensor([t_prove, err, float(n_c)], dtype=torch.float64, device=dev if is_nccl else None)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            if rank == 0:
                if tt[1].item() == 0.0 and tt[0].item() > 0.0:
                    line["groth16"] = {"metric": "Groth16 prove constraints/sec (BLS12-381, Poseidon arity-2 hash chain, config 5)", "n_gpus": world,
                                       "scaling": "replicas (one independent proof per GPU)", "hashes": args.groth16_k, "constraints": int(tt[2].item()),
                                       "prove_ms": tt[0].item() * 1e3, "constraints_per_s": world * tt[2].item() / tt[0].item(), "verified": True}
                else:
                    line["groth16"] = {"error": "a rank failed (see stderr)"}
        if is_nccl and not args.no_ntt and (world & (world - 1)) == 0 and world <= 16:
            try:
                dinfo = distributed_ntt_leg(be, dist, torch, dev, rank, world, args.ntt_log_n)
            except Exception as e:  # noqa: BLE001 -- reported in the JSON line, the headline number stands
                dinfo = {"error": f"{type(e).__name__}: {e}"}
            if rank == 0:
                line["ntt"]["distributed"] = dinfo
        dog.cancel()
    if rank == 0:
        print(json.dumps(line), flush=True)
    be.bases_free(h)
    be.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

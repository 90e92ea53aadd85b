#!/usr/bin/env python3
"""bench.py -- LCQPs/sec of the batched dense penalty-homotopy path on MI355X (BASELINE.json metric).

A "step" is one complete solve of one synthetic batch per GPU: B = 1024 dense LCQPs with n = 256,
nC = 512, nComp = 64 (BASELINE.json configs[2]), generated directly in HBM before the timed region
(include/lcqp_synth.h).  Every step re-runs everything runSolver does: constant-matrix setup (C, the two
factorisations, Et) and the homotopy megakernel from x0 = 0.

N > 1 (launched by torch.distributed.run, one rank per GPU): independent instances are sharded over
the ranks (rank r solves instance ids [r*B, (r+1)*B)); there is no data-path collective, RCCL is used
only for the barrier and the max-over-ranks of the step time ("scaling": "weak").

One JSON line is printed by rank 0 (contract in the task description) with two extra objects:
  roofline      HBM roofline of the dominant kernel (k_lcqp_run), algorithmic bytes from the work
                counters the kernel keeps, duration from HIP events on the launch stream.
  cpu_baseline  the CPU oracle (a port of the same algorithm; the reference's qpOASES path cannot be
                built here) on a bounded sample of the same workload, all host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def shard_range(rank, world, per_rank):
    """instance ids solved by `rank`: contiguous slices of per_rank instances"""
    first = rank * per_rank
    return first, first + per_rank


def pmc_traffic(B, n, nC, nComp):
    """HBM bytes per k_lcqp_run launch from the committed rocprofv3 PMC passes of this same command
    (profiles/round1/README.md: (2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes); counters cannot be
    read from inside an un-profiled run, so the value is null for any other workload or when the file is absent."""
    if (B, n, nC, nComp) != (1024, 256, 512, 64):
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "latest_traffic.json")) as f:
            return float(json.load(f)["traffic_bytes_guide_recipe"])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU")
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--nC", type=int, default=512)
    ap.add_argument("--nComp", type=int, default=64)
    ap.add_argument("--cpu-sample", type=int, default=64, help="instances of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-backsolve", action="store_true", help="skip the standalone back-solve kernel measurement")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the two-batches-in-flight measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch
    import lcqpow_amd as la

    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        # launched by torch.distributed.run: one rank per GPU over RCCL (backend "nccl" is RCCL on ROCm)
        import torch.distributed as dist_
        dist = dist_
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if la.device_count() < 1:
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")

    B, n, nC, nComp = args.batch, args.n, args.nC, args.nComp
    opt = la.default_options(perturbStep=0, printLevel=0)     # defaults except these two
    bt = la.BatchLCQP(B, n, nC, nComp, device=local_rank, opt=opt)
    first, _ = shard_range(rank, world, B)
    bt.generate_synthetic(first)
    bt.synchronize()

    def barrier():
        bt.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        bt.run()
    barrier()
    t0 = time.perf_counter()
    setup_ms = solve_ms = 0.0
    for _ in range(args.steps):
        bt.run()
        s_ms, k_ms = bt.last_timing()      # HIP events on the launch stream (also waits for the step)
        setup_ms += s_ms
        solve_ms += k_ms
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    x, y, st = bt.solution()
    n_ok = sum(1 for s in st if s["returnValue"] == 0)
    alg_bytes = bt.algorithmic_bytes()                      # per launch of k_lcqp_run (last step)
    kernel_s = (solve_ms / args.steps) * 1e-3
    achieved = alg_bytes / kernel_s / 1e9 if kernel_s > 0 else 0.0
    mean = lambda k: float(np.mean([s[k] for s in st]))
    ws = bt.work_sums()
    n_corr = max(1, sum(s["corrections"] for s in st)); n_fact = max(1, sum(s["factorizations"] for s in st))

    if dist is not None:
        ok_t = torch.tensor([n_ok], dtype=torch.int64, device="cuda")
        dist.all_reduce(ok_t, op=dist.ReduceOp.SUM)
        n_ok_total = int(ok_t.item())
    else:
        n_ok_total = n_ok

    total_units = B * world * args.steps
    value = total_units / elapsed

    out = {
        "metric": "LCQPs/sec (batched dense n=256,nC=512,nComp=64)" if (n, nC, nComp) == (256, 512, 64)
                  else f"LCQPs/sec (batched dense n={n},nC={nC},nComp={nComp})",
        "value": value, "unit": "LCQPs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"synthetic dense batch={B}/GPU n={n} nC={nC} nComp={nComp} (BASELINE configs[2]; "
                               f"SplitMix64 seed0=0x4C43515000000001, perturbStep=0, printLevel=NONE)",
                   "global_batch": B * world, "parallelism": f"batch-sharded x{world}, no collective",
                   "solved": n_ok_total, "mean_lcqp_iterates": mean("iterTotal"), "mean_outer": mean("iterOuter"),
                   "mean_qp_trials": mean("trials"), "mean_residual_sweeps": mean("reserved"), "mean_factor_updates": mean("factorizations"),
                   "mean_backsolve_pairs": mean("corrections"), "mean_admm_iters": mean("admmIter"),
                   "mean_active_rows_per_backsolve": float(ws[0] / n_corr), "mean_factor_update_kbytes": float(ws[2] / n_fact / 1e3),
                   "setup_ms_per_step": setup_ms / args.steps, "homotopy_kernel_ms_per_step": solve_ms / args.steps},
        "roofline": {"bound": "hbm", "kernel": "k_lcqp_run", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(B, n, nC, nComp),
                     "algorithmic_bytes_per_launch": alg_bytes},
    }

    if rank == 0 and world == 1 and not args.no_pipelined:
        # not the headline: the same K steps with two batch objects in flight on two streams (step k+1 is launched while step k
        # still runs), so that the launch tail of one step -- its slowest instances -- overlaps with the bulk of the next
        bt2 = la.BatchLCQP(B, n, nC, nComp, device=local_rank, opt=opt)
        bt2.generate_synthetic(first)
        bt2.run(); bt2.synchronize()
        pair = (bt, bt2)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for k in range(max(2, args.steps)):
            pair[k % 2].run()                 # asynchronous: returns after the launches
        bt.synchronize(); bt2.synchronize()
        dtp = time.perf_counter() - tp
        x2, _, st2 = bt2.solution()
        out["pipelined"] = {"depth": 2, "steps": max(2, args.steps), "value": B * max(2, args.steps) / dtp, "unit": "LCQPs/s",
                            "ms_per_step": 1e3 * dtp / max(2, args.steps), "solved_last_step": sum(1 for s_ in st2 if s_["returnValue"] == 0),
                            "bitwise_equal_to_sequential": bool(np.array_equal(x2, x)),
                            "note": "two independent batches of the same workload in flight on two streams; every step still does setup + homotopy"}
        bt2.close()

    if rank == 0 and not args.no_backsolve:
        # the factor-once / back-solve-many kernel pair on its own: B factors of order n resident in HBM,
        # one right-hand side each (bytes_bs(N) = 8 N (N+2))
        rng = np.random.default_rng(0)
        nb = min(B, 1024)
        K = rng.standard_normal((nb, n, n)) * 0.05
        K = K + K.transpose(0, 2, 1)
        K[:, np.arange(n), np.arange(n)] += 0.1 * n
        rhs = rng.standard_normal((nb, n))
        _, ms = la.chol_solve(K, rhs, repeat=20)
        bs_bytes = nb * 8.0 * n * (n + 2)
        gbs = bs_bytes / (ms * 1e-3) / 1e9
        out["backsolve_kernel"] = {"kernel": "k_backsolve", "batch": nb, "N": n, "ms": ms, "achieved": gbs,
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
        del K, rhs

    if rank == 0 and world == 1 and args.cpu_sample > 0:
        import oracle_py as O
        threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        oopt = O.default_options(perturbStep=0, printLevel=0)
        cnt = max(args.cpu_sample, threads)      # at least one LCQP per host core
        tc = time.perf_counter()
        ok, xo, yo, so = O.synth_batch_solve(0, cnt, n, nC, nComp, opt=oopt, threads=threads)
        dtc = time.perf_counter() - tc
        t1 = time.perf_counter()
        O.synth_batch_solve(0, 4, n, nC, nComp, opt=oopt, threads=1, want_xy=False)     # one core, no contention
        single = 4 / (time.perf_counter() - t1)
        dx = float(np.abs(xo[: min(cnt, B)] - x[: min(cnt, B)]).max())
        out["cpu_baseline"] = {"value": cnt / dtc, "unit": "LCQPs/s", "cores": threads, "kind": "port",
                               "sample": f"instances 0..{cnt - 1} of the same synthetic workload, CPU oracle "
                                         f"(oracle/lcqp_oracle.c, same algorithm in scalar C; the reference's qpOASES "
                                         f"path cannot be built: external/qpOASES is empty), one LCQP per thread, "
                                         f"{ok}/{cnt} solved in {dtc:.2f} s",
                               "single_core_value": single, "max_abs_dx_vs_gpu": dx}
    bt.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Benchmark of the hot path: batched BoundMPC NLP solves on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL all-gather of the solutions)

One "step" = one pass of the hot path over one batch of synthetic problem instances:
BASELINE.json configs[2] -- 8192 instances per GPU, randomized convex-set obstacles, horizon
N = 20, iiwa14 (SURVEY.md 8(d) generator, seed 8192 + rank).  Inputs are resident in HBM when the
timed region starts (bmpc_solve_dev with device pointers).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HORIZON = 20
BATCH_PER_GPU = 8192
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VEC_PEAK_TFLOPS = 78.6    # vector FP64 peak (spec); the binding resource of this kernel


def alg_bytes_per_solve(N):
    # read x0 + write x + read p + read state0
    return 8 * (2 * (44 * N + 6) + 875 + 40)


def alg_flops_per_solve(N, iters):
    # iters * N * F_stage, F_stage = 1.2e5 (condensed stage n_x=26, n_u=9)
    return iters * N * 1.2e5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="instances per GPU")
    ap.add_argument("--horizon", type=int, default=HORIZON)
    ap.add_argument("--hess", type=int, default=None)
    ap.add_argument("--wpi", type=int, default=None, help="wavefronts per instance (1, 2, 4)")
    ap.add_argument("--bpc", type=int, default=None, help="resident workgroups per CU")
    ap.add_argument("--depth", type=int, default=3, help="solves in flight (2: the straggler tail of one batch overlaps "
                    "the bulk of the next, two handles used alternately; 1: one at a time)")
    ap.add_argument("--merge", type=int, default=2, help="8192-instance batches handed to the solver per call (they are independent: "
                    "a larger launch amortises the straggler tail over more bulk work)")
    ap.add_argument("--gate", type=float, default=1.0, help="start the next solver call when the others have < gate * their instances "
                    "active (1.0: at once -- with two batches per call holding calls back no longer pays)")
    ap.add_argument("--engine", type=int, default=None, help="0 pipeline (default), 1 persistent kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from boundplanner_amd import scenes
    from boundplanner_amd.solver import HipBoundMPC

    N, B = args.horizon, args.batch
    kw = {} if args.hess is None else {"hess": args.hess}
    if args.wpi is not None:
        kw["waves_per_instance"] = args.wpi
    if args.bpc is not None:
        kw["blocks_per_cu"] = args.bpc
    if args.engine is not None:
        kw["engine"] = args.engine
    depth = max(1, min(args.depth, 4))
    M = max(1, min(args.merge, 4))
    bes = [HipBoundMPC(N, device=local_rank, max_batch=M * B, **kw) for _ in range(depth)]
    be = bes[0]
    t0 = time.time()
    batch = scenes.make_batch(B, N, 8192 + rank, be.fk, randomize_sets=True)
    t_gen = time.time() - t0
    big = lambda a: np.nan_to_num(a, posinf=1e20, neginf=-1e20)
    # M batches per solver call: the synthetic batch repeated M times (every batch of the run is this same batch anyway)
    d = {k: torch.from_numpy(big(batch[k])).to(dev).repeat(M, 1).contiguous() for k in ("x0", "lbx", "ubx", "p")}
    n_w = be.n_w
    # one set of output buffers per handle in flight
    outs = [dict(x=torch.empty((M * B, n_w), dtype=torch.float64, device=dev), f=torch.empty(M * B, dtype=torch.float64, device=dev),
                 viol=torch.empty(M * B, dtype=torch.float64, device=dev), iters=torch.empty(M * B, dtype=torch.int32, device=dev),
                 status=torch.empty(M * B, dtype=torch.int32, device=dev)) for _ in range(depth)]
    gathered = torch.empty((world * M * B, n_w), dtype=torch.float64, device=dev) if world > 1 else None
    torch.cuda.synchronize(dev)      # inputs complete before any handle's own stream reads them
    busy = [0] * depth               # batches in the solve in flight on each handle
    kernel_ms = []

    def retire(j):
        """Wait for the solve in flight on handle j; all-gather its solutions (RCCL over xGMI)."""
        if not busy[j]:
            return
        bes[j].wait()
        m, busy[j] = busy[j], 0
        kernel_ms.append(bes[j].last_kernel_ms() / m)
        if world > 1:
            dist.all_gather_into_tensor(gathered[:world * m * B], outs[j]["x"][:m * B])
            # the next solve on this handle overwrites outs[j]["x"]: the gather must have read it (only torch's current
            # stream is waited for, the other handles' solver streams keep running)
            torch.cuda.current_stream(dev).synchronize()

    def run(nsteps):
        """nsteps batches, M per solver call (the last call takes what is left)."""
        calls, left = 0, nsteps
        while left > 0:
            m = min(M, left)
            j = calls % depth
            retire(j)
            # start the next call when the solves in flight have left their bulk phase (most of their
            # instances finished): the launch-latency-bound straggler tail of one solve then runs beside
            # the throughput-bound bulk of the next
            while any(busy[q] and bes[q].active() > args.gate * busy[q] * B for q in range(depth)):
                time.sleep(0.0005)
            o = outs[j]
            bes[j].solve_dev_async(m * B, d["x0"].data_ptr(), d["lbx"].data_ptr(), d["ubx"].data_ptr(), d["p"].data_ptr(),
                                   o["x"].data_ptr(), o["f"].data_ptr(), o["iters"].data_ptr(), o["status"].data_ptr(),
                                   o["viol"].data_ptr())
            busy[j] = m
            calls += 1
            left -= m
        for j in range(depth):
            retire((calls + j) % depth)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    run(args.warmup)
    barrier()
    kernel_ms.clear()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    x, iters, status, viol = outs[0]["x"][:B], outs[0]["iters"][:B], outs[0]["status"][:B], outs[0]["viol"][:B]

    it_np, st_np, viol_np = iters.cpu().numpy(), status.cpu().numpy(), viol.cpu().numpy()
    ok = (st_np == 0) | (viol_np < 1e-4)            # the reference's acceptance test (BoundMPC.py:617)
    total_solves = world * B * args.steps
    value = total_solves / elapsed
    # GPU time per batch: HIP events on the solver's own stream around one batch (with two batches in
    # flight they overlap, so the per-batch share of the timed region is the honest denominator)
    k_ms = float(np.mean(kernel_ms)) if depth == 1 else 1e3 * elapsed / args.steps
    ach_gbs = alg_bytes_per_solve(N) * B / (k_ms * 1e-3) / 1e9
    mean_it = float(it_np.mean())
    ach_tf = alg_flops_per_solve(N, mean_it) * B / (k_ms * 1e-3) / 1e12

    # HBM bytes of one batch from the PMC passes of tools/profile_round.sh (separate rocprofv3 --pmc runs of
    # one synchronous batch of this same workload; FETCH_SIZE raw, see the note) -- null for other sizes
    traffic, traffic_note = None, None
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tj) and (B, N) == (BATCH_PER_GPU, HORIZON):
        t = json.load(open(tj))
        traffic = t["fetch_bytes_raw"] + t["write_bytes"]
        traffic_note = ("profiles/pmc_traffic.json: FETCH_SIZE (raw, 8-byte-per-lane loads are uncalibrated on gfx950, at most 2x low) "
                        f"{t['fetch_bytes_raw'] / 1e9:.1f} GB + WRITE_SIZE {t['write_bytes'] / 1e9:.1f} GB per batch, all kernels")

    out = {
        "metric": "MPC solves/sec (whole node), iiwa14 7-DOF, N=20",
        "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[2]: {B}-batch per GPU, randomized convex-set obstacles, N={N}, "
                               "cold start, tol 1e-5, max_iter 100", "batch_per_gpu": B, "horizon": N,
                   "sharding": "independent instances per rank + RCCL all-gather of x" if world > 1 else "single GPU",
                   "hess": int(be.opts.hess), "engine": int(be.opts.engine), "solver_handles_in_flight": depth,
                   "batches_per_solver_call": M, "batches_in_flight": depth * M},
        "solver": {"iters_mean": mean_it, "iters_p50": float(np.median(it_np)), "iters_p99": float(np.percentile(it_np, 99)),
                   "iters_max": int(it_np.max()), "converged_frac": float((st_np == 0).mean()),
                   "accepted_frac": float(ok.mean()), "gen_s": t_gen},
        "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note, "kernel": "bmpc_k_ric (+ bmpc_k_eval, k_step, k_trial): one batch",
                     "kernel_ms": k_ms, "event_ms_per_batch": float(np.mean(kernel_ms)), "alg_bytes_per_solve": alg_bytes_per_solve(N),
                     "note": "not HBM- or MFMA-bound: latency/VALU/LDS-bound small-matrix IP loop (DESIGN.md); "
                             "the meaningful limiter is FP64 VALU, reported in valu_fp64"},
        "valu_fp64": {"achieved": ach_tf, "peak": FP64_VEC_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "frac": ach_tf / FP64_VEC_PEAK_TFLOPS, "alg_flops_per_solve": alg_flops_per_solve(N, mean_it)},
    }

    if rank == 0 and not args.no_cpu_baseline:
        import oracle_lib as O               # cpu_baseline leg only
        cores = len(os.sched_getaffinity(0))
        nthr = min(cores, 64)
        ns = min(B, 128 * nthr)         # ~12 s of host work at ~670 solves/s on 64 threads
        hess = int(be.opts.hess)
        t0 = time.perf_counter()
        ro = O.solve_batch(N, batch["x0"][:ns], batch["lbx"][:ns], batch["ubx"][:ns], batch["p"][:ns],
                           nthreads=nthr, hess=hess)
        tc = time.perf_counter() - t0
        x_gpu = x[:ns].cpu().numpy()
        both = (ro["status"] == 0) & (st_np[:ns] == 0)
        out["cpu_baseline"] = {
            "value": ns / tc, "unit": "solves/s", "cores": nthr, "kind": "port",
            "sample": f"first {ns} instances of the same batch, oracle/bmpc_solve.c (same algorithm, FP64, -O3 -march=x86-64-v3, "
                      f"OpenMP over instances) on {nthr} host threads in {tc:.1f} s; the reference's CasADi+IPOPT "
                      "cannot run here (no wheel, no network)",
            "iters_mean": float(ro["iters"].mean()),
            "max_abs_dx_vs_gpu": float(np.abs(ro["x"][both][:, :40 * N] - x_gpu[both][:, :40 * N]).max()) if both.any() else None,
        }
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

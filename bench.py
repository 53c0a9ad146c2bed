"""Headline benchmark: Higgs-shaped binary training (10.5 M x 28, 255 bins, 255 leaves).

    python bench.py --gpus N --steps K --warmup W

One "step" is one boosting iteration (one 255-leaf tree over the whole matrix,
including gradients, histogram building, split search, partitioning and score
update).  The data is synthetic of the Higgs shape (no network for the real
set).  With N > 1 (launched through torch.distributed.run, one rank per GPU)
the SAME 10.5 M rows are sharded over the ranks (strong scaling) and trained
data-parallel with RCCL histogram all-reduces.  Started WITHOUT a launcher
(``WORLD_SIZE`` unset) and ``--gpus N`` > 1, the script starts the N rank
processes itself - fresh children, spawned before this process has touched a
GPU - and passes rank 0's result line through.

``--precision full`` (default) times the full-precision learner (float gradients
as 36-bit fixed point in exact 64-bit histogram cells, the precision class of the
reference's float histograms); ``--precision q16`` the 16-bit fixed-point
histograms.  The other mode's 500-tree time is reported in ``config`` as well.

The reported metric is the wall-clock training time of the reference
configuration: 500 trees.  ``ms_per_step`` is the mean over the K timed steps
(after W warm-up steps).  Trees get more expensive as boosting goes on (they
grow deeper and less balanced), so K early steps times 500 would flatter the
number: when K < 500 the script ALSO trains a complete 500-tree model from a
fresh booster, bracketed by the same barrier + synchronize, and reports THAT
wall time as ``value`` (``config.value_source`` says which it is).  Lower is
better; vs_baseline = upstream LightGBM's published 238.5 s divided by ours.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

BASELINE_SECONDS = 238.5   # upstream LightGBM CPU, Higgs, 500 iterations (docs/Experiments.rst, recalled)
N_ROWS = 10_500_000
N_TREES_REF = 500


def launch_ranks(n: int) -> int:
    """Starts this script once per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) and waits.
    The parent never touches a GPU; rank 0 writes the result line to the inherited stdout."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                # a rank that died leaves the others waiting in a collective: end them (exactly these processes)
                for q in alive:
                    q.terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=N_ROWS)
    ap.add_argument("--leaves", type=int, default=255)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--stats", action="store_true", help="report round/host-time counters (no device event timers)")
    ap.add_argument("--auc", action="store_true", help="also report train AUC after the timed steps")
    ap.add_argument("--no-full-run", action="store_true",
                    help="with --steps < 500: report steps x 500 instead of also timing a complete 500-tree run")
    ap.add_argument("--extra", default="", help="JSON object of extra training parameters (experiments only: the "
                                                "reported config then names them)")
    ap.add_argument("--precision", choices=["full", "q16"], default="full",
                    help="histogram precision of the timed run: full = 36-bit fixed point of the float gradients "
                         "(use_quantized_grad=false), q16 = 16-bit fixed point")
    ap.add_argument("--no-other-precision", action="store_true", help="skip the extra 500-tree run in the other precision")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    # stdout carries exactly one line, the result: whatever libraries print on the way (RCCL announces its
    # version on stdout when a communicator comes up) goes to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    # the host driver of this pool shares device memory between processes through dmabuf only; RCCL needs
    # this before the HIP runtime starts
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LGA_BENCH_FORCE_DIST=1 takes the multi-rank code path (process group, agreed bin boundaries, RCCL
    # communicator) even with one rank - a plumbing check for boxes that have a single GPU
    distributed = world > 1 or os.environ.get("LGA_BENCH_FORCE_DIST") == "1"
    if distributed and world == 1:
        os.environ["LGA_DP_KEEP_SINGLE_RANK"] = "1"   # train through the one-rank communicator, not around it
    # LGA_BENCH_BACKEND=gloo rehearses a multi-rank run on a machine with fewer GPUs than ranks: the ranks
    # share the devices and reduce through host memory instead of RCCL (timings then mean nothing)
    backend = os.environ.get("LGA_BENCH_BACKEND", "nccl")
    # no GPU at all (and a host backend): the same flow on the CPU engine - a rehearsal of the plumbing only
    on_gpu = torch.cuda.device_count() > 0
    if not on_gpu and backend == "nccl":
        raise SystemExit("bench.py needs a GPU (LGA_BENCH_BACKEND=gloo rehearses the multi-rank flow without one)")
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"--gpus {world} asks for {world} devices, {torch.cuda.device_count()} visible")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    if on_gpu:
        torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import lightgbm_amd as lgb
    from lightgbm_amd.utils.datasets import make_higgs_like
    from lightgbm_amd.parallel import default_comm, distributed_dataset

    n_total = args.rows
    lo = n_total * rank // world
    hi = n_total * (rank + 1) // world
    X, y = make_higgs_like(hi - lo, seed=1234 + rank, device="cuda" if on_gpu else "cpu")
    params = {
        "objective": "binary", "num_leaves": args.leaves, "max_bin": 255, "learning_rate": 0.1,
        "min_data_in_leaf": 1, "min_sum_hessian_in_leaf": 100.0, "verbose": -1, "device_type": "gpu" if on_gpu else "cpu",
        "tree_learner": "data" if distributed else "serial", "profile": bool(args.profile),
        "use_quantized_grad": args.precision == "q16",
    }
    extra = json.loads(args.extra) if args.extra else {}
    params.update(extra)
    if args.auc:
        params["metric"] = ["auc", "binary_logloss"]   # evaluated once, after the timed steps
    t0 = time.time()
    if distributed:
        ds = distributed_dataset(X, y, params=params)
        comm = default_comm()
    else:
        ds = lgb.Dataset(X, label=y, params=params)
        comm = None
    ds.construct()
    if on_gpu:
        torch.cuda.synchronize()
    t_bin = time.time() - t0
    del X
    booster = lgb.Booster(params, ds, comm=comm)

    def sync():
        if distributed:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        booster.update()
    sync()
    t1 = time.time()
    for _ in range(args.steps):
        booster.update()
    sync()
    elapsed = time.time() - t1
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1000.0 / max(args.steps, 1)

    def max_over_ranks(v):
        if not distributed:
            return v
        t = torch.tensor([v], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def full_run(p):
        """A complete 500-tree training run from a fresh booster, bracketed like the timed steps."""
        bo = lgb.Booster(p, ds, comm=comm)
        sync()
        t2 = time.time()
        for _ in range(N_TREES_REF):
            bo.update()
        sync()
        return max_over_ranks(time.time() - t2), bo

    if args.steps >= N_TREES_REF or args.no_full_run:
        value = ms_per_step * N_TREES_REF / 1000.0
        value_source = ("mean of the timed steps x 500" if args.steps < N_TREES_REF
                        else "the timed steps (>= 500 trees), scaled to 500")
    else:
        # the metric itself: a complete 500-tree training run, fresh booster, same bracketing
        del booster
        value, booster = full_run(params)
        value_source = "wall time of a complete 500-tree run in this process (trees 1..500 of a fresh booster)"
    seconds_by_precision = {args.precision: round(value, 4)}
    if not (args.no_full_run or args.no_other_precision):
        other = "q16" if args.precision == "full" else "full"
        t_other, b_other = full_run({**params, "use_quantized_grad": other == "q16"})
        del b_other
        seconds_by_precision[other] = round(t_other, 4)
    dtypes = {
        "full": "fp32 gradients; histograms accumulate them exactly as 36-bit fixed point in 64-bit integer cells (finer than "
                "the reference's fp32 / fp64 float histograms and order-independent), fp64 split gains, scores and leaf values",
        "q16": "fp32 gradients quantised per iteration to 16-bit fixed point for the histograms (exact int64 sums), fp64 split "
               "gains and scores, leaf values from 36-bit sums of the fp32 gradients",
    }
    out = {
        "metric": "higgs_10.5Mx28_train_seconds_500_trees",
        "value": round(value, 4),
        "unit": "s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": False,
        "scaling": "strong",
        "vs_baseline": round(BASELINE_SECONDS / value, 3) if value > 0 else None,
        "dtype": dtypes[args.precision] + " (held-out AUC of both modes against the CPU full-precision learner: profiles/r3/auc_parity_500trees.txt)",
        "data": "synthetic Higgs-shaped (make_higgs_like), random labels from a non-linear score; no real Higgs download possible",
        "config": {"model": "gbdt binary, 255 leaves, 255 bins, lr 0.1, min_sum_hessian 100", "rows": n_total,
                   "features": 28, "trees_timed": args.steps, "precision": args.precision,
                   "seconds_500_trees_by_precision": seconds_by_precision,
                   "parallelism": (f"dp{world}" + ("" if backend == "nccl" else f" ({backend} rehearsal, shared GPU)"))
                   if distributed else "single",
                   "comm_ranks": comm.size if comm is not None else 1,
                   "comm": type(comm).__name__ if comm is not None else None,
                   "device": "MI355X (HIP engine)" if on_gpu else "cpu engine (no GPU visible: plumbing rehearsal)",
                   "binning_seconds": round(t_bin, 3), "value_source": value_source},
    }
    if extra:
        out["config"]["extra_params"] = extra
    if args.auc:
        res = dict((m, v) for _, m, v, _ in booster._eval(-1, "training"))
        out["config"]["train_metric"] = res
    if args.profile or args.stats:
        out["config"]["profile"] = booster.profile()
        out["config"]["last_tree"] = booster.grow_stats()
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if comm is not None:
        comm.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

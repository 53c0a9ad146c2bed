"""Rehearsal of the 8-GPU data-parallel benchmark on ONE GPU: eight rank processes share the device and reduce through
gloo (the device engine stages its buffers through pinned memory), each with its 1/8 shard of the 10.5 M x 28
Higgs-shaped table, bins agreed over the ranks; rank 0 then trains the serial learner on the whole table with the
same boundaries and compares the tree text.  Timings mean nothing here (one GPU, host reductions); identity does.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29561 \
        tools/runs/dp_rehearsal_full.py [--rows 10500000 --trees 12 --precision full]
"""
import argparse
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_500_000)
    ap.add_argument("--trees", type=int, default=12)
    ap.add_argument("--precision", default="full", choices=["full", "q16"])
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import lightgbm_amd as lgb
    from lightgbm_amd.parallel import default_comm
    from lightgbm_amd.parallel.data_parallel import gather_bin_sample
    from lightgbm_amd.utils.datasets import make_higgs_like
    lo, hi = args.rows * rank // world, args.rows * (rank + 1) // world
    X, y = make_higgs_like(hi - lo, seed=1234 + rank, device="cuda")
    params = {"objective": "binary", "num_leaves": 255, "max_bin": 255, "learning_rate": 0.1, "min_data_in_leaf": 1,
              "min_sum_hessian_in_leaf": 100.0, "verbose": -1, "device_type": "gpu", "tree_learner": "data",
              "use_quantized_grad": args.precision == "q16"}
    sample = gather_bin_sample(X, params)
    comm = default_comm()
    t0 = time.time()
    b = lgb.train(params, lgb.Dataset(X, label=y, params=params, bin_sample=sample), args.trees, comm=comm)
    t_dp = time.time() - t0
    text = b.model_to_string()
    cut = text[text.index("Tree=0"):text.index("end of trees")]
    digests = [None] * world
    dist.all_gather_object(digests, hashlib.sha256(cut.encode()).hexdigest())
    comm.close()
    del b, X, y
    torch.cuda.empty_cache()
    dist.barrier()
    if rank == 0:
        parts = [make_higgs_like(args.rows * (r + 1) // world - args.rows * r // world, seed=1234 + r, device="cuda") for r in range(world)]
        Xa, ya = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
        del parts
        q = {**params, "tree_learner": "serial"}
        t0 = time.time()
        s = lgb.train(q, lgb.Dataset(Xa, label=ya, params=q, bin_sample=sample), args.trees).model_to_string()
        t_serial = time.time() - t0
        scut = s[s.index("Tree=0"):s.index("end of trees")]
        rec = {"ranks": world, "rows": args.rows, "trees": args.trees, "precision": args.precision,
               "ranks_agree": all(d == digests[0] for d in digests),
               "data_parallel_equals_serial": scut == cut,
               "trees_sha256_data_parallel": digests[0], "trees_sha256_serial": hashlib.sha256(scut.encode()).hexdigest(),
               "leaves_per_tree": [int(t.count("leaf_value=") and len(t.split("leaf_value=")[1].split("\n")[0].split())) for t in scut.split("Tree=")[1:]],
               "seconds_data_parallel_shared_gpu_gloo": round(t_dp, 2), "seconds_serial": round(t_serial, 2)}
        line = json.dumps(rec)
        print(line, flush=True)
        if args.out:
            with open(args.out, "a") as f:
                f.write(line + "\n")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Distributed learners on the CPU engine: data-parallel and feature-parallel training must grow the SAME
trees as serial training (integer histograms make the reduction exact).  Covered two ways:
an in-process loop-back communicator (threads) and real torch.distributed/gloo processes."""
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

import lightgbm_amd as lgb
from lightgbm_amd.parallel import ThreadComm
from lightgbm_amd.utils.datasets import make_higgs_like, make_regression

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU = {"device_type": "cpu", "verbose": -1}


def trees_of(b):
    s = b.model_to_string() if hasattr(b, "model_to_string") else b
    return s[s.index("Tree=0"):s.index("end of trees")]


def run_ranks(P, fn):
    comms = ThreadComm.make_group(P)
    out, err = [None] * P, [None] * P

    def work(r):
        try:
            out[r] = fn(r, comms[r])
        except BaseException as e:   # noqa
            err[r] = e
            comms[r]._hub.barrier.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    [c.close() for c in comms]
    return out


@pytest.mark.parametrize("P", [2, 3, 4, 8])
def test_data_parallel_equals_serial_regression(P):
    X, y = make_regression(6000, 9, seed=0)
    params = {**CPU, "num_leaves": 31, "min_data_in_leaf": 20}
    serial = lgb.train(params, lgb.Dataset(X, label=y, bin_sample=X), 12)
    bounds = np.linspace(0, len(y), P + 1).astype(int)

    def rank(r, comm):
        sl = slice(bounds[r], bounds[r + 1])
        d = lgb.Dataset(X[sl], label=y[sl], bin_sample=X)
        ev = {}
        b = lgb.train({**params, "tree_learner": "data", "metric": "l2"}, d, 12, comm=comm, valid_sets=[d],
                      evals_result=ev)
        return b.model_to_string(), ev["training"]["l2"][-1]
    res = run_ranks(P, rank)
    for s, _ in res:
        assert trees_of(s) == trees_of(serial)
    # the distributed metric is the global one
    global_l2 = float(np.mean((serial.predict(X) - y) ** 2))
    assert all(abs(m - global_l2) < 1e-6 for _, m in res)


def test_data_parallel_l1_leaf_values_are_the_mean_of_the_local_medians():
    """Percentile leaf outputs under row sharding: same split structure as serial, every rank ends with the
    same model, and a leaf's value is the mean of the ranks' local residual medians."""
    X, y = make_regression(4000, 5, seed=4)
    params = {**CPU, "objective": "regression_l1", "num_leaves": 7, "learning_rate": 1.0, "boost_from_average": False}
    serial = lgb.train(params, lgb.Dataset(X, label=y, bin_sample=X), 1)
    half = len(y) // 2

    def rank(r, comm):
        sl = slice(0, half) if r == 0 else slice(half, None)
        b = lgb.train({**params, "tree_learner": "data"}, lgb.Dataset(X[sl], label=y[sl], bin_sample=X), 1, comm=comm)
        return b.model_to_string()
    a, b = run_ranks(2, rank)
    assert a == b
    dp = lgb.Booster(model_str=a)
    ts, td = serial.dump_model()["tree_info"][0], dp.dump_model()["tree_info"][0]

    def shape(n):
        return (n["split_feature"], n["threshold"], shape(n["left_child"]), shape(n["right_child"])) \
            if "split_index" in n else None
    assert shape(ts["tree_structure"]) == shape(td["tree_structure"])
    leaf = dp.predict(X, pred_leaf=True)[:, 0]
    y32 = y.astype(np.float32)

    def median_rule(v):       # percentile.hpp: (1 - 0.5) * n values down from the top, interpolated
        v = np.sort(v)
        n = len(v)
        if n <= 1:
            return float(v[0])
        pos = int(0.5 * n)
        bias = 0.5 * n - pos
        v1, v2 = float(v[n - pos]), float(v[n - 1 - pos])
        return v1 - (v1 - v2) * bias
    for l in np.unique(leaf):
        parts = [y32[:half][leaf[:half] == l], y32[half:][leaf[half:] == l]]
        want = np.mean([median_rule(p) for p in parts if len(p)])
        assert abs(dp.get_leaf_output(0, int(l)) - want) < 1e-12


def test_data_parallel_equals_serial_binary_first_trees_and_auc():
    X, y = make_higgs_like(8000, seed=1, device="numpy")
    params = {**CPU, "objective": "binary", "num_leaves": 15, "min_sum_hessian_in_leaf": 5.0, "metric": "auc"}
    ds = lgb.Dataset(X, label=y, bin_sample=X)
    ev0 = {}
    serial = lgb.train(params, ds, 10, valid_sets=[ds], evals_result=ev0)

    def rank(r, comm):
        sl = slice(r * 4000, (r + 1) * 4000)
        d = lgb.Dataset(X[sl], label=y[sl], bin_sample=X)
        ev = {}
        b = lgb.train({**params, "tree_learner": "data"}, d, 10, comm=comm, valid_sets=[d], evals_result=ev)
        return b.model_to_string(), ev["training"]["auc"][-1]
    res = run_ranks(2, rank)
    assert trees_of(res[0][0]) == trees_of(res[1][0]) == trees_of(serial)
    # AUC over ranks goes through a shared score grid: equal to the exact one within 1e-4
    assert abs(res[0][1] - ev0["training"]["auc"][-1]) < 1e-4 and res[0][1] == res[1][1]


@pytest.mark.parametrize("P", [2, 4])
def test_voting_parallel_with_enough_votes_equals_serial(P):
    """2 x top_k >= number of features: every histogram row is summed, so the vote changes nothing."""
    X, y = make_regression(6000, 9, seed=0)
    params = {**CPU, "num_leaves": 31, "min_data_in_leaf": 20}
    serial = lgb.train(params, lgb.Dataset(X, label=y, bin_sample=X), 10)
    bounds = np.linspace(0, len(y), P + 1).astype(int)

    def rank(r, comm):
        sl = slice(bounds[r], bounds[r + 1])
        d = lgb.Dataset(X[sl], label=y[sl], bin_sample=X)
        return lgb.train({**params, "tree_learner": "voting", "top_k": 5}, d, 10, comm=comm).model_to_string()
    for s in run_ranks(P, rank):
        assert trees_of(s) == trees_of(serial)


def test_voting_parallel_with_few_votes_agrees_across_ranks_and_keeps_the_quality():
    """top_k = 3 of 28 features: at most 6 histogram rows per node cross the ranks.  The ranks must still grow the
    same model, and - PV-Tree's claim - lose next to nothing against the learner that sums every row."""
    X, y = make_higgs_like(12000, seed=5, device="numpy")
    Xt, yt = make_higgs_like(6000, seed=6, device="numpy")
    params = {**CPU, "objective": "binary", "num_leaves": 15, "min_data_in_leaf": 20, "learning_rate": 0.1}
    from sklearn.metrics import roc_auc_score
    serial = lgb.train(params, lgb.Dataset(X, label=y, bin_sample=X), 40)
    auc_serial = roc_auc_score(yt, serial.predict(Xt))
    P = 3
    bounds = np.linspace(0, len(y), P + 1).astype(int)

    def rank(r, comm):
        sl = slice(bounds[r], bounds[r + 1])
        d = lgb.Dataset(X[sl], label=y[sl], bin_sample=X)
        return lgb.train({**params, "tree_learner": "voting", "top_k": 3}, d, 40, comm=comm).model_to_string()
    res = run_ranks(P, rank)
    assert trees_of(res[0]) == trees_of(res[1]) == trees_of(res[2])
    assert trees_of(res[0]) != trees_of(serial)          # it IS an approximation at this top_k
    auc_vote = roc_auc_score(yt, lgb.Booster(model_str=res[0]).predict(Xt))
    assert auc_vote > auc_serial - 0.01, (auc_vote, auc_serial)


@pytest.mark.parametrize("P,F", [(2, 70), (3, 100)])
def test_feature_parallel_equals_serial(P, F):
    X, y = make_regression(3000, F, seed=2)
    params = {**CPU, "num_leaves": 15}
    serial = lgb.train(params, lgb.Dataset(X, label=y), 6)

    def rank(r, comm):
        b = lgb.train({**params, "tree_learner": "feature"}, lgb.Dataset(X, label=y), 6, comm=comm)
        return b.model_to_string()
    for s in run_ranks(P, rank):
        assert trees_of(s) == trees_of(serial)


def test_feature_parallel_with_fewer_tiles_than_ranks():
    X, y = make_regression(2000, 10, seed=3)     # one 32-feature tile, two ranks: rank 1 owns nothing
    serial = lgb.train({**CPU, "num_leaves": 7}, lgb.Dataset(X, label=y), 4)
    res = run_ranks(2, lambda r, c: lgb.train({**CPU, "num_leaves": 7, "tree_learner": "feature"},
                                              lgb.Dataset(X, label=y), 4, comm=c).model_to_string())
    assert trees_of(res[0]) == trees_of(res[1]) == trees_of(serial)


def test_data_parallel_ranks_agree_on_category_sets():
    """The sorted category search runs on every rank from the reduced histograms: same sets, same trees."""
    rng = np.random.RandomState(0)
    n = 20000
    codes = rng.randint(0, 60, size=n)
    X = np.column_stack([codes, rng.randn(n, 3)])
    y = (rng.randn(60) * 2)[codes] + 0.5 * X[:, 1] + 0.3 * rng.randn(n)
    params = {**CPU, "num_leaves": 15, "min_data_in_leaf": 20, "boost_from_average": False}
    serial = lgb.train(params, lgb.Dataset(X, label=y, categorical_feature=[0], bin_sample=X), 5)
    assert "||" in "".join(str(t) for t in serial.trees_to_dataframe()["threshold"])

    def rank(r, comm):
        sl = slice(r * n // 2, (r + 1) * n // 2)
        d = lgb.Dataset(X[sl], label=y[sl], categorical_feature=[0], bin_sample=X)
        return lgb.train({**params, "tree_learner": "data"}, d, 5, comm=comm).model_to_string()
    for s in run_ranks(2, rank):
        assert trees_of(s) == trees_of(serial)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gloo_two_processes_data_parallel(tmp_path):
    out = str(tmp_path / "model_rank{}.txt")
    script = f"""
import os, sys
sys.path.insert(0, {ROOT!r})
os.environ["LGA_FORCE_HOST"] = "1"
import numpy as np, torch.distributed as dist
dist.init_process_group("gloo")
import lightgbm_amd as lgb
from lightgbm_amd.parallel import train_data_parallel
from lightgbm_amd.utils.datasets import make_regression
X, y = make_regression(5000, 8, seed=7)
r, P = dist.get_rank(), dist.get_world_size()
sl = slice(r * 2500, (r + 1) * 2500)
b = train_data_parallel({{"device_type": "cpu", "verbose": -1, "num_leaves": 15, "bin_construct_sample_cnt": 5000}},
                        X[sl], y[sl], num_boost_round=8)
b.save_model({out!r}.format(r))
dist.destroy_process_group()
"""
    port = _free_port()
    path = str(tmp_path / "worker.py")
    with open(path, "w") as f:
        f.write(script)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), path],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    m0, m1 = open(out.format(0)).read(), open(out.format(1)).read()
    assert trees_of(m0) == trees_of(m1)
    # the gathered sample holds every row here, so serial training with the same sample must match
    X, y = make_regression(5000, 8, seed=7)
    order = np.concatenate([np.arange(0, 2500), np.arange(2500, 5000)])
    serial = lgb.train({**CPU, "num_leaves": 15}, lgb.Dataset(X, label=y, bin_sample=X[order]), 8)
    assert trees_of(serial) == trees_of(m0)


def test_every_rank_falls_back_when_rank0_cannot_create_a_communicator_id(tmp_path):
    """The native communicator's bootstrap keeps all ranks on the same collective sequence: when rank 0 fails to
    create the id, every rank learns it from the one broadcast and every rank ends up with the fallback."""
    script = tmp_path / "fallback.py"
    script.write_text("""
import os, sys
import numpy as np, torch.distributed as dist
sys.path.insert(0, %r)
from lightgbm_amd.parallel import comm as C
dist.init_process_group("gloo")
rank = dist.get_rank()

class Native:
    def __init__(self, group):
        def make_id():
            if os.environ.get("FAIL_ID") == "1":
                raise RuntimeError("no librccl here")
            return np.arange(128, dtype=np.uint8)
        uid, err = C._share_unique_id(group, make_id)
        if uid is None:
            raise RuntimeError(err)
        assert (uid == np.arange(128)).all()
        if os.environ.get("FAIL_INIT_RANK") == str(rank):
            raise RuntimeError("comm init failed on this rank")
    def close(self):
        pass

import warnings
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    c = C._native_or_fallback(None, Native)
want_native = os.environ.get("FAIL_ID") != "1" and "FAIL_INIT_RANK" not in os.environ
assert isinstance(c, Native) == want_native and (want_native or isinstance(c, C.TorchComm)), type(c)
buf = np.array([rank + 1], dtype=np.int64)
if not want_native:
    c.allreduce(buf, "sum")
    assert buf[0] == 3
c.close()
dist.destroy_process_group()
print("rank", rank, "done", flush=True)
""" % ROOT)
    for extra in ({"FAIL_ID": "1"}, {"FAIL_INIT_RANK": "1"}, {}):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                            "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                           capture_output=True, text=True, timeout=120, env={**os.environ, **extra})
        assert r.returncode == 0 and r.stdout.count("done") == 2, r.stdout[-1500:] + r.stderr[-1500:]


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """``python bench.py --gpus N`` without ``WORLD_SIZE`` in the environment starts N rank processes itself (before
    touching any device) and passes rank 0's single result line through: it names N ranks and the communicator saw
    N ranks.  Run here over gloo on the CPU engine; a rank that fails ends the others and the exit code says so."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["LGA_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "30000", "--steps", "2", "--warmup", "1",
           "--leaves", "15", "--no-full-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["comm_ranks"] == 2 and out["config"]["parallelism"].startswith("dp2")
    assert out["steps"] == 2 and out["value"] > 0 and out["config"]["precision"] == "full"
    bad = subprocess.run(cmd + ["--extra", '{"num_leaves": "x"}'], capture_output=True, text=True, timeout=300, env=env)
    assert bad.returncode != 0 and not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]

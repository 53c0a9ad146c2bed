"""Extreme row weights (the search's row-count estimates are then far off): device against CPU engine, by path."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

def child():
    import numpy as np
    import lightgbm_amd as lgb
    from lightgbm_amd.utils.datasets import make_regression
    dev = os.environ["WC_DEVICE"]; rows = int(os.environ["WC_ROWS"])
    X, y = make_regression(rows, 8, seed=21)
    rng = np.random.RandomState(0)
    w = np.where(rng.rand(len(y)) < 0.02, 1e3, 1e-3)
    p = {"num_leaves": 31, "min_data_in_leaf": 50, "min_sum_hessian_in_leaf": 0.0, "verbose": -1, "device_type": dev}
    b = lgb.train(p, lgb.Dataset(X, label=y, weight=w, params=p), 2)
    s = b.model_to_string()
    print("MODEL" + json.dumps(s[s.index("Tree=0"):s.index("end of trees")]))

if os.environ.get("WC_DEVICE"):
    child(); sys.exit(0)

def run(dev, rows, **env):
    e = dict(os.environ, WC_DEVICE=dev, WC_ROWS=str(rows), **env)
    out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, capture_output=True, text=True, timeout=400)
    for line in out.stdout.splitlines():
        if line.startswith("MODEL"):
            return json.loads(line[5:])
    raise RuntimeError(out.stdout[-2000:] + out.stderr[-2000:])

for rows in (1500, 120000):
    cpu = run("cpu", rows)
    for v in [{}, {"LGA_ROW_LINES": "0"}, {"LGA_ROW_LINES": "1"}]:
        g = run("gpu", rows, **v)
        first = ""
        if g != cpu:
            for i, (x, y) in enumerate(zip(cpu.split("\n"), g.split("\n"))):
                if x != y:
                    first = " first difference at line %d:\n   cpu %s\n   gpu %s" % (i, x[:150], y[:150]); break
        print(rows, v, "equal:", g == cpu, first, flush=True)

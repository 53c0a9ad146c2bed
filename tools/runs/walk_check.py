"""Large-data paths of the device engine (>= 2^20 rows: row lines, rounds in flight, score update by tree walk)
against the CPU engine: which switch, if any, changes the model.  python tools/runs/walk_check.py [rows]"""
import os, sys, subprocess, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
ROWS = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1_200_000

def child():
    import lightgbm_amd as lgb
    from lightgbm_amd.utils.datasets import make_regression
    dev = os.environ["WC_DEVICE"]
    X, y = make_regression(ROWS, 12, seed=3)
    p = {"num_leaves": 63, "min_data_in_leaf": 20, "verbose": -1, "device_type": dev}
    b = lgb.train(p, lgb.Dataset(X, label=y, params=p), 4)
    s = b.model_to_string()
    print("MODEL" + json.dumps(s[s.index("Tree=0"):s.index("end of trees")]))

if os.environ.get("WC_DEVICE"):
    child()
    sys.exit(0)

def run(dev, **env):
    e = dict(os.environ, WC_DEVICE=dev, **env)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), str(ROWS)], env=e, capture_output=True, text=True, timeout=400)
    for line in out.stdout.splitlines():
        if line.startswith("MODEL"):
            return json.loads(line[5:])
    raise RuntimeError(out.stdout[-2000:] + out.stderr[-2000:])

cpu = run("cpu")
variants = [{}, {"LGA_ROUNDS_IN_FLIGHT": "1"}, {"LGA_ROW_LINES": "0"}, {"LGA_CERTAIN_SPLITS": "0"}, {"LGA_SCORE_UPDATE": "leaf"},
            {"LGA_ROUNDS_IN_FLIGHT": "1", "LGA_ROW_LINES": "0", "LGA_SCORE_UPDATE": "leaf"}]
for v in variants:
    g = run("gpu", **v)
    same = g == cpu
    first = ""
    if not same:
        a, b = cpu.split("\n"), g.split("\n")
        for i, (x, y) in enumerate(zip(a, b)):
            if x != y:
                first = " first difference at line %d:\n   cpu %s\n   gpu %s" % (i, x[:160], y[:160])
                break
    print(v, "equal to the CPU engine:", same, first, flush=True)

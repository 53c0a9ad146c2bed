"""Degenerate and extreme inputs (tools/edge_cases.py) on the CPU engine; the GPU tier repeats them on the
device, first under the index-checking build."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import edge_cases  # noqa: E402

CASES = list(edge_cases.cases())


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_cpu_engine_survives(case):
    trees = edge_cases.train_case(case, "cpu")
    assert trees.startswith("Tree=0")


@pytest.mark.gpu
@pytest.mark.parametrize("debug_lib", ["1", "0"])
def test_device_engine_survives_and_matches_the_cpu(debug_lib):
    """All cases in one child process: first with the index-checking build (a bad access would be reported,
    not executed), then with the release build.  Every case must grow the CPU engine's trees."""
    import subprocess
    env = dict(os.environ, LGA_DEBUG_LIB=debug_lib)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "..", "tools", "edge_cases.py"), "gpu"],
                       env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0 and "all cases passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]

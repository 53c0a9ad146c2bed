"""Builds the host core + tests/native/host_driver.cpp with AddressSanitizer and UBSan and runs it.

    python tools/sanitize_host.py            # exit code 0: checks passed, sanitizers silent

The sanitizer runtime is linked statically, so the binary does not depend on preload order.
"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from lightgbm_amd.build import CSRC, HOST_ONLY, HOST_SOURCES   # noqa: E402


def main() -> int:
    out_dir = ROOT / "build" / "asan"
    out_dir.mkdir(parents=True, exist_ok=True)
    flags = ["-O1", "-g", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-fno-omit-frame-pointer",
             "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-static-libasan", "-static-libubsan",
             "-I", str(CSRC)]
    objs = []
    procs = []
    for src in list(HOST_SOURCES) + list(HOST_ONLY) + ["../../tests/native/host_driver.cpp"]:
        sp = (CSRC / src).resolve()
        obj = out_dir / (sp.name + ".o")
        objs.append(str(obj))
        if not obj.exists() or obj.stat().st_mtime < max(sp.stat().st_mtime, max(p.stat().st_mtime for p in CSRC.glob("*.hpp"))):
            procs.append(subprocess.Popen(["g++", *flags, "-c", str(sp), "-o", str(obj)]))
            if len(procs) >= 4:
                if procs.pop(0).wait() != 0:
                    return 1
    for p in procs:
        if p.wait() != 0:
            return 1
    exe = out_dir / "host_driver"
    subprocess.check_call(["g++", *flags, *objs, "-o", str(exe), "-ldl", "-lpthread"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               OMP_NUM_THREADS="4")
    return subprocess.call([str(exe)], env=env)


if __name__ == "__main__":
    sys.exit(main())

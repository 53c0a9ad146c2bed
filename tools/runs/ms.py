"""stdin: bench.py's JSON line -> 'ms_per_step value' (for sweeps)."""
import json
import sys
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print(f"{d['ms_per_step']:.3f} ms/tree  rounds(last) {d['config'].get('last_tree', {}).get('rounds')}  value {d['value']}")

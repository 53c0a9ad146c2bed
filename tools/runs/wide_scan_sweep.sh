#!/bin/bash
# the split search's workgroup target (LGA_SCAN_WGS) on upstream's wide GPU-benchmark shapes; "default" = the engine's rule
for w in default 384 640 768; do
  echo "== LGA_SCAN_WGS=$w"
  if [ "$w" = default ]; then unset LGA_SCAN_WGS; else export LGA_SCAN_WGS=$w; fi
  timeout -k 10 300 python tools/runs/wide_shapes.py --trees 20 2>/dev/null | grep -v Expo | cut -c1-100 || exit 1
done

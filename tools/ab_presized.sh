#!/bin/bash
keep=$(mktemp); cp g4splat_amd/libg4s_hip.so "$keep"
for v in "$@"; do
  cp "$v" g4splat_amd/libg4s_hip.so
  echo "== $v"
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-timing --sustained-seconds 0 --views-in-flight 0 2>/dev/null | python -c '
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); print("headline ms/step %.4f"%d["ms_per_step"])'
  python tools/views_in_flight.py --steps 200 --max 2 2>/dev/null | grep "views in flight"
done
cp "$keep" g4splat_amd/libg4s_hip.so

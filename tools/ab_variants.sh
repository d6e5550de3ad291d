#!/bin/bash
# A/B timing of library variants: tools/ab_variants.sh var/A.so var/B.so ...   (run on the GPU box)
# Each variant is copied over g4splat_amd/libg4s_hip.so and bench.py prints its per-kernel ms.
set -u
keep=$(mktemp)
cp g4splat_amd/libg4s_hip.so "$keep"
for v in "$@"; do
  cp "$v" g4splat_amd/libg4s_hip.so
  touch g4splat_amd/libg4s_hip.so
  echo "== $v"
  python bench.py --steps 24 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c '
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); k=d.get("kernels_ms",{})
        print("ms/step %.3f"%d["ms_per_step"], " ".join("%s=%.3f"%(n,v) for n,v in k.items()))'
done
cp "$keep" g4splat_amd/libg4s_hip.so

#!/bin/bash
# A/B timing of library options: tools/ab_options.sh [--workload s1] "side_stream=0" "side_stream=1" ...   (GPU box)
set -u
WL=""
if [ "$1" = "--workload" ]; then WL="--workload $2"; shift 2; fi
for o in "$@"; do
  echo "== $o"
  G4S_BENCH_OPTIONS="$o" python bench.py $WL --steps 24 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c '
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); k=d.get("kernels_ms",{})
        print("ms/step %.4f"%d["ms_per_step"], " ".join("%s=%.3f"%(n,v) for n,v in k.items()))'
done

#!/bin/bash
# rocprofv3 kernel-trace stats only (quick). Usage: tools/stats_gpu.sh <tag> [workload] [extra env]
TAG=${1:-x}; WL=${2:-s3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/stats_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $REPO/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $OUT/log.txt 2>&1
python - <<PY
import csv,glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(f"{r['Name'].split('(')[0].replace('g4s::','')[:44]:46s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY

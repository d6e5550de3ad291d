#!/bin/bash
# Kernel timeline (start, duration, gap to the previous kernel) of one steady-state step for library variants:
#   tools/ab_timeline.sh var/A.so var/B.so ...     (run on the GPU box; raw traces stay in /tmp)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
keep=$(mktemp); cp $REPO/g4splat_amd/libg4s_hip.so "$keep"
export TMPDIR=/tmp
for v in "$@"; do
  tag=$(basename "$v" .so)
  cp "$REPO/$v" $REPO/g4splat_amd/libg4s_hip.so
  rm -rf /tmp/tr_$tag
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -- python $REPO/bench.py --steps 6 --warmup 3 \
      --no-cpu-baseline --no-kernel-timing --sustained-seconds 0 --views-in-flight 0 > /tmp/tr_$tag.log 2>&1)
  f=$(find /tmp/tr_$tag -name '*kernel_trace.csv' | head -1)
  echo "== $v"
  python $REPO/tools/timeline.py "$f" 2
done
cp "$keep" $REPO/g4splat_amd/libg4s_hip.so

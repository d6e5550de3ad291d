#!/bin/bash
# A/B of library variants: bit-identity (tools/grad_digest.py) + kernel times (bench.py, S3 and S2), two alternating rounds
# usage: tools/ab_digest_and_time.sh out.txt var/a.so var/b.so ...
out=gpurun_out/${ROUND:-r06}/$1; shift
mkdir -p gpurun_out/${ROUND:-r06}; : > $out
keep=$(mktemp); cp g4splat_amd/libg4s_hip.so "$keep"
for v in "$@"; do
  cp "$v" g4splat_amd/libg4s_hip.so; touch g4splat_amd/libg4s_hip.so
  echo "== $v digests" >> $out
  timeout 600 python tools/grad_digest.py 2>&1 | grep -v amdgpu.ids >> $out
done
for rep in 1 2; do for v in "$@"; do
  cp "$v" g4splat_amd/libg4s_hip.so; touch g4splat_amd/libg4s_hip.so
  for wl in s3 s2; do
  timeout 600 python bench.py --workload $wl --steps 24 --warmup 8 --no-cpu-baseline --sustained-seconds 0 --views-in-flight 0 2>/dev/null | python -c '
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); k=d.get("kernels_ms",{})
        print("%-16s %s ms/step %.4f"%(sys.argv[1], sys.argv[2], d["ms_per_step"]), " ".join("%s=%.4f"%(n,v) for n,v in k.items() if n in sys.argv[3].split(",")))' $v $wl "${KERNELS:-blend_fwd,blend_bwd,preprocess_bwd,preprocess_fwd}" >> $out
  done
done; done
cp "$keep" g4splat_amd/libg4s_hip.so
cat $out

#!/bin/bash
# round 6: AFFINE_TOL in {1e-5, 2e-5, 4e-5} -- certified share, flips, errors (tools/affine_tol_sweep.py) and kernel times (bench.py)
# for each variant library var/tol_<t>.so (built in the container: EXTRA=-DG4S_AFFINE_TOL=<t>f tools/build_variant.sh preprocess=... )
mkdir -p gpurun_out/r06
keep=$(mktemp); cp g4splat_amd/libg4s_hip.so "$keep"
out=gpurun_out/r06/affine_tol.txt; : > $out
rm -f /tmp/g4s_s3t_scene.npz
for t in 4e-5 2e-5 1e-5; do
  cp var/tol_$t.so g4splat_amd/libg4s_hip.so; touch g4splat_amd/libg4s_hip.so
  echo "== AFFINE_TOL $t, gate MARGIN $t" >> $out
  python tools/affine_tol_sweep.py $t s1 s3 s3t 2>&1 | grep '^{' >> $out
  if [ $t != 1e-5 ]; then
    echo "== AFFINE_TOL $t, gate MARGIN 1e-5 (round 4's)" >> $out
    python tools/affine_tol_sweep.py 1e-5 s1 s3 2>&1 | grep '^{' >> $out
  fi
  for rep in 1 2; do for wl in s3 s1 s2; do
    python bench.py --workload $wl --steps 24 --warmup 8 --no-cpu-baseline --sustained-seconds 0 --views-in-flight 0 2>/dev/null | python -c '
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); k=d.get("kernels_ms",{})
        print("   bench", sys.argv[1], "ms/step %.4f"%d["ms_per_step"], " ".join("%s=%.4f"%(n,v) for n,v in k.items()))' $wl >> $out
  done; done
done
cp "$keep" g4splat_amd/libg4s_hip.so
cat $out | cut -c1-400

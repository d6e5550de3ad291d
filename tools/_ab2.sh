mkdir -p gpurun_out/r04
keep=$(mktemp); cp g4splat_amd/libg4s_hip.so "$keep"
for v in var/vE_tree.so var/vK.so var/vE_tree.so var/vK.so; do
  cp "$v" g4splat_amd/libg4s_hip.so; touch g4splat_amd/libg4s_hip.so
  echo "== $v"
  python bench.py --steps 24 --warmup 8 --no-cpu-baseline --views-in-flight 0 --sustained-seconds 0.3 2>/tmp/err.txt | python -c '
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); k=d.get("kernels_ms",{})
        print("ms/step %.3f"%d["ms_per_step"], " ".join("%s=%.3f"%(n,v) for n,v in k.items()))'
done
cp "$keep" g4splat_amd/libg4s_hip.so

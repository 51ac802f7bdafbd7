# conv time and throughput of the main models after a kernel or table change: scratch/bench3.sh [models...]
set -e
D=gpurun_out/quick; mkdir -p $D
for m in ${@:-v5s v3 v7pose v8s}; do
  timeout -k 10 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $D/bench_$m.json 2> $D/bench_$m.err
  python - $D/bench_$m.json <<'PY'
import json,sys
r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
rf=r["roofline"]
print(r["config"]["workload"][:40], "ms/step", r["ms_per_step"], "value", r["value"], "conv ms", rf.get("conv_ms_in_pipeline"), "TF", rf["achieved"], "frac", rf["frac"])
PY
done

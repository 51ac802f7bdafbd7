set -e
D=gpurun_out/quick; mkdir -p $D
for m in v5s v8s; do for f in 0 1; do
  if [ $f = 1 ]; then export YAMD_NO_STEM_FUSION=1; else unset YAMD_NO_STEM_FUSION; fi
  timeout -k 10 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $D/ab_$m$f.json 2> $D/ab.err
  python - $D/ab_$m$f.json $f <<'PY'
import json,sys
r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
rf=r["roofline"]
print("no-fusion" if sys.argv[2]=="1" else "fused    ", r["config"]["workload"][:28], "ms/step", r["ms_per_step"], "value", r["value"], "conv ms", rf.get("conv_ms_in_pipeline"), "isolated", rf.get("conv_ms_isolated"))
PY
done; done

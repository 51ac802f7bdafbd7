set -e
D=gpurun_out/quick; mkdir -p $D
for m in "$@"; do for f in "" "--two-stream"; do
  timeout -k 10 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-extras $f > $D/ts_$m.json 2> $D/ts.err || { tail -5 $D/ts.err; exit 1; }
  python - $D/ts_$m.json "$f" <<'PY'
import json,sys
r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
rf=r["roofline"]
print("%-12s" % (sys.argv[2] or "one-stream"), r["config"]["workload"][:28], "ms/step", r["ms_per_step"], "value", r["value"], "conv ms in pipeline", rf.get("conv_ms_in_pipeline"), "isolated", rf.get("conv_ms_isolated"), r["stages_ms"])
PY
done; done

set -e
D=gpurun_out/quick; mkdir -p $D
run() {
  timeout -k 10 300 python bench.py --model v5s --steps 20 --warmup 5 --no-cpu-baseline --no-extras --two-stream > $D/p.json 2> $D/p.err || { tail -5 $D/p.err; exit 1; }
  python - $D/p.json "$1" <<'PY'
import json,sys
r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
rf=r["roofline"]
print("%-28s" % sys.argv[2], "ms/step", r["ms_per_step"], "value", r["value"], "conv ms in pipeline", rf.get("conv_ms_in_pipeline"), "isolated", rf.get("conv_ms_isolated"))
PY
}
run "default prios"
YAMD_MAIN_PRIO=-1 run "main high (-1), post 0"
YAMD_MAIN_PRIO=0 run "main new stream 0, post 0"
YAMD_MAIN_PRIO=-1 YAMD_POST_PRIO=-1 run "both high"
YAMD_MAIN_PRIO=0 YAMD_POST_PRIO=-1 run "main 0 post high"

#!/bin/bash
# bench.py for every model of the registry, one JSON line each (no CPU baseline): usage scratch/bench_all.sh out.txt
out=$1; : > $out
for m in v5s v8s v3 v4 v7pose v6face v5face v5pose v7 v7face v6 v6pose v5m v8m v8n v6facen v6m v6l v6facem v6posel v7x v5facem v5posex; do
  timeout -k 10 240 python bench.py --model $m --steps 8 --warmup 3 --no-cpu-baseline --no-extras --realistic > /tmp/b_$m.json 2> /tmp/b_$m.err || { echo "$m FAILED" >> $out; tail -3 /tmp/b_$m.err >> $out; exit 1; }
  python - $m <<'PY' >> $out
import json, sys
d = json.loads(open(f"/tmp/b_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("%-8s B=%-4d %9.1f images/s  %8.3f ms/step  conv %7.3f ms (%6.1f TFLOP/s)  decode %6.0f GB/s  stages %s  kept/img %.0f" % (
    sys.argv[1], d["config"]["images_per_rank"][0], d["value"], d["ms_per_step"], d["stages_ms"]["conv"], d["roofline"]["achieved"],
    d["decode_hbm"]["achieved_GBs"], {k: v for k, v in d["stages_ms"].items() if k != "note"}, d["config"]["kept_per_image"]))
r = d.get("nms_regime_realistic", {})
if "images_per_s" in r:
    print("         realistic regime (%s): %9.1f images/s  %8.3f ms/step  candidates/img %.0f  kept/img %.1f  nms %.3f ms" % (
        r["weights"], r["images_per_s"], r["ms_per_step"], r["candidates_per_image"], r["kept_per_image"], r["nms_ms"]))
PY
  tail -2 $out
done

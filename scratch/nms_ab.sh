#!/bin/bash
# NMS stage time of some models: usage nms_ab.sh tag model...
tag=$1; shift; for m in "$@"; do
  timeout -k 10 240 python bench.py --model $m --steps 8 --warmup 3 --no-cpu-baseline --no-extras > /tmp/n_$m.json 2> /tmp/n_$m.err || { tail -3 /tmp/n_$m.err; exit 1; }
  python - $m $tag <<'PY'
import json, sys
d = json.loads(open(f"/tmp/n_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[1], "nms ms", d["stages_ms"]["nms"], "step", d["ms_per_step"], "value", d["value"])
PY
done

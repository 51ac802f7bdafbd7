"""NMS regime of a model's benchmark workload: python scratch/show_regime.py <bench.json>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: v for k, v in d["config"].items() if k != "workload"})
print(d["nms"])

# quick look after a conv-kernel change: ablation rows, kernel parity tests, conv time of three models
set -e
D=gpurun_out/quick; mkdir -p $D
timeout -k 10 120 python scratch/conv_ablate.py run 1 1 1024 1024 40 20 2 4 64 4 128 > $D/ablate.txt 2>&1
timeout -k 10 120 python scratch/conv_ablate.py run 3 1 256 512 38 20 2 4 16 4 >> $D/ablate.txt 2>&1
timeout -k 10 120 python scratch/conv_ablate.py run 3 2 256 512 76 20 2 4 16 4 >> $D/ablate.txt 2>&1
grep -v amdgpu.ids $D/ablate.txt
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py -x -q -m gpu --timeout 120 --timeout-method thread > $D/pytest.txt 2>&1 || { tail -30 $D/pytest.txt; exit 1; }
tail -2 $D/pytest.txt
for m in v5s v3 v7pose; do
  timeout -k 10 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $D/bench_$m.json 2> $D/bench_$m.err
  python - $D/bench_$m.json <<'PY'
import json,sys
r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
rf=r["roofline"]
print(r["config"]["workload"][:40], "ms/step", r["ms_per_step"], "value", r["value"], "conv ms", rf.get("conv_ms_in_pipeline"), "TF", rf["achieved"], "frac", rf["frac"])
PY
done

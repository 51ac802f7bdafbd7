set -e
D=gpurun_out/quick; mkdir -p $D
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py -x -q -m gpu --timeout 120 --timeout-method thread > $D/pytest.txt 2>&1 || { tail -30 $D/pytest.txt; exit 1; }
tail -2 $D/pytest.txt
timeout -k 10 120 python scratch/conv_ablate.py run 3 1 128 256 100 20 2 4 16 8 64 2 2>&1 | grep -E "tiling|product"
timeout -k 10 120 python scratch/conv_ablate.py run 3 1 256 512 38 20 2 4 16 8 256 2 2>&1 | grep -E "tiling|product"
timeout -k 10 120 python scratch/conv_ablate.py run 1 1 1024 1024 40 20 2 4 32 8 128 2 2>&1 | grep -E "tiling|product"
timeout -k 10 120 python scratch/conv_ablate.py run 3 1 64 64 80 20 1 2 32 8 256 2 2>&1 | grep -E "tiling|product"
timeout -k 10 120 python scratch/conv_ablate.py run 3 1 64 64 80 20 1 2 32 8 256 1 2>&1 | grep -E "tiling|product"

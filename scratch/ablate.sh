set -e
mkdir -p gpurun_out/ablate
timeout -k 10 120 python scratch/conv_ablate.py run 1 1 1024 1024 40 20 2 4 64 4 128 > gpurun_out/ablate/out.txt 2>&1
timeout -k 10 120 python scratch/conv_ablate.py run 3 1 256 512 38 20 2 4 16 4 >> gpurun_out/ablate/out.txt 2>&1
timeout -k 10 120 python scratch/conv_ablate.py run 3 2 256 512 76 20 2 4 16 4 >> gpurun_out/ablate/out.txt 2>&1
cat gpurun_out/ablate/out.txt

set -e
D=gpurun_out/quick; mkdir -p $D; : > $D/ablate3.txt
run() { timeout -k 10 120 python scratch/conv_ablate.py run "$@" >> $D/ablate3.txt 2>&1; }
ABLATE_ZEROS=1 run 3 1 256 512 38 20 2 4 16 4
run 3 1 256 512 38 20 2 4 16 4
ABLATE_ZEROS=1 run 1 1 1024 1024 40 20 2 4 64 4 128
run 1 1 1024 1024 40 20 2 4 64 4 128
grep -v amdgpu.ids $D/ablate3.txt

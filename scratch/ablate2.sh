set -e
D=gpurun_out/quick; mkdir -p $D; : > $D/ablate2.txt
run() { timeout -k 10 120 python scratch/conv_ablate.py run "$@" >> $D/ablate2.txt 2>&1; }
run 3 1 256 512 38 20 2 4 16 8 256 2
run 3 1 256 512 38 20 2 4 16 4 256 2
run 1 1 1024 1024 40 20 2 4 64 4 128 2
run 1 1 1024 1024 40 20 2 4 32 8 128 2
run 3 2 256 512 76 20 2 4 16 4 256 2
run 3 2 256 512 76 20 2 4 16 8 256 2
grep -v amdgpu.ids $D/ablate2.txt

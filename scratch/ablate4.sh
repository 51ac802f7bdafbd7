set -e
D=gpurun_out/quick; mkdir -p $D; : > $D/ablate4.txt
for cfg in "4 4 16 4 64 2" "4 4 16 4 64 1"; do
  timeout -k 10 120 python scratch/conv_ablate.py run 3 1 128 256 100 20 $cfg 2>&1 | grep -v amdgpu.ids | grep -E "tiling|product|MFMA|rror" >> $D/ablate4.txt
done
cat $D/ablate4.txt

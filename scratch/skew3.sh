set -e
for sk in 0 4000 8000 16000 32000 64000; do
  echo "slot skew $sk"
  CONV_SKEW_BLOCKS=1024 CONV_SKEW=$sk timeout -k 10 120 python scratch/conv_ablate.py run 3 1 128 256 76 16 2 4 16 4 256 1 2>&1 | grep product
  CONV_SKEW_BLOCKS=1024 CONV_SKEW=$sk timeout -k 10 120 python scratch/conv_ablate.py run 3 1 64 128 152 16 2 4 16 4 256 1 2>&1 | grep product
done

set -e
for sk in 0 -5000 -15000 -30000 -60000 -120000; do
  echo "cu skew $sk"
  CONV_SKEW_BLOCKS=256 CONV_SKEW=$sk timeout -k 10 120 python scratch/conv_ablate.py run 3 1 128 256 76 16 2 4 16 8 256 2 2>&1 | grep product
  CONV_SKEW_BLOCKS=512 CONV_SKEW=$sk timeout -k 10 120 python scratch/conv_ablate.py run 3 1 64 128 152 16 2 4 16 4 256 1 2>&1 | grep product
done

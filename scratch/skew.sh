set -e
D=gpurun_out/quick; mkdir -p $D; : > $D/skew.txt
for sk in 0 20000 60000; do
  echo "all blocks skew $sk" >> $D/skew.txt
  CONV_SKEW_BLOCKS=100000000 CONV_SKEW=$sk timeout -k 10 120 python scratch/conv_ablate.py run 3 1 256 512 38 20 2 4 16 4 2>&1 | grep product >> $D/skew.txt
done
cat $D/skew.txt

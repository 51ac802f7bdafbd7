set -e
for rep in 1 2; do
for cfg in "3 1 128 256 100 20 2 4 16 8 64 2" "3 1 256 512 38 20 2 4 16 8 256 2" "3 1 512 1024 19 20 2 4 16 8 256 2"; do
  for t in A B; do echo -n "$t "; ABLATE_TAG=$t timeout -k 10 120 python scratch/conv_ablate.py run $cfg 2>&1 | grep -E "product"; done
done; done

set -e
export ACT=2
for r in "" 1; do
export RES=$r
timeout -k 10 60 python scratch/conv_bench.py 3 1 128 256 76 256 20 2 4 16 8 2
timeout -k 10 60 python scratch/conv_bench.py 3 1 256 512 38 256 20 2 4 32 8 1
timeout -k 10 60 python scratch/conv_bench.py 3 1 512 1024 19 256 20 2 4 16 8 2
timeout -k 10 60 python scratch/conv_bench.py 3 1 64 128 152 256 20 2 4 16 4 1
timeout -k 10 60 python scratch/conv_bench.py 3 1 32 64 304 256 10 2 2 32 4 1
done

#!/bin/bash
# Per-layer counters of a model's conv program: durations (kernel trace) + MFMA-busy / wave-cycle counters + HBM bytes,
# each in its own rocprofv3 run.  usage: scratch/layer_counters.sh <model> <batch> <tag>
set -e
model=$1; batch=$2; tag=$3
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/scratch/run_program.py $model $batch 4 $O/ops.json > $O/trace.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python3 $R/scratch/run_program.py $model $batch 2 $O/ops.json > $O/pmc_sq.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/scratch/run_program.py $model $batch 2 $O/ops.json > $O/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/scratch/run_program.py $model $batch 2 $O/ops.json > $O/pmc_write.log 2>&1
cd $R
python3 scratch/layer_table.py $O $O/conv_hbm_traffic.json > $O/layers.txt
head -5 $O/layers.txt; tail -4 $O/layers.txt
# keep the merge small: the raw csv files are large
rm -rf $O/trace $O/pmc_sq $O/pmc_fetch $O/pmc_write

#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the bench command, into gpurun_out/prof/.
# Counters are collected in their own passes with --kernel-trace only (no sys/hip/hsa tracing), FETCH_SIZE and
# WRITE_SIZE separately (they do not fit one pass), as MI355X_MICROARCH.md prescribes.  The program after `--` is python3 itself.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 200 --warmup 20 --launch native --no-cpu-baseline --no-kernel-breakdown --no-pooled-only --no-secondary --no-secondary-shapes --repeats 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_graph -- python3 $R/bench.py --steps 200 --warmup 20 --launch graph --no-cpu-baseline --no-kernel-breakdown --no-pooled-only --no-secondary --no-secondary-shapes --repeats 1 > $OUT/trace_graph.log 2>&1 || exit 1
C5="--batch 128 --seq 300 --din 600 --hidden 300 --prune-k 2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5 -- python3 $R/bench.py $C5 --steps 100 --warmup 10 --launch native --no-cpu-baseline --no-kernel-breakdown --no-pooled-only --no-secondary --no-secondary-shapes --repeats 1 > $OUT/trace_c5.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5_packed -- python3 $R/bench.py $C5 --lengths tacred --layout packed --steps 100 --warmup 10 --launch native --no-cpu-baseline --no-kernel-breakdown --no-pooled-only --no-secondary-shapes --repeats 1 > $OUT/trace_c5_packed.log 2>&1 || exit 1
PMC="--steps 30 --warmup 5 --launch native --no-cpu-baseline --no-kernel-breakdown --no-pooled-only --no-secondary --no-secondary-shapes --repeats 1"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py $PMC > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py $PMC > $OUT/write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/sq -- python3 $R/bench.py $PMC > $OUT/sq.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/mfma -- python3 $R/bench.py $PMC > $OUT/mfma.log 2>&1 || echo "mfma pass failed (non-fatal)"
echo profile done

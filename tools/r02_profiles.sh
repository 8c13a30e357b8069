#!/bin/bash
# Round-2 profile collection (GPU box, through gpurun); raw outputs under gpurun_out/r02/prof,
# the summaries that get committed are written by tools/r02_profiles_summarise.py.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=gpurun_out/r02/prof; mkdir -p $P
# 1. steady-state kernel trace of the default bench command (timed steps bracketed by markers)
MF_BENCH_MARK=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/bench -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $P/bench.log 2>&1
# 2. HBM counters, separate passes (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2): ICC loop and one predict
WHAT=icc REPS=2 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/icc_fetch -o i -- python tools/prof_icc.py > $P/icc_fetch.log 2>&1
WHAT=icc REPS=2 timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/icc_write -o i -- python tools/prof_icc.py > $P/icc_write.log 2>&1
WHAT=predict REPS=6 CUDNN_BENCH=1 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/pred_fetch -o p -- python tools/prof_icc.py > $P/pred_fetch.log 2>&1
WHAT=predict REPS=6 CUDNN_BENCH=1 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/pred_write -o p -- python tools/prof_icc.py > $P/pred_write.log 2>&1
# 3. MFMA pipe utilisation of the network's convolutions / GEMMs (stock MIOpen + rocBLAS + k_sc_gemm)
WHAT=predict REPS=6 CUDNN_BENCH=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $P/pred_mfma -o p -- python tools/prof_icc.py > $P/pred_mfma.log 2>&1
python tools/r02_profiles_summarise.py $P
cp $P/bench/bench_kernel_stats.csv $P/summary/r02_bench_full_run_kernel_stats.csv 2>/dev/null
# only the summaries travel back (gpurun merges <= 64 MiB)
for d in bench icc_fetch icc_write pred_fetch pred_write pred_mfma; do rm -rf $P/$d; done
ls $P $P/summary; tail -1 $P/bench.log | cut -c1-300

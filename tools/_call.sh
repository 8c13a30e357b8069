cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04q; mkdir -p $O
echo "=== bench steady profile (10 timed steps between markers)"
MF_BENCH_MARK=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o p -- python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-latency-probe > $O/bench_prof.json 2> $O/bench_prof.err; echo "rc $?"
MF_MARK=k_icc_scene_setup python tools/kernel_stats.py $O/prof_bench > $O/bench_steady_kernel_stats.csv; head -30 $O/bench_steady_kernel_stats.csv | cut -c1-140; rm -rf $O/prof_bench
echo "=== bench scenes8"; timeout 900 python bench.py --steps 10 --warmup 3 --scenes-per-gpu 8 --no-cpu-baseline --no-latency-probe > $O/bench_scenes8.json 2> $O/bench_scenes8.err; echo "rc $?"; cut -c1-200 $O/bench_scenes8.json
echo "=== bench bf16"; timeout 900 python bench.py --steps 20 --warmup 5 --dtype bf16 --no-cpu-baseline --no-latency-probe > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "rc $?"; cut -c1-200 $O/bench_bf16.json
echo "=== mfma pmc bf16"; bash tools/gpu_call.sh r04q "pmc=bf16_mfma=SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE,SQ_INSTS_VALU_MFMA_MOPS_BF,SQ_BUSY_CYCLES=python+tools/time_gemm_bf16.py+16+--no-stock" 2>&1 | tail -3 | cut -c1-300

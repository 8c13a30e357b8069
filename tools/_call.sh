cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MF_MARK=k_icc_scene_setup
bash tools/gpu_call.sh r05j "prof=bench=MF_BENCH_MARK=1+python+bench.py+--no-cpu-baseline+--no-latency-probe+--steps+10" > gpurun_out/r05j_prof.log 2>&1

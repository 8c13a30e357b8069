cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04n; mkdir -p $O
echo "=== tests"; timeout 1200 python -m pytest tests/test_gpu_backbone2d.py tests/test_gpu_predict_parity.py tests/test_gpu_checkpoint.py tests/test_gpu_training.py tests/test_gpu_bf16_kernels.py tests/test_gpu_conv3d.py -m gpu -q > $O/tests.log 2>&1; echo "rc $?"; tail -4 $O/tests.log | cut -c1-300
echo "=== train rate"; timeout 600 python examples/singleview_3d_train.py --steps 8 --global-batch 16 --json $O/train_bf16.json > $O/train.log 2>&1; echo "rc $?"; tail -2 $O/train.log
echo "=== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "rc $?"; cut -c1-260 $O/bench_default.json; tail -2 $O/bench_default.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r04n/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["stage_ms"], d["value_serial"], d["latency_batch1_ms"])
print(d["roofline"]["issue_model"]); print({k:(v["achieved"],v["frac"]) for k,v in d["roofline_bf16_kernels"].items()}); print(d["roofline_voxelize"])
P
echo "=== train profile"; MF_TRAIN_MARK=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o p -- python examples/singleview_3d_train.py --steps 6 --global-batch 16 > $O/prof_train.log 2>&1; echo "rc $?"
MF_MARK=erfinv python tools/kernel_stats.py $O/prof_train > $O/train_bf16_steady_step_kernel_stats.csv; head -22 $O/train_bf16_steady_step_kernel_stats.csv | cut -c1-150; rm -rf $O/prof_train

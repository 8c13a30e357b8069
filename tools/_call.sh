cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04p; mkdir -p $O
echo "=== tests"; timeout 1200 python -m pytest tests/test_gpu_backbone2d.py -m gpu -q > $O/tests.log 2>&1; echo "rc $?"; tail -2 $O/tests.log | cut -c1-300
echo "=== train rate"; timeout 600 python examples/singleview_3d_train.py --steps 8 --global-batch 16 --json $O/train_bf16.json > $O/train.log 2>&1; echo "rc $?"; tail -2 $O/train.log
for f in "" "--channels-last"; do
echo "=== bench $f"; timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency-probe $f > $O/bench.json 2> $O/bench.err; echo "rc $?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/r04p/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["stage_ms"], d["value_serial"], d["config"]["backbone_memory_format"])
P
done
echo "=== predict profile b8"
bash tools/gpu_call.sh r04p "prof=predict_b8=WHAT=predict+REPS=5+python+tools/prof_icc.py" 2>&1 | tail -24 | cut -c1-150

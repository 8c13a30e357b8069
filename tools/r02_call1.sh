#!/bin/bash
# round-2 GPU call 1: state at HEAD -- tests, ICC stage stamps, PSPNet decoder A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02/c1_tests.log 2>&1; echo "tests rc $?"
tail -5 gpurun_out/r02/c1_tests.log
timeout 120 python tools/stamps_icc.py > gpurun_out/r02/c1_stamps.log 2>&1; tail -30 gpurun_out/r02/c1_stamps.log
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02/c1_bench_dense.json 2> gpurun_out/r02/c1_bench_dense.err
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sparse-decoder > gpurun_out/r02/c1_bench_sparsedec.json 2> gpurun_out/r02/c1_bench_sparsedec.err
python - <<'PY'
import json
for n in ("dense","sparsedec"):
    try:
        d=json.loads(open(f"gpurun_out/r02/c1_bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["value_serial"], d["stage_ms"])
    except Exception as e: print(n, "failed", e)
PY

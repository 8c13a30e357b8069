#!/bin/bash
# round-2 GPU call 2: ICC v2 (bin + tile) vs v1 -- correctness, bitwise A/B, per-kernel time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_gpu_icc.py -x -q > gpurun_out/r02/c2_tests.log 2>&1; echo "icc tests rc $?"; tail -3 gpurun_out/r02/c2_tests.log
cat > /tmp/ab.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from bench import Workload, parse
sys.argv=[sys.argv[0]]
args = parse(); wl = Workload(args, 0, torch.device("cuda", 0))
wl.q.copy_(wl.q0); wl.t.copy_(wl.t0); wl.m.zero_(); wl.v.zero_()
losses = torch.empty(100, 1).cuda()
wl.icc.refine(wl.q, wl.t, wl.m, wl.v, 100, losses=losses)
torch.cuda.synchronize()
np.save(f"/tmp/ab_{os.environ.get('MF_ICC_IMPL','2')}.npy", torch.cat([wl.q.flatten(), wl.t.flatten(), losses.flatten()]).cpu().numpy())
import time
for _ in range(3): wl._refine()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(10): wl._refine()
torch.cuda.synchronize(); print("impl", os.environ.get('MF_ICC_IMPL','2'), "refine ms", (time.perf_counter()-t0)*100)
PY
python /tmp/ab.py; MF_ICC_IMPL=1 python /tmp/ab.py
python -c "
import numpy as np
a=np.load('/tmp/ab_2.npy'); b=np.load('/tmp/ab_1.npy'); print('GPU v1 vs v2 bitwise equal over 100 iterations:', np.array_equal(a,b), float(np.abs(a-b).max()))"
bash tools/prof_k.sh r02v2 MF_ICC_IMPL=2 2>&1 | tail -6
grep -E "k_icc" gpurun_out/prof_r02v2/icc_kernel_stats.csv | head
bash tools/prof_k.sh r02v1 MF_ICC_IMPL=1 2>&1 | tail -4

#!/bin/bash
# gpurun helper: every example script runs end to end on the box
for cmd in "examples/singleview_3d_from_frame.py" "examples/singleview_3d_demo.py --batch-size 2" \
           "examples/pose_refinement_icc.py --objects 4 --iters 20" "examples/pose_refinement_icc.py --mode step --objects 3 --iters 5" \
           "examples/pose_refinement_icp.py" "examples/pose_refinement_icp.py --autograd" "examples/singleview_3d_train.py --steps 2 --global-batch 2"; do
  echo "== $cmd"; timeout 100 python $cmd 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -4
done

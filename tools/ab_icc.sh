#!/bin/bash
# A/B of ICC builds on the GPU box: tools/ab_icc.sh <tag> "<ENV=v,...>" ...  (each: 1 scene and 8 scenes, us / iteration)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  for sc in 1 8; do
    echo "[$v] scenes $sc: $(env ${v//,/ } timeout 300 python tools/time_icc_quick.py --scenes-per-gpu $sc 2>/dev/null | tail -n 1)"
  done
done

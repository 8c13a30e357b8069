cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/gpu_call.sh r05e smoke tests > gpurun_out/r05e_0.log 2>&1
python examples/singleview_3d_train.py --global-batch 16 --steps 10 --json gpurun_out/r05e/train_rate.json > gpurun_out/r05e/train_rate.log 2>&1
export MF_MARK=erfinv
bash tools/gpu_call.sh r05e "prof=train=MF_TRAIN_MARK=1+python+examples/singleview_3d_train.py+--global-batch+16+--steps+8+--json+gpurun_out/r05e/train_prof.json" > gpurun_out/r05e_1.log 2>&1
unset MF_MARK
bash tools/gpu_call.sh r05e "pmc=bf16_mfma=SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE=python+tools/time_gemm_bf16.py+16+--no-stock" > gpurun_out/r05e_2.log 2>&1

cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04b; mkdir -p $O
echo "=== repro"; hipcc --offload-arch=gfx950 tools/repro_graph_memset.hip -o /tmp/repro 2>&1 | tail -2; timeout 120 /tmp/repro > $O/repro_graph_memset.jsonl 2>&1; echo "rc $?"; cat $O/repro_graph_memset.jsonl
echo "=== graph probes (fill kernels instead of memset nodes)"
tools/probe_graph.sh $O/probes.log 6 volumetric
tools/probe_graph.sh $O/probes.log 4 full
grep -c '"done"' $O/probes.log; grep -E "rc |fault" $O/probes.log | sort | uniq -c
echo "=== tests"; timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_conv3d.py tests/test_gpu_predict_parity.py tests/test_gpu_reference_cuda_text.py -m gpu -x -q > $O/tests.log 2>&1; echo "rc $?"; tail -3 $O/tests.log
echo "=== train profile"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o p -- python examples/singleview_3d_train.py --steps 5 --global-batch 16 > $O/prof_train.log 2>&1; echo "rc $?"; tail -3 $O/prof_train.log
python tools/kernel_stats.py $O/prof_train > $O/train_bf16_kernel_stats.csv; head -40 $O/train_bf16_kernel_stats.csv | cut -c1-200; rm -rf $O/prof_train

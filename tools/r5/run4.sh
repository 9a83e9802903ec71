cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
bash tools/r5/bisect_proxy.sh 2>&1 | grep "^\[" 
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -q -m gpu -k "mha or ffn or dropout or softmax_ce or layer_norm" -s -p no:xdist > $O/run4_kern.txt 2>&1
grep -E "passed|failed|MHA MFMA|^E  " $O/run4_kern.txt | tail -20
timeout 900 python3 -m pytest tests/test_gpu_captured_step.py tests/test_gpu_io_data.py "tests/test_gpu_bench_shapes.py::test_msda_scatter_merging_consecutive_points_gives_the_same_value_gradient" -q -m gpu -s -p no:xdist > $O/run4_cap.txt 2>&1
grep -E "passed|failed|CAPTURED|EAGER vs|step 1 vs|scatter merge|^E  " $O/run4_cap.txt | tail -20

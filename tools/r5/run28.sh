cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_bench_shapes.py tests/test_gpu_fp16.py tests/test_gpu_resnet50c.py tests/test_gpu_io_data.py tests/test_gpu_bn_fold.py -q -m gpu > $O/run28_tests.txt 2>&1; grep -E "passed|failed|^E  " $O/run28_tests.txt | tail -8

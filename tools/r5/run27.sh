cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_io_data.py tests/test_gpu_train_cli.py tests/test_gpu_model.py -q -m gpu -k "segmentation_areas or val or metric or evaluat or inference or identity_contribution" > $O/run27_tests.txt 2>&1; grep -E "passed|failed|^E  " $O/run27_tests.txt | tail -8

cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu -k "non_temporal or block_shapes or memcpy or identity_contribution or sgd or optimizer or trajectory or train_step" -p no:xdist > $O/run23_kern.txt 2>&1; grep -E "passed|failed|^E  " $O/run23_kern.txt | tail -8

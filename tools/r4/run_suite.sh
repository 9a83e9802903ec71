cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4suite
timeout 2300 python3 -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r4suite/gpu_tests.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r4suite/gpu_tests.txt | tail -5
grep -E "^[0-9.]+s (call|setup)" gpurun_out/r4suite/gpu_tests.txt | head -15

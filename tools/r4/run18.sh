cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4u
timeout 600 python3 -m pytest -x -q -m gpu tests/test_gpu_kernels.py -k "consumers_loads" -s 2>&1 | grep -E "^E  |passed|failed|relative errors" | head -30
run() { name=$1; shift; timeout 300 "$@" 2> gpurun_out/r4u/$name.err | grep "^{" > gpurun_out/r4u/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4u/$name.json'));print('$name', d['value'], d['ms_per_step'])"; }

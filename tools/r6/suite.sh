# full -m gpu suite with per-test durations; usage (on the GPU box): bash tools/r5/suite.sh <tag> [extra pytest args]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-suite}; shift
mkdir -p gpurun_out/r6
T0=$(date +%s)
timeout 2300 python3 -m pytest tests -q -m gpu --durations=0 --durations-min=2.0 "$@" > gpurun_out/r6/${TAG}_tests.txt 2>&1
echo "suite wall seconds: $(( $(date +%s) - T0 ))" | tee -a gpurun_out/r6/${TAG}_tests.txt
grep -E "passed|failed|error" gpurun_out/r6/${TAG}_tests.txt | tail -5
grep -E "^[0-9.]+s (call|setup)" gpurun_out/r6/${TAG}_tests.txt | head -60

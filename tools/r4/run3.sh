cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
python3 tools/bench_conv.py wgroup mixes > gpurun_out/r4d/wgroup_mixes.txt 2>&1; cat gpurun_out/r4d/wgroup_mixes.txt
python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs --two-phase > gpurun_out/r4d/two_phase.json 2> gpurun_out/r4d/two_phase.err
python3 -c "import json;d=json.load(open('gpurun_out/r4d/two_phase.json'));print('two-phase', d['value'], d['ms_per_step'])"
python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs --two-phase --no-early-exchange > gpurun_out/r4d/two_phase_ne.json 2> gpurun_out/r4d/two_phase_ne.err
python3 -c "import json;d=json.load(open('gpurun_out/r4d/two_phase_ne.json'));print('two-phase no early', d['value'], d['ms_per_step'])"
python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/r4d/single.json 2> gpurun_out/r4d/single.err
python3 -c "import json;d=json.load(open('gpurun_out/r4d/single.json'));print('single', d['value'], d['ms_per_step'])"
for b in 16 32 64 128; do
python3 bench.py --config cfg5 --batch $b --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4d/cfg5_b$b.json 2> gpurun_out/r4d/cfg5_b$b.err
python3 -c "
import json;d=json.load(open('gpurun_out/r4d/cfg5_b$b.json'));m=d['roofline_msda']
print('cfg5 windows $b', d['value'], d['ms_per_step'], 'msda us', m['avg_launch_us'], 'hbm frac', m['frac'], 'lds frac', m['lds_frac'], 'MB', m['algorithmic_mbytes_per_launch'])"
done
python3 -m pytest -x -q -s -m gpu tests/test_gpu_model.py::test_fifty_step_training_trajectory_tracks_the_oracle 2>&1 | grep "TRAJ\|gradient-norm\|passed\|failed"

cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4c
mkdir -p $OUT
( time python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
tail -3 $OUT/bench_default.time
python3 -c "
import json;d=json.load(open('$OUT/bench_default.json'))
print('cfg2', d['value'], d['roofline']['frac'], d['roofline']['frac_net_of_event_cost'], d['roofline']['algorithmic_bytes_per_launch'], d['cpu_baseline']['value'])
for k,v in d['other_configs'].items(): print(k, v['value'], v['roofline']['frac'], v['roofline_msda']['frac'], v['roofline_msda']['lds_frac'], v['cpu_baseline'])
print(d['roofline_msda'])
"
for blocks in 512 768 1536 2048 4096; do
  EMRT_WGROUP_BLOCKS=$blocks python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/b_$blocks.json 2> $OUT/b_$blocks.err
  python3 -c "import json;d=json.load(open('$OUT/b_$blocks.json'));print('blocks $blocks', d['value'], d['ms_per_step'])"
done
for ms in 1 2 8 16; do
  EMRT_WGROUP_MIN_STEPS=$ms python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/m_$ms.json 2> $OUT/m_$ms.err
  python3 -c "import json;d=json.load(open('$OUT/m_$ms.json'));print('min_steps $ms', d['value'], d['ms_per_step'])"
done
EMRT_WGRAD8P_XCD=0 python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/x0.json 2> $OUT/x0.err
python3 -c "import json;d=json.load(open('$OUT/x0.json'));print('xcd 0', d['value'], d['ms_per_step'])"

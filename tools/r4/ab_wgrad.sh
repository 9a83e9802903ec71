# A/B: layer-by-layer backward (pair kernel) vs batched weight gradients at several batch sizes, cfg2 + cfg3; then the new GPU tests
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4b
mkdir -p $OUT
python3 -m pytest tests/test_gpu_wgrad_group.py -x -q 2>&1 | tail -15 > $OUT/test_wgrad_group.txt
cat $OUT/test_wgrad_group.txt
for b in 0 8 16 24 48 1000; do
  EMRT_WGRAD_BATCH=$b python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --dump-calls $OUT/calls_b$b.txt > $OUT/bench_b$b.json 2> $OUT/bench_b$b.err
  python3 -c "import json;d=json.load(open('$OUT/bench_b$b.json'));print('cfg2 batch $b', d['value'], d['ms_per_step'], d['final_loss'])"
done
for b in 0 24; do
  EMRT_WGRAD_BATCH=$b python3 bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg3_b$b.json 2> $OUT/bench_cfg3_b$b.err
  python3 -c "import json;d=json.load(open('$OUT/bench_cfg3_b$b.json'));print('cfg3 batch $b', d['value'], d['ms_per_step'], d['final_loss'])"
done

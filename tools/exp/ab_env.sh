# usage: bash tools/exp/ab_env.sh "<ENV_A>" "<ENV_B>" [reps]   -- alternating default-config bench runs (cfg2, 60 steps), same box
A="$1"; B="$2"; R="${3:-3}"
for i in $(seq 1 $R); do
  for V in "$A" "$B"; do
    env $V python bench.py --no-cpu-baseline --no-other-configs --steps 60 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$V', j['value'], j['ms_per_step'])"
  done
done

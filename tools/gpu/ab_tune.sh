# Alternating A/B of two bench.py argument sets on the captured step, separate processes on one box (boxes differ by +-4 %, runs on one
# box by +-0.5 %):    bash tools/gpu/ab_tune.sh "<args of arm A>" "<args of arm B>" [rounds] [config] [steps]
A=$1; B=$2; R=${3:-3}; C=${4:-cfg2}; K=${5:-30}
for rnd in $(seq 1 $R); do
  for which in A B; do
    if [ $which = A ]; then X=$A; else X=$B; fi
    timeout 600 python bench.py --config $C --steps $K --warmup 10 --no-cpu-baseline --no-kernel-events --no-clock-probe $X 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$which round $rnd [$X]:', round(d['ms_per_step'],4), 'ms/step', d['timed_regions']['ms_per_step_each'])"
  done
done

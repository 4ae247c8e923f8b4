# A/B of the weight packs' launch position on the captured cfg2 step, alternating processes on one box: in front of the stem (round 4) against
# behind the encoder, in front of their first reader (UNet.late_packs, the default).
#   bash tools/gpu/ab_late_launches.sh [rounds]
R=${1:-3}
for rnd in $(seq 1 $R); do
  for which in front late; do
    case $which in
      front) A="--net-attr late_packs=0";;
      late)  A="--net-attr late_packs=1";;
    esac
    timeout 600 python bench.py --config cfg2 --steps 30 --warmup 10 --no-cpu-baseline $A 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$which round $rnd:', round(d['ms_per_step'],4), 'ms/step')"
  done
done

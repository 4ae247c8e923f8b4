# A/B of the in-tree library against a reference build (libdct_hip_ref.so) on the whole cfg2 step, alternating processes on one box:
#   bash tools/gpu/ab_lib_step.sh [rounds] [config]
R=${1:-2}; C=${2:-cfg2}
P=$PWD/deep-co-training-for-semi-supervised-image-segmentation_amd
for rnd in $(seq 1 $R); do
  for which in ref cur; do
    if [ $which = ref ]; then export DCT_LIB_PATH=$P/libdct_hip_ref.so; else unset DCT_LIB_PATH; fi
    timeout 600 python bench.py --config $C --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$which round $rnd:', round(d['ms_per_step'],4), 'ms/step', d['roofline'].get('per_class_ms_per_step'))"
  done
done

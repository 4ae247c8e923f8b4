# A/B of the round-5 "dead zone" changes on the whole cfg2 step, alternating processes on one box.  Arms: ref = reference build
# (libdct_hip_ref.so) with the weight packs in front of the stem; kern = the in-tree
# library with that order; order = the reference build with the late launches; cur = in-tree library and defaults.
#   bash tools/gpu/ab_dead_zone.sh [rounds] [config]
R=${1:-2}; C=${2:-cfg2}
P=$PWD/deep-co-training-for-semi-supervised-image-segmentation_amd
OFF="--net-attr late_packs=0"
for rnd in $(seq 1 $R); do
  for which in ref kern order cur; do
    case $which in
      ref)   export DCT_LIB_PATH=$P/libdct_hip_ref.so; A=$OFF;;
      kern)  unset DCT_LIB_PATH; A=$OFF;;
      order) export DCT_LIB_PATH=$P/libdct_hip_ref.so; A="";;
      cur)   unset DCT_LIB_PATH; A="";;
    esac
    timeout 600 python bench.py --config $C --steps 30 --warmup 10 --no-cpu-baseline $A 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$which round $rnd:', round(d['ms_per_step'],4), 'ms/step', d['roofline'].get('per_class_ms_per_step'))"
  done
done

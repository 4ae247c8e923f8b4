P=$PWD/deep-co-training-for-semi-supervised-image-segmentation_amd
for C in cfg4 cfg5; do
for rnd in 1 2 3; do
  for which in w4 w8; do
    if [ $which = w8 ]; then export DCT_LIB_PATH=$P/libdct_hip_w8.so; else unset DCT_LIB_PATH; fi
    timeout 600 python bench.py --config $C --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-events --no-clock-probe 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$C $which round $rnd:', round(d['ms_per_step'],4), 'ms/step', d['losses_last_step'])"
  done
done
done

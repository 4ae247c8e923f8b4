O=gpurun_out/r3o
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "kernels_gpu or unet_gpu" > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
grep -E "passed|failed|^FAILED" $O/tests.log | tail -5
timeout 900 python tools/bench_conv.py --batch 16 --what wgrad --ab-knob 33=0,1 > $O/bench_wgrad_shift.txt 2>&1
grep -E "^---|TOTAL" $O/bench_wgrad_shift.txt
for v in 0 1 0 1; do timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --tune 33=$v > $O/b.json 2>> $O/bench.err; python -c "
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('shift=$v', round(d['ms_per_step'],3), d['roofline']['per_class_ms_per_step'])"; done

O=gpurun_out/r3g
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "kernels_gpu or unet_gpu" > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
timeout 900 python tools/bench_conv.py --batch 16 --what fwd,dgrad --only dec1b,dec2a,dec2b,dec3a,dec3b,enc1a,enc1b --ab-knob 31=0,1 > $O/bench_conv_persist.txt 2>&1
timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --tune 31=0 > $O/bench_cfg2_p0.json 2> $O/bench_cfg2.err
timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --tune 31=1 > $O/bench_cfg2_p1.json 2>> $O/bench_cfg2.err
timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --tune 31=0 > $O/bench_cfg2_p0b.json 2>> $O/bench_cfg2.err
timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --tune 31=1 > $O/bench_cfg2_p1b.json 2>> $O/bench_cfg2.err
grep -E "passed|failed" $O/tests.log | tail -2
grep -E "^---|TOTAL" $O/bench_conv_persist.txt
for f in p0 p1 p0b p1b; do python -c "
import json
d=json.loads(open('$O/bench_cfg2_$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['roofline']['per_class_ms_per_step'])"; done

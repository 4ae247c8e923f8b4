O=gpurun_out/r3i
mkdir -p $O
export DCT_PARITY_REPORT=1
timeout 2400 python -m pytest tests -m gpu -q -s > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
timeout 600 python bench.py --config cfg4 --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench.err
timeout 600 python bench.py --config cfg5 --dtype f16 --steps 20 --warmup 10 --no-cpu-baseline > $O/bench_cfg5.json 2>> $O/bench.err
timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_cfg2.json 2>> $O/bench.err
grep -E "passed|failed|^FAILED|^ERROR" $O/tests.log | tail -8
grep -E "epoch" $O/tests.log | tail -12
for f in cfg4 cfg5 cfg2; do python -c "
import json
d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'])"; done

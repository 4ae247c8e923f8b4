set -x
O=gpurun_out/r3d
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "kernels_gpu or unet_gpu" > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
timeout 900 python tools/bench_conv.py --batch 16 --what fwd,dgrad > $O/bench_conv.txt 2>&1
timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err
grep -E "passed|failed" $O/tests.log | tail -2

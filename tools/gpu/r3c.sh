set -x
O=gpurun_out/r3c2
mkdir -p $O
export DCT_PARITY_REPORT=1
timeout 900 python -m pytest tests -m gpu -q -s -k "enet_configs_full_size_vs_oracle and f32 or resync or shared_halo or wgrad" > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
timeout 900 python tools/bench_conv.py --batch 16 --what fwd,dgrad --ab-knob 31=0,1 > $O/bench_conv_stagger_igemm.txt 2>&1
timeout 900 python tools/bench_conv.py --batch 16 --what wgrad --ab-knob 32=0,1 > $O/bench_conv_stagger_wgrad.txt 2>&1
timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_cfg2_base.json 2> $O/bench_cfg2.err
timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --tune 31=1 > $O/bench_cfg2_st31.json 2>> $O/bench_cfg2.err
timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --tune 31=1 --tune 32=1 > $O/bench_cfg2_st31_32.json 2>> $O/bench_cfg2.err
timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_cfg2_base2.json 2>> $O/bench_cfg2.err
grep -E "passed|failed" $O/tests.log | tail -2

# Round-5 evidence (profiles/r05_*): run through gpurun from the repo root:  bash tools/gpu/r5_profiles.sh
set -u
O=gpurun_out/r05p
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-kernel-events --no-clock-probe"
run() { name=$1; shift; timeout 600 rocprofv3 "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; }
for c in cfg2 cfg3; do
  run k_$c  --kernel-trace --stats -d $O/k_$c -o k --output-format csv -- python3 bench.py --config $c --steps 10 --warmup 3 $B --single-stream --no-graph
  run kd_$c --kernel-trace --stats -d $O/kd_$c -o k --output-format csv -- python3 bench.py --config $c --steps 20 --warmup 5 $B
  run f_$c  --pmc FETCH_SIZE --kernel-trace -d $O/f_$c -o f --output-format csv -- python3 bench.py --config $c --steps 4 --warmup 2 $B --single-stream --no-graph
  run w_$c  --pmc WRITE_SIZE --kernel-trace -d $O/w_$c -o w --output-format csv -- python3 bench.py --config $c --steps 4 --warmup 2 $B --single-stream --no-graph
  run m_$c  --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/m_$c -o m --output-format csv -- python3 bench.py --config $c --steps 5 --warmup 3 $B --single-stream --no-graph
  python3 tools/rocprof_summary.py $(find $O/k_$c -name "*kernel_trace.csv" | head -1) > $O/${c}_single_stream_kernel_stats.txt
  python3 tools/rocprof_summary.py $(find $O/kd_$c -name "*kernel_trace.csv" | head -1) > $O/${c}_default_command_kernel_stats.txt
  python3 tools/pmc_traffic.py "$(find $O/f_$c -name "*counter_collection.csv" | head -1)" "$(find $O/w_$c -name "*counter_collection.csv" | head -1)" $c 24 > $O/${c}_pmc_traffic.json
  python3 tools/pmc_mfma.py "$(find $O/m_$c -name "*counter_collection.csv" | head -1)" > $O/${c}_pmc_mfma_busy.txt
done
timeout 600 python tools/bench_conv.py --batch 16 --plan > $O/bench_conv_per_layer.txt 2>&1
timeout 600 python bench.py --steps 30 --warmup 10 > $O/cfg2_bench.json 2> $O/bench.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/cfg2_driver_command_bench.json 2>> $O/bench.err
timeout 600 python bench.py --steps 30 --warmup 10 --data iid --no-cpu-baseline > $O/cfg2_iid_inputs_bench.json 2>> $O/bench.err
timeout 600 python bench.py --config cfg3 --steps 30 --warmup 10 --no-cpu-baseline > $O/cfg3_bench.json 2>> $O/bench.err
timeout 600 python bench.py --config cfg4 --steps 30 --warmup 10 --no-cpu-baseline > $O/cfg4_bench.json 2>> $O/bench.err
timeout 600 python bench.py --config cfg4u --steps 30 --warmup 10 --no-cpu-baseline > $O/cfg4_unet_200_bench.json 2>> $O/bench.err
timeout 600 python bench.py --config cfg5 --dtype f16 --steps 30 --warmup 10 --no-cpu-baseline > $O/cfg5_f16_bench.json 2>> $O/bench.err
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.csv" -size +200k -delete
ls $O

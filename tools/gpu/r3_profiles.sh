# Round-3 evidence (profiles/r03_*): run through gpurun from the repo root:  bash tools/gpu/r3_profiles.sh
set -u
O=gpurun_out/r03p
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-kernel-events"
run() { name=$1; shift; timeout 600 rocprofv3 "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; }
for c in cfg2 cfg3; do
  run k_$c  --kernel-trace --stats -d $O/k_$c -o k --output-format csv -- python3 bench.py --config $c --steps 10 --warmup 3 $B --single-stream --no-graph
  run kd_$c --kernel-trace --stats -d $O/kd_$c -o k --output-format csv -- python3 bench.py --config $c --steps 20 --warmup 5 $B
  run f_$c  --pmc FETCH_SIZE --kernel-trace -d $O/f_$c -o f --output-format csv -- python3 bench.py --config $c --steps 4 --warmup 2 $B --single-stream --no-graph
  run w_$c  --pmc WRITE_SIZE --kernel-trace -d $O/w_$c -o w --output-format csv -- python3 bench.py --config $c --steps 4 --warmup 2 $B --single-stream --no-graph
  run m_$c  --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/m_$c -o m --output-format csv -- python3 bench.py --config $c --steps 5 --warmup 3 $B --single-stream --no-graph
  python3 tools/rocprof_summary.py $(find $O/k_$c -name "*kernel_trace.csv" | head -1) > $O/${c}_single_stream_kernel_stats.txt
  python3 tools/rocprof_summary.py $(find $O/kd_$c -name "*kernel_trace.csv" | head -1) > $O/${c}_default_command_kernel_stats.txt
  python3 tools/pmc_traffic.py "$(find $O/f_$c -name "*counter_collection.csv" | head -1)" "$(find $O/w_$c -name "*counter_collection.csv" | head -1)" $c 16 > $O/${c}_pmc_traffic.json
  python3 tools/pmc_mfma.py "$(find $O/m_$c -name "*counter_collection.csv" | head -1)" "$(find $O/m_$c -name "*kernel_trace.csv" | head -1)" > $O/${c}_pmc_mfma_busy.txt
done
# per-layer table
timeout 600 python tools/bench_conv.py --batch 16 > $O/bench_conv_per_layer.txt 2>&1
# operand statistics: the step after 300 more Adam steps, and the in-kernel clock on dense / half-zero / 90 %-zero activations
timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/cfg2_bench.json 2> $O/bench.err
timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --train-steps 300 > $O/cfg2_after_300_steps.json 2>> $O/bench.err
timeout 600 python bench.py --config cfg3 --steps 30 --warmup 10 --no-cpu-baseline > $O/cfg3_bench.json 2>> $O/bench.err
# (the in-kernel clock of profiles/r03_in_kernel_clock.txt came from a stamped build of the 32x32x16 shared-halo kernel, removed in round 4;
#  round 4's stamps: tools/gpu/i4_stamps.py on the -DDCT_I4_ABLATE build)
# Enet configurations: kernel statistics of the default command, bench lines, the step program's timeline
run kd_cfg4 --kernel-trace --stats -d $O/kd_cfg4 -o k --output-format csv -- python3 bench.py --config cfg4 --steps 20 --warmup 5 $B
python3 tools/rocprof_summary.py $(find $O/kd_cfg4 -name "*kernel_trace.csv" | head -1) > $O/cfg4_default_command_kernel_stats.txt
timeout 600 python bench.py --config cfg4 --steps 30 --warmup 10 --no-cpu-baseline > $O/cfg4_bench.json 2>> $O/bench.err
timeout 600 python bench.py --config cfg5 --steps 30 --warmup 10 --no-cpu-baseline > $O/cfg5_bench.json 2>> $O/bench.err
timeout 300 python tools/probe_step_program.py cfg4 > $O/cfg4_step_program_timeline.txt 2>&1
timeout 300 python tools/probe_step_program.py cfg5 > $O/cfg5_step_program_timeline.txt 2>&1
# fabric traffic of the Enet kernel families (the derived FETCH_SIZE metric crashes rocprofv3 on this workload; its base counter does not)
RX="enet_one|enet_grp"
run f_cfg4  --pmc TCC_EA0_RDREQ_sum --kernel-trace --kernel-include-regex "$RX" -d $O/f_cfg4 -o f --output-format csv -- python3 bench.py --config cfg4 --steps 4 --warmup 2 $B --single-stream --no-graph
run w_cfg4  --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "$RX" -d $O/w_cfg4 -o w --output-format csv -- python3 bench.py --config cfg4 --steps 4 --warmup 2 $B --single-stream --no-graph
python3 tools/pmc_traffic_enet.py "$(find $O/f_cfg4 -name "*counter_collection.csv" | head -1)" "$(find $O/w_cfg4 -name "*counter_collection.csv" | head -1)" > $O/cfg4_pmc_traffic.json
# PMC study of the three dominant kernels (one layer each)
bash tools/gpu/pmc_layer.sh $O/pmc_igemm3m_dec2b "" dec2b fwd
bash tools/gpu/pmc_layer.sh $O/pmc_igemm3p_dec4b "" dec4b fwd
bash tools/gpu/pmc_layer.sh $O/pmc_wgrad3_dec2b "" dec2b wgrad
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.csv" -size +100k -delete
ls $O

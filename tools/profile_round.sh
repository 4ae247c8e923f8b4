#!/bin/bash
# Round profiles on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh r2p
# kernel statistics (rocprofv3 --kernel-trace --stats) of the default bench command and of the single-stream command whose
# per-kernel durations the bench's roofline leg reports; HBM traffic (two --pmc passes: FETCH_SIZE / WRITE_SIZE cannot share one)
set -u
OUT=gpurun_out/${1:-r2p}
mkdir -p $OUT
B="--no-cpu-baseline --no-kernel-events"
run() { name=$1; shift; timeout 600 rocprofv3 "$@" > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
# cfg2
run k_cfg2  --kernel-trace --stats -d $OUT/k_cfg2 -o k --output-format csv -- python3 bench.py --steps 10 --warmup 3 $B --single-stream --no-graph
run kd_cfg2 --kernel-trace --stats -d $OUT/kd_cfg2 -o k --output-format csv -- python3 bench.py --steps 20 --warmup 5 $B
run f_cfg2  --pmc FETCH_SIZE --kernel-trace -d $OUT/f_cfg2 -o f --output-format csv -- python3 bench.py --steps 4 --warmup 2 $B --single-stream --no-graph
run w_cfg2  --pmc WRITE_SIZE --kernel-trace -d $OUT/w_cfg2 -o w --output-format csv -- python3 bench.py --steps 4 --warmup 2 $B --single-stream --no-graph
# cfg4 (Enet)
run k_cfg4  --kernel-trace --stats -d $OUT/k_cfg4 -o k --output-format csv -- python3 bench.py --config cfg4 --steps 10 --warmup 3 $B --single-stream --no-graph
run kd_cfg4 --kernel-trace --stats -d $OUT/kd_cfg4 -o k --output-format csv -- python3 bench.py --config cfg4 --steps 20 --warmup 5 $B
# (the derived FETCH_SIZE metric crashes rocprofv3 on the Enet step; its base counter, restricted to the Enet kernel families, does not)
RX="enet_mconv|enet_mwgrad|enet_reduce|enet_bn_|enet_wgrad_reduce|enet_conv_kernel"
run f_cfg4  --pmc TCC_EA0_RDREQ_sum --kernel-trace --kernel-include-regex "$RX" -d $OUT/f_cfg4 -o f --output-format csv -- python3 bench.py --config cfg4 --steps 4 --warmup 2 $B --single-stream --no-graph
run w_cfg4  --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "$RX" -d $OUT/w_cfg4 -o w --output-format csv -- python3 bench.py --config cfg4 --steps 4 --warmup 2 $B --single-stream --no-graph
find $OUT -name "*.csv" | head -40
for c in cfg2 cfg4; do
  python3 tools/rocprof_summary.py $(find $OUT/k_$c -name "*kernel_trace.csv" | head -1) > $OUT/${c}_single_stream_kernel_stats.txt
  python3 tools/rocprof_summary.py $(find $OUT/kd_$c -name "*kernel_trace.csv" | head -1) > $OUT/${c}_default_command_kernel_stats.txt
done
python3 tools/pmc_traffic.py "$(find $OUT/f_cfg2 -name "*counter_collection.csv" | head -1)" "$(find $OUT/w_cfg2 -name "*counter_collection.csv" | head -1)" cfg2 16 > $OUT/cfg2_pmc_traffic.json
python3 tools/pmc_traffic_enet.py "$(find $OUT/f_cfg4 -name "*counter_collection.csv" | head -1)" "$(find $OUT/w_cfg4 -name "*counter_collection.csv" | head -1)" > $OUT/cfg4_pmc_traffic.json
# raw csvs are large: keep only the summaries
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
ls -la $OUT

set -x
mkdir -p gpurun_out/r3b
export DCT_PARITY_REPORT=1
timeout 1800 python -m pytest tests -m gpu -q -s -k "full_size_vs_oracle or resync or enet_vs_oracle or test_step_gpu or ddp or pack_weights or kernels" > gpurun_out/r3b/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3b/tests.log
timeout 600 python tools/bench_conv.py --batch 16 > gpurun_out/r3b/bench_conv.txt 2>&1
timeout 600 python bench.py --steps 30 --warmup 10 > gpurun_out/r3b/bench_cfg2.json 2> gpurun_out/r3b/bench_cfg2.err
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/r3b/k_cfg2 -o k --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --single-stream --no-graph > gpurun_out/r3b/prof_run.log 2>&1
python3 tools/rocprof_summary.py $(find gpurun_out/r3b/k_cfg2 -name "*kernel_trace.csv" | head -1) > gpurun_out/r3b/cfg2_single_stream_kernel_stats.txt
find gpurun_out/r3b -name "*kernel_trace.csv" -delete
tail -3 gpurun_out/r3b/tests.log

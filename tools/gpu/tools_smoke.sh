# Runs every probe / bench tool once with small arguments and records which still work against the current ABI
# (the index in tools/README.md is written from this):  bash tools/gpu/tools_smoke.sh
O=gpurun_out/tools_smoke
mkdir -p $O
t() { name=$1; shift; timeout 240 "$@" > $O/$name.log 2>&1; echo "$name rc=$?" | tee -a $O/summary.txt; }
t ab_step            python tools/ab_step.py --knob 11 --rounds 1 --steps 3
t bench_conv         python tools/bench_conv.py --batch 4 --reps 3 --only dec3b,enc2T --plan
t bench_enet_layers  python tools/bench_enet_layers.py
t bench_meters       python tools/bench_meters.py
t bench_pointwise    python tools/bench_pointwise.py
t debug_conv_case    python tools/debug_conv_case.py
t debug_enet_blocks  python tools/debug_enet_blocks.py
t debug_fuse_bn      python tools/debug_fuse_bn.py
t debug_unet_layers  python tools/debug_unet_layers.py
t host_time          python tools/host_time.py
t probe_graph_chains python tools/probe_graph_chains.py
t probe_step_program python tools/probe_step_program.py cfg4
t probe_stream_pairs python tools/probe_stream_pairs.py
t probe_uninit       python tools/probe_uninit.py
t run_steps          python tools/run_steps.py cfg4 4
t pmc_layer          python tools/pmc_layer.py --layer dec3b --n 2
t probe_packed_fp32  python tools/probe_packed_fp32/probe.py
t probe_buffer_lds   bash -c '/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/probe_buffer_lds/probe.hip -o tools/probe_buffer_lds/probe.bin && tools/probe_buffer_lds/probe.bin'
t isa_loop_mix       bash -c 'cd deep-co-training-for-semi-supervised-image-segmentation_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S wgrad.hip -o /tmp/wgrad_smoke.s 2>/dev/null; cd ../.. && python tools/isa_loop_mix.py /tmp/wgrad_smoke.s wgrad3_kernelILi64ELi64ELi4ELb0ELi2ELb1ELb1'
t pmc_kernel_mix     bash -c 'export TMPDIR=/tmp; rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d '$O'/pm -o p --output-format csv -- python3 tools/bench_conv.py --batch 4 --reps 2 --only dec3b > /dev/null 2>&1; python tools/pmc_kernel_mix.py $(find '$O'/pm -name "*counter_collection.csv" | head -1); python tools/trace_alone.py $(find '$O'/pm -name "*kernel_trace.csv" | head -1); python tools/trace_runs.py $(find '$O'/pm -name "*kernel_trace.csv" | head -1) 2 | tail -5'
t power_clock        bash tools/gpu/power_clock.sh cfg2 60
t bench_unpool       python tools/bench_unpool.py --reps 3 --batch 4
t bench_grouped      python tools/bench_grouped.py --reps 3
t probe_store_bw     bash -c '/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/probe_store_bw/probe.hip -o tools/probe_store_bw/probe.bin && tools/probe_store_bw/probe.bin | tail -5'
for f in $O/*.log; do echo "==> $f"; tail -n 3 "$f"; done | grep -E "==>|Error|error|Traceback" | tail -n 60
cat $O/summary.txt

# Runs every probe / bench tool once with small arguments and records which still work against the current ABI
# (the index in tools/README.md is written from this):  bash tools/gpu/tools_smoke.sh
O=gpurun_out/tools_smoke
mkdir -p $O
t() { name=$1; shift; timeout 240 "$@" > $O/$name.log 2>&1; echo "$name rc=$?" | tee -a $O/summary.txt; }
t ab_step            python tools/ab_step.py --knob 11 --rounds 1 --steps 3
t bench_conv         python tools/bench_conv.py --batch 4 --reps 3 --only dec3b,enc2T
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
for f in $O/*.log; do echo "==> $f"; tail -n 3 "$f"; done | grep -E "==>|Error|error|Traceback" | tail -n 60
cat $O/summary.txt

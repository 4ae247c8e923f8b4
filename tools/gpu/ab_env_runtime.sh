# Sweep of the HIP runtime's environment switches around graph launches on the captured cfg2 step (one process per setting, alternating with the default).
#   bash tools/gpu/ab_env_runtime.sh
# (ROC_SYSTEM_SCOPE_SIGNAL=0 and DEBUG_HIP_DYNAMIC_QUEUES=1 HANG the captured step -- 600 s each until the timeout -- and are not in the list.)
run() { timeout 600 python bench.py --config cfg2 --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1:', round(d['ms_per_step'],4))
except Exception as e:
    print('$1: FAILED')"; }
for rnd in 1 2; do
  run default
  for kv in DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=2 DEBUG_HIP_FORCE_GRAPH_QUEUES=4 DEBUG_HIP_GRAPH_BATCH_SIZE=16 DEBUG_HIP_GRAPH_BATCH_SIZE=512 \
            GPU_STREAMOPS_CP_WAIT=1 GPU_STREAMOPS_CP_WAIT=0 DEBUG_HIP_DYNAMIC_QUEUES=0 ROC_ACTIVE_WAIT_TIMEOUT=0 DEBUG_HIP_KERNARG_COPY_OPT=0; do
    export $kv; run $kv; unset ${kv%%=*}
  done
done

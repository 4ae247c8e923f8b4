# Board power and shader clock sampled (rocm-smi, 4 Hz) while the bench step runs: is the step power-managed?
#   bash tools/gpu/power_clock.sh [config] [steps]      -> gpurun_out/power_clock_<config>.txt
C=${1:-cfg2}; K=${2:-600}
mkdir -p gpurun_out
OUT=gpurun_out/power_clock_$C.txt
: > $OUT.raw
( while true; do
    echo "t $(date +%s.%N)" >> $OUT.raw
    rocm-smi -d 0 --showpower --showclocks --showmaxpower --showuse --showtemp 2>/dev/null | grep -E "Power|sclk|fclk|mclk|busy|junction|hotspot|Temperature" >> $OUT.raw
    sleep 0.25
  done ) &
SAMPLER=$!
sleep 3                                   # idle samples first
echo "bench-start $(date +%s.%N)" >> $OUT.raw
timeout 900 python bench.py --config $C --steps $K --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/power_clock_$C.json
echo "bench-end $(date +%s.%N)" >> $OUT.raw
sleep 2
kill $SAMPLER
python tools/gpu/power_clock_report.py $OUT.raw gpurun_out/power_clock_$C.json | tee $OUT

# third pass: around the new defaults (packed fill 50, no un-pooling on load)
O=gpurun_out/knob_sweep5; mkdir -p $O; : > $O/third.txt
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-events --no-clock-probe "$@" 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))"; }
for rnd in 1 2; do
  echo "default $(run)" >> $O/third.txt
  for kv in "19=800" "19=1600" "20=90" "20=85" "9=85" "9=80" "23=100" "23=260" "16=600" "14=192" "1002=64" "1002=128" "15=640" "39=0"; do
    args=""; for k in $kv; do args="$args --tune $k"; done
    echo "$kv $(run $args)" >> $O/third.txt
  done
  echo "default $(run)" >> $O/third.txt
done
cat $O/third.txt

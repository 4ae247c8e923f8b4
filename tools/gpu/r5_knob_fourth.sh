# fourth pass: the packed-rows split threshold (knob 23) is a narrow valley (100: +3.3 %, 260: +2.5 %): finer steps; and the split target of the per-tap kernel
O=gpurun_out/knob_sweep5; mkdir -p $O; : > $O/fourth.txt
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-events --no-clock-probe "$@" 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))"; }
for rnd in 1 2; do
  echo "default $(run)" >> $O/fourth.txt
  for kv in "23=128" "23=129" "23=145" "23=175" "23=192" "23=208" "23=224" "23=240" "24=45" "24=35"; do
    args=""; for k in $kv; do args="$args --tune $k"; done
    echo "$kv $(run $args)" >> $O/fourth.txt
  done
  echo "default $(run)" >> $O/fourth.txt
done
cat $O/fourth.txt

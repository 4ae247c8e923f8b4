# second pass of the round-5 planner sweep: the candidates that beat the default in the first pass, repeated and combined
O=gpurun_out/knob_sweep5; mkdir -p $O; : > $O/refine.txt
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-events --no-clock-probe "$@" 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))"; }
for rnd in 1 2; do
  echo "default $(run)" >> $O/refine.txt
  for kv in "24=60" "24=50" "24=40" "19=800" "20=90" "9=85" "24=60 20=90" "24=60 9=85" "24=60 20=90 9=85" "24=60 19=800"; do
    args=""; for k in $kv; do args="$args --tune $k"; done
    echo "$kv $(run $args)" >> $O/refine.txt
  done
  echo "default $(run)" >> $O/refine.txt
  echo "unpool off $(run --net-attr unpool_on_load=0)" >> $O/refine.txt
  echo "24=60 unpool off $(run --tune 24=60 --net-attr unpool_on_load=0)" >> $O/refine.txt
done
cat $O/refine.txt

# cfg2 step under one planner knob at a time (defaults between the candidates), on the round-5 build (sub-step skipping, XCD order,
# un-pooling on load at level 1, streaming stores): did the optima move?
O=gpurun_out/knob_sweep5; mkdir -p $O; : > $O/sweep.txt
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-events --no-clock-probe "$@" 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))"; }
echo "default $(run)" >> $O/sweep.txt
for kv in 15=512 15=1024 14=192 14=384 default 16=300 16=600 23=100 23=260 24=60 24=90 default 9=50 9=85 20=60 20=90 19=200 19=800 default 1002=64 1002=160 39=0 39=2; do
  if [ $kv = default ]; then echo "default $(run)" >> $O/sweep.txt; else echo "$kv $(run --tune $kv)" >> $O/sweep.txt; fi
done
echo "net unpool_max_level=0 $(run --net-attr unpool_on_load=0)" >> $O/sweep.txt
echo "net unpool_max_level=3 $(run --net-attr unpool_max_level=3)" >> $O/sweep.txt
echo "attr batch_lab_unlab=0 $(run --attr batch_lab_unlab=0)" >> $O/sweep.txt
echo "default $(run)" >> $O/sweep.txt
cat $O/sweep.txt

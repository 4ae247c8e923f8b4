# cfg2 step under one planner knob at a time (defaults between the candidates), after the round-4 instruction diet: did the optima move?
O=gpurun_out/knob_sweep4; mkdir -p $O; : > $O/sweep.txt
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-events "$@" 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))"; }
echo "default $(run)" >> $O/sweep.txt
for kv in 15=512 15=1024 15=1536 14=256 14=576 14=768 default 16=300 16=600 23=60 23=160 23=260 24=60 default 9=50 9=85 20=60 20=90 19=200 19=800; do
  if [ $kv = default ]; then echo "default $(run)" >> $O/sweep.txt; else echo "$kv $(run --tune $kv)" >> $O/sweep.txt; fi
done
echo "default $(run)" >> $O/sweep.txt
cat $O/sweep.txt

# cfg2 step under one knob at a time (defaults between the candidates): did this round's kernel changes move the planner optima?
O=gpurun_out/knob_sweep; mkdir -p $O; : > $O/sweep.txt
run() { python bench.py --steps 150 --warmup 20 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
echo "default $(run)" >> $O/sweep.txt
for kv in 15=512 15=1024 15=1536 14=256 14=512 14=768 16=300 16=600 16=900 23=60 23=160 24=60 24=90 9=50 9=85 20=60 20=90; do
  echo "$kv $(run --tune $kv)" >> $O/sweep.txt
done
echo "default $(run)" >> $O/sweep.txt
cat $O/sweep.txt

# cfg4 step under one Enet knob at a time
O=gpurun_out/knob_sweep; mkdir -p $O; : > $O/sweep_cfg4.txt
run() { python bench.py --config cfg4 --steps 100 --warmup 20 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
echo "default $(run)" >> $O/sweep_cfg4.txt
for kv in 28=1024 28=4096 29=8 22=512 21=0; do
  echo "$kv $(run --tune $kv)" >> $O/sweep_cfg4.txt
done
for t in 1280 5120; do echo "DCT_ENET_STATS_TILES=$t $(DCT_ENET_STATS_TILES=$t run)" >> $O/sweep_cfg4.txt; done
echo "default $(run)" >> $O/sweep_cfg4.txt
cat $O/sweep_cfg4.txt

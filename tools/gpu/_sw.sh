run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-events "$@" 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))"; }
for r in 1 2; do
echo "default(96) $(run)"
echo "1002=64 $(run --tune 1002=64)"
echo "1002=120 $(run --tune 1002=120)"
echo "16=300 $(run --tune 16=300)"
echo "16=700 $(run --tune 16=700)"
echo "1003=24 $(run --tune 1003=24)"
done

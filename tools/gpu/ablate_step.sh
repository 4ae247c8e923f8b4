# What each kernel family costs the CAPTURED cfg2 step: bench.py with the family's launches skipped (dct_tune_set(1100, mask): garbage
# results, timing only) against the full step, alternating on one box.  cost = full - without; "serialized" = the family's kernel time on one stream
# (profiles/r05_cfg2_single_stream_kernel_stats.txt).   bash tools/gpu/ablate_step.sh [config]
C=${1:-cfg2}
run() { python bench.py --config $C --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events --no-clock-probe --allow-nan "$@" 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))"; }
for rnd in 1 2; do
  echo "full            $(run)"
  echo "-igemm2         $(run --tune 1100=1)"
  echo "-igemm3m        $(run --tune 1100=2)"
  echo "-igemm3p        $(run --tune 1100=4)"
  echo "-wgrad2         $(run --tune 1100=8)"
  echo "-wgrad3         $(run --tune 1100=16)"
  echo "-folds          $(run --tune 1100=32)"
  echo "-adam           $(run --tune 1100=64)"
  echo "-pointwise      $(run --tune 1100=128)"
  echo "-all wgrad+folds $(run --tune 1100=56)"
  echo "-all igemm      $(run --tune 1100=7)"
  echo "-everything conv $(run --tune 1100=63)"
  echo "full            $(run)"
done

# A/B of two whole TREES on a step, alternating processes on one box: _ab_ref/ (an export of an earlier commit with its own library:
#   rm -rf _ab_ref && mkdir _ab_ref && git archive <commit> bench.py dct_amd.py tests/helpers.py __graft_entry__.py BASELINE.json deep-co-training-for-semi-supervised-image-segmentation_amd oracle include profiles/r05_cfg2_pmc_traffic.json profiles/r05_cfg2_pmc_mfma_busy.txt | tar -x -C _ab_ref
#   make -C _ab_ref/deep-co-training-for-semi-supervised-image-segmentation_amd/csrc -j8
# ) against the working tree -- for changes that span the C ABI and the host code, which a library swap (ab_lib_step.sh) cannot compare.
#   bash tools/gpu/ab_tree_step.sh [rounds] [config]
R=${1:-3}; C=${2:-cfg2}
ROOT=$PWD
for rnd in $(seq 1 $R); do
  for which in ref cur; do
    if [ $which = ref ]; then cd $ROOT/_ab_ref; else cd $ROOT; fi
    timeout 600 python bench.py --config $C --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$which round $rnd:', round(d['ms_per_step'],4), 'ms/step', d['roofline'].get('per_class_ms_per_step'))"
  done
done
cd $ROOT

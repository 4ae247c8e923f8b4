set -x
O=gpurun_out/r3e
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
P=deep-co-training-for-semi-supervised-image-segmentation_amd
for L in dec2b dec2a; do
  bash tools/gpu/pmc_layer.sh $O/new_$L "" $L fwd
  bash tools/gpu/pmc_layer.sh $O/old_$L $P/libdct_hip_old.so $L fwd
done
bash tools/gpu/pmc_layer.sh $O/wgrad_dec2b "" dec2b wgrad
bash tools/gpu/pmc_layer.sh $O/p_dec4b "" dec4b fwd
cat $O/new_dec2b/report.txt | head -40

# A/B of the in-tree library against a reference build (libdct_hip_ref.so) and of one knob, per-layer conv bench, two rounds:
#   bash tools/gpu/ab_lib.sh <outdir> <knob=v0,v1,...> [layers]
O=${1:-gpurun_out/ab}; K=${2:-33=0,1,2}; L=${3:-dec1b,dec2a,dec2b,dec3a,dec3b,enc1a,enc1b}
mkdir -p $O
P=$PWD/deep-co-training-for-semi-supervised-image-segmentation_amd
for rnd in 0 1; do
  if [ -f $P/libdct_hip_ref.so ]; then
    echo "=== ref round $rnd" >> $O/ab.txt
    DCT_LIB_PATH=$P/libdct_hip_ref.so timeout 300 python tools/bench_conv.py --batch 16 --what fwd,dgrad --only $L 2>&1 | grep -E "^(dec|enc|cen|TOTAL)" >> $O/ab.txt
  fi
  echo "=== cur round $rnd" >> $O/ab.txt
  timeout 600 python tools/bench_conv.py --batch 16 --what fwd,dgrad --only $L --ab-knob $K --rounds 1 2>&1 | grep -E "^(dec|enc|cen|TOTAL|---)" >> $O/ab.txt
done
grep -E "===|TOTAL|---" $O/ab.txt

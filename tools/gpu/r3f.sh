O=gpurun_out/r3f
mkdir -p $O
P=$PWD/deep-co-training-for-semi-supervised-image-segmentation_amd
for rnd in 0 1; do
for v in old u1w1 u1w0 u3w0 cur; do
  if [ $v = cur ]; then L=$P/libdct_hip.so; else L=$P/libdct_hip_$v.so; fi
  echo "=== $v round $rnd" >> $O/ab.txt
  DCT_LIB_PATH=$L timeout 300 python tools/bench_conv.py --batch 16 --what fwd,dgrad --only dec1b,dec2a,dec2b,dec3a,dec3b,enc1a,enc1b 2>&1 | grep -E "^(dec|enc|TOTAL)" >> $O/ab.txt
done
done
cat $O/ab.txt | grep -E "===|TOTAL"

O=gpurun_out/r3h
mkdir -p $O
P=$PWD/deep-co-training-for-semi-supervised-image-segmentation_amd
for rnd in 0 1; do
  echo "=== ref round $rnd" >> $O/ab.txt
  DCT_LIB_PATH=$P/libdct_hip_ref.so timeout 300 python tools/bench_conv.py --batch 16 --what fwd,dgrad --only dec1b,dec2a,dec2b,dec3a,dec3b,enc1a,enc1b 2>&1 | grep -E "^(dec|enc|TOTAL)" >> $O/ab.txt
  echo "=== cur round $rnd" >> $O/ab.txt
  timeout 300 python tools/bench_conv.py --batch 16 --what fwd,dgrad --only dec1b,dec2a,dec2b,dec3a,dec3b,enc1a,enc1b --ab-knob 31=0,1 --rounds 1 2>&1 | grep -E "^(dec|enc|TOTAL|---)" >> $O/ab.txt
done
grep -E "===|TOTAL|---" $O/ab.txt

set -x
mkdir -p gpurun_out/r3a
export DCT_PARITY_REPORT=1
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "full_size_vs_oracle or resync or enet_vs_oracle or test_step_gpu or ddp" > gpurun_out/r3a/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a/tests.log
timeout 600 python tools/bench_conv.py --batch 16 > gpurun_out/r3a/bench_conv.txt 2>&1
for L in dec1b dec2a dec2b dec3b; do
  DCT_LIB_PATH=$PWD/deep-co-training-for-semi-supervised-image-segmentation_amd/libdct_hip_stamps.so timeout 120 python tools/stamps_igemm3.py --layer $L >> gpurun_out/r3a/stamps.txt 2>&1
done
tail -5 gpurun_out/r3a/tests.log

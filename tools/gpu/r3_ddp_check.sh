# DDP averaging (SUM + 1/world in Adam / dct_flat_scale): kernel test, two-rank tests, one-rank RCCL step tests
O=gpurun_out/ddp_check; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_ddp_gpu.py tests/test_step_gpu.py -q -m gpu -k "flat_scale or ddp or rank or exchange" > $O/tests_full.txt 2>&1
echo "rc=$?" >> $O/tests_full.txt
grep -E "passed|failed|error|rc=" $O/tests_full.txt | tail -5

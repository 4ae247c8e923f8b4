# usage: bash tools/gpu/pmc_layer.sh <outdir> <lib .so or ""> <layer> <what> [extra pmc_layer.py arguments, e.g. --tune 35=1]
O=$1; LIB=$2; L=$3; W=$4; shift 4; EXTRA="$*"
mkdir -p $O
export TMPDIR=/tmp
[ -n "$LIB" ] && export DCT_LIB_PATH=$PWD/$LIB
run() { n=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "igemm|wgrad" -d $O/$n -o p --output-format csv -- python3 tools/pmc_layer.py --layer $L --what $W --n 12 $EXTRA > $O/$n.log 2>&1; echo "$n rc=$?"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run b SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM
run c SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE
python3 tools/pmc_layer_report.py $O > $O/report.txt 2>&1
find $O -name "*.csv" -size +200k -delete

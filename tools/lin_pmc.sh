#!/bin/bash
# PMC counters of one GEMM shape on the own core and on the library: tools/lin_pmc.sh <tag> M K N
export TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/$1; mkdir -p $O
for w in own lib; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $O/pmc1_$w -o runc --output-format csv -- python3 tools/lin_one.py $2 $3 $4 $w 3 > $O/pmc1_$w.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc2_$w -o runc --output-format csv -- python3 tools/lin_one.py $2 $3 $4 $w 3 > $O/pmc2_$w.log 2>&1
  python3 tools/pmc_summary.py $O/pmc1_$w > $O/pmc_$w.txt; python3 tools/pmc_summary.py $O/pmc2_$w >> $O/pmc_$w.txt
  grep -h "TF/s\|rror" $O/pmc1_$w.log $O/pmc2_$w.log | head -4 >> $O/pmc_$w.txt
done
rm -rf $O/pmc1_* $O/pmc2_*
cat $O/pmc_own.txt $O/pmc_lib.txt

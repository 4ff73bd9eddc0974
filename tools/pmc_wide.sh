#!/bin/bash
# PMC passes for the wide-shape kernel (run on the GPU box through gpurun; the program goes directly after `--`).
#   bash tools/pmc_wide.sh <tag> <kernel-substring> <script.py> [args...]
# SQ passes incl. the matrix-pipe counters; memory counters one per pass (FETCH_SIZE and WRITE_SIZE do not share a pass).
tag=$1; sub=$2; script=$3; shift 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
            "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pass$i -- python3 $R/$script "$@" > $out/pass$i.log 2>&1
  echo "pass $i ($ctrs): rc=$?"
done
python3 $R/tools/pmc_summary.py $out "$sub" 2>&1 | grep -v "^ *$"

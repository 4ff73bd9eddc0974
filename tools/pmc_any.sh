#!/bin/bash
# PMC passes over any of the repo's python probes (run on the GPU box through gpurun; program directly after `--`).
#   bash tools/pmc_any.sh <tag> <kernel-substring> <script.py> [args...]
# (two SQ passes; a third one with FETCH_SIZE / WRITE_SIZE / GRBM_GUI_ACTIVE made rocprofv3 abort on this image)
# Output: gpurun_out/pmc_<tag>/pass<N>/..., summary printed by tools/pmc_summary.py
tag=$1; sub=$2; script=$3; shift 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
            "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pass$i -- python3 $R/$script "$@" > $out/pass$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $out "$sub" 2>&1 | grep -v "^ *$"

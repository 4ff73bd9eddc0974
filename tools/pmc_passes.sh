#!/bin/bash
# PMC passes over the batch kernel (run on the GPU box through gpurun; program directly after `--`).
#   bash tools/pmc_passes.sh <tag> [quick_bench args...]      (HIPNMF_* variables are taken from the environment)
# Output: gpurun_out/pmc_<tag>/pass<N>/..., summary printed by tools/pmc_summary.py
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
args="--batch 2048 --iters 200 --rowmajor --threads 512 --reps 1 $@"
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" \
            "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  # (the memory-side counters one set per pass: FETCH_SIZE with GRBM_GUI_ACTIVE, or WRITE_SIZE with TCC_*, in ONE pass made
  #  rocprofv3 abort on this image and the summary then ran on incomplete data)
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pass$i -- python3 $R/tools/quick_bench.py $args > $out/pass$i.log 2>&1
  rc=$?
  [ $rc -ne 0 ] && echo "pass $i ($ctrs): rocprofv3 exit code $rc -- its counters are missing from the summary below"
done
python3 $R/tools/pmc_summary.py $out "${PMC_KERNEL:-fit_}" 2>&1 | grep -v "^ *$"

#!/bin/bash
# A/B of the 4x4x1 / 4x4x4 wide kernels against the 16x16x4 kernel they replace (HIPNMF_WIDE4=0), same box, same inputs:
#   gpurun -- bash tools/wide4_ab.sh   ->  gpurun_out/wide4_ab.log  (committed as profiles/r03_wide4_ab.log)
cd $GRAFT_REPO_ROOT
out=gpurun_out/wide4_ab.log
: > $out
for cfg in "1024 10000 64 8 float32" "4096 2500 64 8 float32" "8192 1250 64 8 float32" "4096 2500 48 6 float32" "2048 2500 128 8 float32" "4096 2500 64 4 float32" \
           "512 10000 64 8 float64" "2048 2500 64 8 float64" "2048 2500 48 6 float64" "1024 2500 128 8 float64" "1024 2500 96 6 float64"; do
  set -- $cfg
  for w4 in 0 1; do
    line=$(HIPNMF_WIDE4=$w4 timeout 300 python tools/quick_bench.py --batch $1 --T $2 --m $3 --k $4 --dtype $5 --iters 100 --rowmajor --threads 0 --reps 3 2>&1 | grep "rep=2")
    echo "B=$1 T=$2 m=$3 k=$4 $5 HIPNMF_WIDE4=$w4 : $line" | tee -a $out
  done
done

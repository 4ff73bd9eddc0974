#!/bin/bash
# ONE long wide matrix (the reference's single-DataFrame call on an HD-EMG recording): the row-sliced path on the 16x16x4 kernel
# (HIPNMF_WIDE4_SLICED=0) vs on the 4x4 kernels.   gpurun -- bash tools/sliced_ab.sh  ->  gpurun_out/sliced_ab.log
cd $GRAFT_REPO_ROOT
out=gpurun_out/sliced_ab.log
: > $out
for cfg in "float32 64 8 10000" "float32 64 8 100000" "float32 128 8 20000" "float32 40 4 50000" "float64 64 8 10000" "float64 64 8 100000" "float64 48 6 30000" "float64 128 8 20000"; do
  set -- $cfg
  for v in 0 1; do
    line=$(HIPNMF_WIDE4_SLICED=$v timeout 300 python tools/quick_bench.py --batch 1 --T $4 --m $2 --k $3 --dtype $1 --iters 300 --rowmajor --threads 0 --reps 3 2>&1 | grep "rep=2")
    us=$(echo "$line" | sed 's/.*kernel=\([0-9.]*\) ms.*/\1/' | awk '{printf "%.1f", $1 * 1000 / 300}')
    echo "$1 m=$2 k=$3 T=$4 HIPNMF_WIDE4_SLICED=$v : ${us} us per iteration  | $(echo "$line" | sed 's/.*err0/err0/')" | tee -a $out
  done
done

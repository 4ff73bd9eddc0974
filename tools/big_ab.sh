#!/bin/bash
# A/B of the one-pass kernel (nmf_big1.hpp) against the two-pass pair (HIPNMF_BIG1=0) over general shapes; run on the GPU box:
#   gpurun -- 'bash tools/big_ab.sh > gpurun_out/big_ab.log'
R=${GRAFT_REPO_ROOT:-.}
run() {  # batch m k [extra env]
  for on in 1 0; do
    HIPNMF_BIG1=$on $4 python3 $R/tools/quick_bench.py --batch $1 --m $2 --k $3 --iters 30 --threads 0 --reps 2 --rowmajor 2>&1 | grep "rep=1" | \
      awk -v on=$on -v b=$1 -v m=$2 -v k=$3 -v e="$4" '{print "B=" b " m=" m " k=" k " BIG1=" on " " e " : " $6 " ms  " $8 " M matrix-it/s  " $NF}'
  done
}
run 64 512 32
run 64 512 64
run 64 512 48
run 64 384 48
run 64 512 16
run 64 300 20
run 128 256 32
run 128 256 64
run 128 200 48
run 256 128 40
run 256 128 64
run 256 96 48
run 512 64 40
run 128 256 16 "env HIPNMF_WIDE_XL=0"
run 128 160 12 "env HIPNMF_WIDE_XL=0"
echo "-- 129..256 channels, k <= 16 on the one-wave kernel (fit_wide xl) for comparison"
python3 $R/tools/quick_bench.py --batch 128 --m 256 --k 16 --iters 30 --threads 0 --reps 2 --rowmajor 2>&1 | grep "rep=1"
python3 $R/tools/quick_bench.py --batch 128 --m 160 --k 12 --iters 30 --threads 0 --reps 2 --rowmajor 2>&1 | grep "rep=1"

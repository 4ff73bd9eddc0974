#!/bin/bash
# Trial-sized matrices (a few hundred to a few thousand samples), 16 384 per batch: a workgroup per matrix (lane mappings,
# HIPNMF_SMALL=0 HIPNMF_FORCE_WIDE=-1) vs the library's choice vs the 4x4 matrix-pipe kernels forced (HIPNMF_FORCE_WIDE=1).
#   gpurun -- bash tools/short_ab.sh   ->  gpurun_out/short_ab.log  (committed as profiles/r03_short_matrices_ab.log)
cd $GRAFT_REPO_ROOT
out=gpurun_out/short_ab.log
: > $out
run() {  # label, env..., then quick_bench args
  local label=$1; shift
  local line=$(env "$@" 2>&1 | grep "rep=1" | sed 's/threads=0 rep=1 //')
  echo "  $label: $line" | tee -a $out
}
for cfg in "float32 16 5 300" "float32 16 5 500" "float32 16 5 700" "float32 16 5 1000" "float32 16 5 2400" "float32 16 8 500" "float32 12 7 500" "float32 8 4 500" "float32 8 4 1000" \
           "float32 32 4 300" "float32 32 4 1200" "float32 24 3 600" "float32 32 8 2500" \
           "float64 8 4 500" "float64 8 6 500" "float64 5 3 400" "float64 16 5 300" "float64 16 5 600" "float64 16 8 1200" "float64 32 8 2500" "float64 24 6 2500"; do
  set -- $cfg
  B=16384; [ $4 -gt 1200 ] && B=8192
  echo "== $1 m=$2 k=$3 T=$4 B=$B" | tee -a $out
  args="python tools/quick_bench.py --batch $B --T $4 --m $2 --k $3 --dtype $1 --iters 100 --rowmajor --threads 0 --reps 2"
  run "workgroup per matrix (lane mapping)" HIPNMF_SMALL=0 HIPNMF_FORCE_WIDE=-1 timeout 300 $args
  run "library's choice                   " timeout 300 $args
  run "4x4 matrix-pipe kernel forced      " HIPNMF_FORCE_WIDE=1 timeout 300 $args
done

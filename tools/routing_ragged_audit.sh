#!/bin/bash
# Routes of the ragged / multi-trial entry points on reference-shaped input (float64 frames of 8-16 muscles, trials of 200-5 000 rows,
# k = 2..8: project/segment.py:160-207 -> analysis.py:907-912), each call under three routings: the library's choice, the lane mappings
# pinned (HIPNMF_FORCE_WIDE=-1) and the matrix-pipe kernels pinned (HIPNMF_FORCE_WIDE=1).  A default that is the slowest of the
# three by more than 10 % is a losing route.  Output -> profiles/<round>_routing_ragged.log   
R=$(cd "$(dirname "$0")/.." && pwd)
run() {  # args of tools/ragged_bench.py
  for v in "HIPNMF_FORCE_WIDE=0" "HIPNMF_FORCE_WIDE=-1" "HIPNMF_FORCE_WIDE=1"; do
    case $v in *=0) tag="default";; *=-1) tag="lanes  ";; *) tag="matrix ";; esac
    printf '%s ' "$tag"
    env $v python3 "$R/tools/ragged_bench.py" "$@" 2>&1 | tail -1
  done
  echo
}
for dt in float64 float32; do
  for m in 8 12 16; do
    run --entry fit_ragged --dtype $dt --m $m --k 4 --trials 6 --tmin 200 --tmax 600
    run --entry fit_ragged --dtype $dt --m $m --k 4 --trials 40 --tmin 200 --tmax 600
    run --entry fit_ragged --dtype $dt --m $m --k 4 --trials 300 --tmin 200 --tmax 600
    run --entry fit_ragged --dtype $dt --m $m --k 4 --trials 40 --tmin 800 --tmax 1500
    run --entry fit_ragged --dtype $dt --m $m --k 4 --trials 40 --tmin 2000 --tmax 5000
  done
  run --entry fit_ragged --dtype $dt --m 16 --k 2 --trials 40 --tmin 200 --tmax 600
  run --entry fit_ragged --dtype $dt --m 16 --k 8 --trials 40 --tmin 200 --tmax 600
  run --entry fit_ragged --dtype $dt --m 16 --k 8 --trials 40 --tmin 2000 --tmax 5000
  run --entry fit_ragged --dtype $dt --m 16 --k 8 --trials 300 --tmin 800 --tmax 1500
  run --entry fit_ragged --dtype $dt --m 12 --k 4 --trials 40 --tmin 200 --tmax 5000
  run --entry fit_ragged --dtype $dt --m 12 --k 4 --trials 40 --tmin 200 --tmax 600 --loss kullback-leibler
  run --entry rank_sweep --dtype $dt --m 16 --kmin 2 --kmax 8 --trials 60 --tmin 1500 --tmax 1500
  run --entry rank_sweep_native --dtype $dt --m 16 --kmin 2 --kmax 8 --trials 60 --tmin 1500 --tmax 1500
  run --entry rank_sweep_native --dtype $dt --m 8 --kmin 2 --kmax 6 --trials 300 --tmin 400 --tmax 400
  run --entry rank_sweep_native --dtype $dt --m 12 --kmin 2 --kmax 8 --trials 12 --tmin 3000 --tmax 3000
  run --entry find_synergies_batched --dtype $dt --m 12 --kmin 2 --kmax 6 --trials 40 --tmin 200 --tmax 600
  run --entry find_synergies_batched --dtype $dt --m 16 --kmin 2 --kmax 8 --trials 12 --tmin 2000 --tmax 5000
done

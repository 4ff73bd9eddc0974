#!/bin/bash
# Kullback-Leibler routing A/B (round 5): the lane mappings / the 16x16x4 kernel (HIPNMF_FORCE_WIDE=-1, HIPNMF_WIDE4=0) against
# the library's choice (fit_wide4_kernel / fit_wide4d_kernel with LOSS = 1).  Output: one line per run, copied to
# profiles/<round>_kl_routing_ab.log by tools/profile_round.sh.
R=$(cd "$(dirname "$0")/.." && pwd)
run() {  # dtype m k T B
  for v in "HIPNMF_FORCE_WIDE=-1 HIPNMF_WIDE4=0" "HIPNMF_FORCE_WIDE=0"; do
    printf '%s m=%d k=%d T=%d B=%d [%s] ' "$1" "$2" "$3" "$4" "$5" "$v"
    env $v python3 "$R/tools/quick_bench.py" --m "$2" --k "$3" --T "$4" --batch "$5" --iters 100 --loss kullback-leibler --threads 0 --rowmajor --dtype "$1" 2>&1 | tail -1 | sed 's/^threads=0 max_slices=0 rep=1 //'
  done
}
run float32 32 8 2500 4096
run float32 32 8 300 8192
run float32 32 6 2500 4096
run float32 24 6 2500 4096
run float32 28 7 1000 4096
run float32 32 5 2500 4096
run float64 64 8 2500 2048
run float64 48 6 2500 2048
run float64 96 4 1000 2048
run float64 128 8 1000 2048
run float64 32 8 2500 2048
run float64 24 6 2500 2048
run float64 16 5 2500 2048
run float64 32 8 128 8192
run float64 12 3 1000 4096
run float64 8 4 2500 2048

#!/bin/bash
# Round-5 routing audit in one table: the routes of the start of the round (lane mappings wherever they exist, one workgroup per matrix
# for the Kullback-Leibler loss, 16x16x4 in the row-sliced path up to 32 channels, 4-wave KL instances) against the library's choice now.
# Output -> profiles/<round>_routing_before_after.log
R=$(cd "$(dirname "$0")/.." && pwd)
OLD="HIPNMF_FORCE_WIDE=-1 HIPNMF_KL_SLICED=0 HIPNMF_WIDE4_SLICED_NARROW=0 HIPNMF_KL_WAVES=4"
run() {  # dtype loss m k T B
  for v in "$OLD" "HIPNMF_FORCE_WIDE=0"; do
    tag=$([ "$v" = "$OLD" ] && echo before || echo "after ")
    printf '%s %-16s m=%-3d k=%-2d T=%-7d B=%-5d %s ' "$1" "$2" "$3" "$4" "$5" "$6" "$tag"
    env $v python3 "$R/tools/quick_bench.py" --m "$3" --k "$4" --T "$5" --batch "$6" --iters 100 --loss "$2" --threads 0 --rowmajor --dtype "$1" 2>&1 | tail -1 | awk '{print $6, $7, $NF}'
  done
}
run float64 frobenius 24 6 10000 1
run float64 frobenius 32 8 10000 1
run float64 frobenius 32 8 3000 32
run float64 frobenius 16 8 30000 2
run float64 frobenius 12 4 600 8
run float64 frobenius 32 8 600 100
run float32 frobenius 32 8 3000 32
run float32 frobenius 24 6 600 100
run float64 kullback-leibler 24 6 10000 1
run float64 kullback-leibler 128 6 5000 1
run float64 kullback-leibler 32 8 128 8192
run float64 kullback-leibler 16 5 2500 2048
run float32 kullback-leibler 64 8 100000 1
run float32 kullback-leibler 16 8 10000 1
run float32 kullback-leibler 32 8 2500 4096
run float32 kullback-leibler 32 8 300 8192

#!/bin/bash
# Routes of the ragged / multi-trial entry points on reference-shaped input (float64 frames of 8-16 muscles, trials of 200-5 000 rows,
# k = 2..8: project/segment.py:160-207 -> analysis.py:907-912), every call under three routings: the library's choice, the lane mappings
# pinned (HIPNMF_FORCE_WIDE=-1) and the matrix-pipe kernels pinned (HIPNMF_FORCE_WIDE=1).  One process per routing (the switches are
# read once per process), all cases inside it (tools/ragged_bench.py --suite); tools/routing_ragged_table.py joins the three logs and
# marks a default that is more than 10 % slower than the better pinned route.   
#   bash tools/routing_ragged_audit.sh <outdir>       -> <outdir>/routing_ragged.log
R=$(cd "$(dirname "$0")/.." && pwd)
O=${1:-$R/gpurun_out/r06}
mkdir -p "$O"
HIPNMF_FORCE_WIDE=0 python3 "$R/tools/ragged_bench.py" --suite > "$O/ragged_default.log" 2>&1
HIPNMF_FORCE_WIDE=-1 python3 "$R/tools/ragged_bench.py" --suite > "$O/ragged_lanes.log" 2>&1
HIPNMF_FORCE_WIDE=1 python3 "$R/tools/ragged_bench.py" --suite > "$O/ragged_matrix.log" 2>&1
python3 "$R/tools/routing_ragged_table.py" "$O/ragged_default.log" "$O/ragged_lanes.log" "$O/ragged_matrix.log" | tee "$O/routing_ragged.log"

#!/bin/bash
# Soak of the cooperative kernel's exchange on the library as built (run on the GPU box: gpurun -- bash tools/coop_soak.sh [fits]).
# Float same-XCD, float device-scope (one and eight matrices), double; then every flavour with the generation numbers
# started next to the 32-bit wrap.  Summary: gpurun_out/coop_soak_summary.log (copy it to profiles/).
cd $GRAFT_REPO_ROOT
N=${1:-10000}
mkdir -p gpurun_out
S=gpurun_out/coop_soak_summary.log
echo "library sha256: $(sha256sum muscle_synergies_amd/lib/libhip_nmf.so | cut -c1-16)  $(date -u +%FT%TZ)" > $S
run() { timeout 1500 python -u tools/coop_soak.py "$@" 2>&1 | grep "coop_soak:\|MISMATCH\|Error\|error" | tail -3 >> $S; echo "rc=$? args: $*" >> $S; }
run --fits $N --dtype float32
run --fits $N --dtype float32 --device-scope
run --fits $N --dtype float64
run --fits $((N / 4)) --dtype float32 --matrices 8
run --fits $((N / 10)) --dtype float32 --gen-base 4294967200 --iters 200
run --fits $((N / 10)) --dtype float32 --device-scope --gen-base 4294967200 --iters 200
run --fits $((N / 10)) --dtype float64 --gen-base 4294967200 --iters 200
cat $S

#!/bin/bash
# Long randomised soak of all four cross-checks (run on the GPU box: gpurun -- bash tools/fuzz_soak.sh); every driver under a timeout.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soak
for s in $(seq 500 517); do timeout 280 python -u tests/fuzz_gpu.py --cases 400 --seed $s --wide-frac 0.4 --verbose > gpurun_out/soak/nmf$s.log 2>&1; echo "nmf $s rc=$? $(grep -v '^RUN' gpurun_out/soak/nmf$s.log | grep 'fuzz:\|MISMATCH\|ERROR' | tail -2)"; done
for s in $(seq 500 509); do timeout 280 python -u tests/fuzz_envelope_gpu.py --cases 300 --seed $s > gpurun_out/soak/env$s.log 2>&1; echo "env $s rc=$? $(grep 'cases\|MISMATCH\|ERROR' gpurun_out/soak/env$s.log | tail -2)"; done
for s in $(seq 500 505); do timeout 280 python -u tests/fuzz_sosfilt_gpu.py --cases 250 --seed $s > gpurun_out/soak/sos$s.log 2>&1; echo "sos $s rc=$? $(grep 'cases\|MISMATCH\|ERROR' gpurun_out/soak/sos$s.log | tail -2)"; done
for s in $(seq 500 505); do timeout 280 python -u tests/fuzz_shard_gpu.py --cases 150 --seed $s > gpurun_out/soak/shard$s.log 2>&1; echo "shard $s rc=$? $(grep 'cases\|MISMATCH\|ERROR' gpurun_out/soak/shard$s.log | tail -2)"; done

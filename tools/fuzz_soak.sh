#!/bin/bash
# the four randomised cross-checks against the oracle at soak sizes (gpurun -- 'bash tools/fuzz_soak.sh r05')
tag=${1:-r05}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/fuzz_$tag
mkdir -p $O
cd $R
{
  echo "library: $(sha256sum muscle_synergies_amd/lib/libhip_nmf.so | cut -c1-16)"
  for seed in 101 102 103 104; do echo -n "fuzz_gpu seed $seed: "; python3 tests/fuzz_gpu.py --cases 500 --seed $seed 2>&1 | tail -1; done
  for seed in 201 202 203; do echo -n "fuzz_envelope_gpu seed $seed: "; python3 tests/fuzz_envelope_gpu.py --cases 600 --seed $seed 2>&1 | tail -1; done
  for seed in 301 302 303; do echo -n "fuzz_sosfilt_gpu scan seed $seed: "; python3 tests/fuzz_sosfilt_gpu.py --cases 500 --seed $seed --mode scan 2>&1 | tail -1; done
  for seed in 501 502; do echo -n "fuzz_shard_gpu seed $seed: "; python3 tests/fuzz_shard_gpu.py --cases 200 --seed $seed 2>&1 | tail -1; done
  for seed in 401 402; do echo -n "fuzz_sosfilt_gpu exact seed $seed: "; python3 tests/fuzz_sosfilt_gpu.py --cases 400 --seed $seed 2>&1 | tail -1; done
} | tee $O/${tag}_fuzz_soak_summary.log

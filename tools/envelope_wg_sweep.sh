#!/bin/bash
# Full-length envelope: emg_wave_kernel (HIPNMF_ENV_WG=0) against emg_wg_kernel over the series length (run on the GPU box).
cd $GRAFT_REPO_ROOT
for dt in float32 float64; do for T in 4096 6144 8192 10240 12288 16384 20000; do
  a=$(HIPNMF_ENV_WG=0 python tools/envelope_bench.py --T $T --dtype $dt 2>/dev/null | grep "^B=" | head -1 | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  b=$(python tools/envelope_bench.py --T $T --dtype $dt 2>/dev/null | grep "^B=" | head -1 | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  echo "$dt T=$T wave=$a wg=$b"
done; done

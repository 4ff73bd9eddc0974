#!/bin/bash
# emg_chunk_kernel against the kernels it replaces, by series length (fp32 and fp64; full length and time-normalised to 200)
for dt in float32 float64; do
for T in 1000 1500 2000 3000 4000 6000 8000 10000 12000 14000 17000 20000; do
  B=$(( 327680000 / T / 16 ))
  for c in 1 0; do
    echo -n "T=$T $dt chunk=$c: "
    HIPNMF_ENV_CHUNK=$c HIPNMF_ENV_CHUNK_MIN_T=0 python3 tools/envelope_bench.py --T $T --batch $B --dtype $dt --reps 9 2>&1 | grep -o "reduce_to=[A-Za-z0-9]*\|[0-9.]* ms (min [0-9.]*)" | paste - - - - 
  done
done; done

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
L=$R/muscle_synergies_amd/lib
python3 -m pytest tests/test_filters.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
python3 tests/fuzz_sosfilt_gpu.py --mode scan 2>&1 | tail -2
for rep in 1 2; do
  for lib in libhip_nmf.so libhip_nmf_sosnosplit.so; do
    HIPNMF_LIBRARY=$L/$lib python3 tools/filter_bench.py --orders 4 --dtypes float32 float64 --modes scan 2>&1 | grep -v amdgpu.ids | sed "s|^|$lib |"
  done
done | tee $O/r06_filter_split_ab.log
for lib in libhip_nmf_sostiming.so libhip_nmf_sostimingold.so; do echo "== $lib"; HIPNMF_LIBRARY=$L/$lib python3 tools/sos_phase_timing.py 2>&1 | grep -v amdgpu.ids; done | tee $O/r06_sos_phase_timing.txt
for lib in "$L/libhip_nmf.so" "$L/libhip_nmf_klnopk.so"; do
  for rep in 1 2; do HIPNMF_LIBRARY=$lib python3 tools/quick_bench.py --batch 2048 --iters 200 --threads 512 --loss kullback-leibler --rowmajor --reps 2 2>&1 | tail -1 | sed "s|^|KL narrow lib=$(basename $lib) |"; done
  HIPNMF_LIBRARY=$lib python3 tools/quick_bench.py --batch 4096 --T 2500 --m 8 --k 4 --iters 200 --threads 512 --loss kullback-leibler --rowmajor --reps 2 2>&1 | tail -1 | sed "s|^|KL 8x4 lib=$(basename $lib) |"
done | tee $O/r06_kl_narrow_pk_ab.log

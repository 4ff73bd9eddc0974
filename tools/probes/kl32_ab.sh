# A/B of the Kullback-Leibler routing for up to 32 channels: lane mappings (HIPNMF_FORCE_WIDE=-1) vs the library's choice
python -m pytest tests/test_gpu_wide4.py tests/test_gpu_wide.py tests/test_gpu_kl.py -x -q -m gpu -k "kullback or kl" 2>&1 | tail -15
for shape in "32 8 300 8192" "32 8 10000 1024" "32 6 2500 4096" "28 7 1000 4096" "17 6 2500 4096" "24 8 600 128" ; do
  set -- $shape
  for fw in -1 0; do
    echo "SHAPE m=$1 k=$2 T=$3 B=$4 FORCE_WIDE=$fw"
    HIPNMF_FORCE_WIDE=$fw python tools/quick_bench.py --m $1 --k $2 --T $3 --batch $4 --iters 100 --loss kullback-leibler --threads 0 --rowmajor 2>&1 | tail -1
  done
done
echo FROBENIUS wide4
python tools/quick_bench.py --m 64 --k 8 --T 2500 --batch 4096 --iters 100 --threads 0 --rowmajor 2>&1 | tail -2
python tools/quick_bench.py --m 48 --k 6 --T 2500 --batch 4096 --iters 100 --threads 0 --rowmajor 2>&1 | tail -2
python tools/quick_bench.py --m 32 --k 8 --T 2500 --batch 4096 --iters 100 --threads 0 --rowmajor 2>&1 | tail -2
python tools/quick_bench.py --m 64 --k 8 --T 2500 --batch 4096 --iters 100 --threads 0 --rowmajor --loss kullback-leibler 2>&1 | tail -2

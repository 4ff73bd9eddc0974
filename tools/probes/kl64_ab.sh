# A/B of the float64 Kullback-Leibler routing for up to 32 channels: lane mappings (HIPNMF_FORCE_WIDE=-1) vs the library's choice
for shape in "8 4 2500 2048" "12 3 1000 4096" "16 8 600 4096" "16 5 200 16384" "4 2 1000 4096" "16 5 10000 1024" "6 3 128 16384" "32 8 128 8192" "2 1 5000 1024"; do
  set -- $shape
  for v in "HIPNMF_FORCE_WIDE=-1" "HIPNMF_FORCE_WIDE=0"; do
    echo "SHAPE m=$1 k=$2 T=$3 B=$4 $v"
    env $v python tools/quick_bench.py --m $1 --k $2 --T $3 --batch $4 --iters 100 --loss kullback-leibler --threads 0 --rowmajor --dtype float64 2>&1 | tail -1
  done
done

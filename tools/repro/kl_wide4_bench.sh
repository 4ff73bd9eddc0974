for mk in "64 8" "64 4" "128 8" "96 8" "48 4"; do set -- $mk
for w4 in 1 0; do
echo "== m=$1 k=$2 WIDE4=$w4"
HIPNMF_WIDE4=$w4 python3 tools/quick_bench.py --batch 2048 --T 1000 --m $1 --k $2 --iters 200 --loss kullback-leibler --reps 3 2>&1 | tail -2
done; done

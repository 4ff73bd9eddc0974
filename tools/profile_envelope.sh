#!/bin/bash
# rocprofv3 kernel statistics and beyond-L2 bytes of the envelope benchmark (gpurun -- 'bash tools/profile_envelope.sh r04')
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_env_$tag
rm -rf $O && mkdir -p $O
for mode in none 200; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$mode -- python3 $R/tools/envelope_bench.py --reduce-to $mode > $O/${tag}_envelope_bench_$mode.log 2>&1
  f=$(find $O/stats_$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -v "at::native" $f > $O/${tag}_kernel_stats_envelope_bench_$mode.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${mode}_$c -- python3 $R/tools/envelope_bench.py --reps 2 --reduce-to $mode > $O/pmc_${mode}_$c.log 2>&1
  done
done
python3 - <<PY > $O/${tag}_pmc_envelope_chunk_traffic.txt
import csv, glob, collections
print("1024 x 16 x 20 000 fp32, W = 200, zero-centred, max-normalised: raw samples 1 310 720 000 B in; out: full length the same, 200 points 13 107 200 B")
print("(FETCH_SIZE counts 64 B per 128-B request on gfx950: double it -- MI355X_MICROARCH.md, HBM section)")
for mode in ("none", "200"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = collections.defaultdict(list)
        for f in glob.glob("$O/pmc_%s_%s/**/*_counter_collection.csv" % (mode, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c and "emg_" in r["Kernel_Name"]:
                    vals[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in vals.items():
            print("reduce_to=%s" % mode, c, "KiB per launch of", k, ":", v)
PY
grep -h reduce_to $O/${tag}_envelope_bench_*.log; cat $O/${tag}_pmc_envelope_chunk_traffic.txt; cat $O/${tag}_kernel_stats_envelope_bench_*.csv

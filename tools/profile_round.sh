#!/bin/bash
# rocprofv3 kernel statistics of the round's benchmarks (run on the GPU box: gpurun -- 'bash tools/profile_round.sh r06')
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
rm -rf $O && mkdir -p $O
run() {  # name, program args...
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 "$@" > $O/$name.log 2>&1
  f=$(find $O/$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${tag}_kernel_stats_$name.csv
}
run bench_steps2 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline
run bench_wide_m64_k8 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --m 64 --k 8
run bench_wide_m128_k16 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --m 128 --k 16 --batch 512
run bench_xl_m256_k16 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --m 256 --k 16 --batch 256 --iters 100
run bench_big_m512_k32 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --m 512 --k 32 --batch 64 --iters 50
run bench_config4 $R/bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline
run bench_config5 $R/bench.py --config 5 --steps 1 --warmup 1 --no-cpu-baseline
run bench_config2 $R/bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline
# float64 -- the reference's own dtype -- on configs 3 / 4 / 2, and config 5 with RCCL in a world of one (round 6)
run bench_f64 $R/bench.py --dtype f64 --steps 2 --warmup 1 --no-cpu-baseline
run bench_f64_config4 $R/bench.py --dtype f64 --config 4 --steps 1 --warmup 1 --no-cpu-baseline
run bench_f64_config2 $R/bench.py --dtype f64 --config 2 --steps 2 --warmup 1 --no-cpu-baseline
run bench_config5_nccl $R/bench.py --config 5 --force-nccl --steps 1 --warmup 1 --no-cpu-baseline
run filter_bench $R/tools/filter_bench.py --orders 4 --dtypes float32 float64
for n in bench_steps2 bench_wide_m64_k8 bench_wide_m128_k16 bench_xl_m256_k16 bench_big_m512_k32 bench_config4 bench_config5 bench_config2 bench_f64 bench_f64_config4 bench_f64_config2 bench_config5_nccl; do
  grep -h '^{' $O/$n.log > $O/${tag}_${n/bench_steps2/bench}.json
done
cp $O/filter_bench.log $O/${tag}_filter_bench.log
# bytes beyond L2 of the time-parallel filter kernel (separate PMC passes, program directly after `--`)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_scan_$c -- python3 $R/tools/filter_bench.py --orders 4 --dtypes float32 --modes scan --zero-lag 1 > $O/pmc_scan_$c.log 2>&1
done
python3 - <<PY > $O/${tag}_pmc_scan_traffic.txt
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob("$O/pmc_scan_%s/**/*_counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and ("sosfilt_chunk_kernel" in r["Kernel_Name"] or "sosfilt_scan_kernel" in r["Kernel_Name"]):
                vals.append(float(r["Counter_Value"]))
    print(c, "KiB per launch of sosfilt_chunk_kernel<float,2,79>:", vals, "(1024 x 16 x 20 000 fp32, zero-lag order 4; algorithmic 2 x 1 310 720 000 B; FETCH_SIZE counts 64 B per 128-B request on gfx950: double it)")
PY
bash $R/tools/profile_envelope.sh $tag > $O/profile_envelope.log 2>&1
cd /tmp
run kl_narrow $R/tools/quick_bench.py --batch 2048 --iters 200 --threads 512 --loss kullback-leibler --rowmajor --reps 3
run kl_wide4_64x8 $R/tools/quick_bench.py --batch 2048 --T 1000 --m 64 --k 8 --iters 200 --loss kullback-leibler --reps 3
run kl_big_m512_k32 $R/tools/quick_bench.py --batch 64 --m 512 --k 32 --iters 50 --threads 0 --reps 3 --rowmajor --loss kullback-leibler
run big_stop_rule_m512_k32 $R/tools/quick_bench.py --batch 64 --m 512 --k 32 --iters 50 --threads 0 --reps 3 --rowmajor --tol 1e-9
run kl_wide4_32x8 $R/tools/quick_bench.py --batch 4096 --T 2500 --m 32 --k 8 --iters 100 --threads 0 --loss kullback-leibler --rowmajor --reps 3
run kl_wide4d_64x8_f64 $R/tools/quick_bench.py --batch 2048 --T 2500 --m 64 --k 8 --iters 100 --threads 0 --loss kullback-leibler --rowmajor --reps 3 --dtype float64
for n in kl_narrow kl_wide4_64x8 kl_big_m512_k32 big_stop_rule_m512_k32 kl_wide4_32x8 kl_wide4d_64x8_f64; do grep -h "rep=" $O/$n.log > $O/${tag}_$n.log; done
if [ "${PROFILE_ROUND_FULL:-0}" = "1" ]; then  # round 5's routing audits (minutes of process start-ups; unchanged routes since)
  bash $R/tools/kl_routing_ab.sh > $O/${tag}_kl_routing_ab.log 2>&1
  bash $R/tools/routing_before_after.sh > $O/${tag}_routing_before_after.log 2>&1
fi
# round 6: PMC traffic entries of the bench lines (profiles/traffic.json), the routing crossovers re-derived on this box, the phase
# breakdown of config #2's iteration (timing build), DISPATCH.md regenerated from real calls
cd $R
export HIPNMF_ROUND=$tag
bash tools/measure_traffic.sh > $O/traffic_headline.log 2>&1
bash tools/measure_traffic.sh --dtype f64 > $O/traffic_f64.log 2>&1
bash tools/measure_traffic.sh --m 64 --k 8 --batch 1024 > $O/traffic_wide4.log 2>&1
bash tools/measure_traffic.sh --m 128 --k 16 --batch 512 > $O/traffic_wide128.log 2>&1
bash tools/measure_traffic.sh --m 256 --k 16 --batch 256 --iters 100 > $O/traffic_wide256.log 2>&1
bash tools/measure_traffic.sh --m 512 --k 32 --batch 64 --iters 50 > $O/traffic_big1.log 2>&1
cp $R/profiles/traffic.json $O/traffic.json
python3 tools/calibrate_routes.py --quick --out $O/${tag}_calibrate_routes.log > /dev/null 2>&1
if [ -f $R/muscle_synergies_amd/lib/libhip_nmf_timing.so ]; then
  HIPNMF_LIBRARY=$R/muscle_synergies_amd/lib/libhip_nmf_timing.so python3 tools/coop_phase_timing.py 2>&1 | grep -v amdgpu.ids > $O/${tag}_coop_phase_timing.txt
fi
python3 tools/dispatch_table.py > $O/dispatch_table.log 2>&1 && cp $R/DISPATCH.md $O/DISPATCH.md
cd /tmp
# the threading clause of the ABI over every entry point, 2 / 3 / 8 host threads (plain ctypes, no torch)
for nt in 2 3 8; do python3 $R/tools/abi_threads_stress.py --threads $nt --rounds 2 2>&1 | tail -1; done > $O/${tag}_abi_threads_stress.log
ls $O/*.csv $O/*.json $O/*.txt $O/*_filter_bench.log $O/*_ab.log

# where the wide4 Kullback-Leibler kernel's cycles go: SQ wait / active breakdown, LDS conflicts, MFMA busy (one counter set per pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/kl_pmc
rm -rf $O && mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/quick_bench.py --m 64 --k 8 --T 2500 --batch 2048 --iters 100 --threads 0 --rowmajor --loss kullback-leibler --reps 1 > $O/p$i.log 2>&1
done
python3 - <<PY
import csv, glob
tot = {}
for f in glob.glob("$O/p*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fit_wide4_kernel" in r["Kernel_Name"]:
            tot.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(tot.items()):
    print(k, sum(v) / len(v), len(v))
PY

#!/bin/bash
# N consecutive runs of the full GPU suite, each a fresh child process with faulthandler on, one line per test in the log
# (-v: the test that was running when a process died is the last line), stderr kept, HIP error logging on (AMD_LOG_LEVEL=1:
# a "Memory access fault by GPU node" or a runtime assertion shows up with its text).  VERDICT r05 next-round item 2b.
#   tools/abort_hunt.sh <runs> <outdir> [extra pytest args...]
# Writes <outdir>/run_<i>.log and <outdir>/summary.txt (rc, passed / failed counts, duration per run).
set -u
N=${1:-5}; OUT=${2:-gpurun_out/r06/hunt}; shift 2 || true
mkdir -p "$OUT"
export PYTHONFAULTHANDLER=1 PYTHONUNBUFFERED=1 AMD_LOG_LEVEL=${AMD_LOG_LEVEL:-1}
: > "$OUT/summary.txt"
for i in $(seq 1 "$N"); do
  t0=$(date +%s)
  python3 -X faulthandler -m pytest tests -m gpu -v -p no:cacheprovider -o faulthandler_timeout=900 "$@" > "$OUT/run_$i.log" 2>&1
  rc=$?
  t1=$(date +%s)
  tail_line=$(grep -E "passed|failed|error" "$OUT/run_$i.log" | tail -1)
  echo "run $i rc=$rc seconds=$((t1 - t0)) :: $tail_line" | tee -a "$OUT/summary.txt"
  if [ $rc -ne 0 ]; then
    echo "---- last 60 lines of run $i ----" >> "$OUT/summary.txt"
    tail -60 "$OUT/run_$i.log" >> "$OUT/summary.txt"
  else
    # a clean run's log is only kept as its last 5 lines (64 MiB limit on what comes home)
    tail -5 "$OUT/run_$i.log" > "$OUT/run_$i.tail" && rm -f "$OUT/run_$i.log"
  fi
done
clean=$(grep -c "rc=0 " "$OUT/summary.txt")
echo "ABORT-HUNT runs=$N clean=$clean" | tee -a "$OUT/summary.txt"

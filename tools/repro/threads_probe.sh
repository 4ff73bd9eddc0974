#!/bin/bash
# Root-cause probe of the threading clause: the torch-free stress (tools/abi_threads_stress.py) on every path, then the
# wide row-sliced (hipGraph) cases alone, then the same under AMD_LOG_LEVEL=3 with the first failing HIP call extracted.
out=gpurun_out/thr; mkdir -p $out
timeout 900 python3 tools/abi_threads_stress.py --threads 3 --rounds 2 > $out/all.log 2>&1; echo "rc=$?" >> $out/all.log
timeout 600 python3 tools/abi_threads_stress.py --threads 3 --rounds 3 --same-order --only wide_sliced,wide4_sliced,wide4d_sliced_stop,wide_sliced_auto > $out/ws.log 2>&1; echo "rc=$?" >> $out/ws.log
AMD_LOG_LEVEL=3 timeout 600 python3 tools/abi_threads_stress.py --threads 3 --rounds 2 --same-order --only wide_sliced_auto,wide4_sliced > $out/ws_amdlog.out 2> /tmp/amdlog.txt; echo "rc=$?" >> $out/ws_amdlog.out
wc -l /tmp/amdlog.txt >> $out/ws_amdlog.out
grep -n "Returned hipError\|returned hipError\|: hipError" /tmp/amdlog.txt | head -100 > $out/ws_amdlog_errors.txt
first=$(grep -n "Returned hipError" /tmp/amdlog.txt | grep -v "hipErrorNotReady" | head -1 | cut -d: -f1)
if [ -n "$first" ]; then sed -n "$((first>400?first-400:1)),$((first+60))p" /tmp/amdlog.txt > $out/ws_amdlog_context.txt; fi
tail -5 $out/all.log $out/ws.log $out/ws_amdlog.out

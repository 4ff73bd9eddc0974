#!/bin/bash
# The torch-free stress of the threading clause (tools/abi_threads_stress.py): every path from 2, 3 and 8 host threads, the
# row-sliced (replayed hipGraph) cases alone in lock step, and the Python-host repro with the frame-size gate lifted.
out=gpurun_out/thr; mkdir -p $out
for n in 2 3 8; do
  timeout 900 python3 tools/abi_threads_stress.py --threads $n --rounds 2 > $out/all_$n.log 2>&1; echo "rc=$?" >> $out/all_$n.log
done
timeout 600 python3 tools/abi_threads_stress.py --threads 3 --rounds 4 --same-order --only wide_sliced,wide4_sliced,wide4d_sliced_stop,wide_sliced_auto,sliced_graph,sliced_graph_stop,coop > $out/ws.log 2>&1; echo "rc=$?" >> $out/ws.log
REPRO_UNLIMITED=1 timeout 600 python3 tools/repro/rank_threads_long_matrix.py > $out/repro_unlimited.log 2>&1; echo "rc=$?" >> $out/repro_unlimited.log
tail -n 4 $out/all_2.log $out/all_3.log $out/all_8.log $out/ws.log $out/repro_unlimited.log

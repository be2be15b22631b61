#!/bin/bash
# round 6, step 1: the dynamic unit queue (units appended by unit 0) + prefix priority: quick probes first, then parity tests, then A/B on one box
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_step1; mkdir -p $O; rm -f $O/*
timeout 120 python tools/probes/split_probe.py tail 64 > $O/tail64_append_prio2.txt 2>&1 || { echo "QUICK PROBE FAILED/HUNG rc $?"; tail -5 $O/tail64_append_prio2.txt; exit 1; }
timeout 900 python -m pytest tests/test_hip_gpu.py -x -q -m gpu -p timeout --timeout 300 --timeout-method thread -k "split_regions_are_bit_identical or concurrent_handles or both_workgroup or g3_assembly" > $O/pytest_split.log 2>&1
echo "pytest rc $?" >> $O/pytest_split.log
for n in 64 256; do
  [ $n = 256 ] && timeout 120 python tools/probes/split_probe.py tail $n > $O/tail${n}_append_prio2.txt 2>&1
  BK_PROBE_FLAGS=16384 timeout 120 python tools/probes/split_probe.py tail $n > $O/tail${n}_prequeue_prio2.txt 2>&1
  BK_LIB=breakmer_amd/libbreakmer_hip_prio0.so timeout 120 python tools/probes/split_probe.py tail $n > $O/tail${n}_append_prio0.txt 2>&1
done
BK_PROBE_WG=256 timeout 120 python tools/probes/split_probe.py tail 256 > $O/tail256_append_prio2_wg256.txt 2>&1
BK_VARIANT=diag BK_DEBUG_SPLIT=1 timeout 120 python tools/probes/split_probe.py tail 256 2>&1 | grep -v "bk launch" | cut -c1-3000 > $O/tail256_diag.txt
tail -n 3 $O/pytest_split.log; grep -H "^n " $O/tail*.txt | cut -c1-220

#!/bin/bash
# round 6, step 4: wide column tiles of the score sweep (13 columns per lane, no per-column mismatch register): parity, then configs[3] / [4]
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_step4; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_hip_gpu.py tests/test_sweep_encoding_cpu.py -x -q -m gpu -p timeout --timeout 400 --timeout-method thread -k "g1_nw or nw_random or g3_assembly or config3 or config4 or overflow_a_cap or long_reads or both_workgroup or lookahead or batch_vs_oracle or full_size_config2_properties" > $O/pytest_tiles.log 2>&1
echo "pytest rc $?" >> $O/pytest_tiles.log
BK_PROBE_KIND=cfg3 BK_PROBE_WGS=0 BK_PROBE_HANDLES=1 timeout 400 python tools/probes/noisy_inflight.py 4096 0 3 > $O/cfg3_4096.txt 2>&1
BK_PROBE_KIND=cfg3 BK_PROBE_WGS=0 BK_PROBE_HANDLES=2 timeout 400 python tools/probes/noisy_inflight.py 2048 0 6 > $O/cfg3_2048x2.txt 2>&1
BK_PROBE_KIND=cfg4 BK_PROBE_WGS=0 BK_PROBE_HANDLES=1 timeout 600 python tools/probes/noisy_inflight.py 768 0 2 > $O/cfg4_768.txt 2>&1
tail -n 3 $O/pytest_tiles.log; grep -h "regions/batch" $O/*.txt

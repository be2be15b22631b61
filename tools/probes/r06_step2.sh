#!/bin/bash
# round 6, step 2: one wavefront per slot for long contigs (wide rounds): configs[3] / configs[4] one batch at a time and in flight
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_step2; mkdir -p $O; rm -f $O/*
BK_PROBE_KIND=cfg3 BK_PROBE_WGS=0 BK_PROBE_HANDLES=1,2 timeout 400 python tools/probes/noisy_inflight.py 2048 0 6 > $O/cfg3_2048.txt 2>&1
BK_PROBE_KIND=cfg3 BK_PROBE_WGS=0 BK_PROBE_HANDLES=1 timeout 400 python tools/probes/noisy_inflight.py 4096 0 3 > $O/cfg3_4096.txt 2>&1
BK_PROBE_KIND=cfg4 BK_PROBE_WGS=0 BK_PROBE_HANDLES=1,2 timeout 600 python tools/probes/noisy_inflight.py 384 0 4 > $O/cfg4_384.txt 2>&1
BK_PROBE_KIND=cfg4 BK_PROBE_WGS=0 BK_PROBE_HANDLES=1 timeout 600 python tools/probes/noisy_inflight.py 768 0 2 > $O/cfg4_768.txt 2>&1
grep -h "regions/batch" $O/*.txt

#!/bin/bash
# parity evidence of a round on the GPU box (randomised sweep vs the oracle + run-to-run identity of a large batch):
#   bash tools/evidence_round.sh r04
# The last two steps exercise the EXPERIMENTAL component split (flags 1280 = on + whatever the size); that path can fault the device
# (DESIGN 4.5), so they come last and each runs in its own process.
tag=${1:-r04}; out=gpurun_out/evidence_$tag; mkdir -p $out
BK_FUZZ_WG=256 timeout 1500 python3 tools/fuzz_parity.py ${FUZZ_N:-400} 411 > $out/fuzz_parity_wg256.log 2>&1
BK_FUZZ_WG=512 timeout 1500 python3 tools/fuzz_parity.py ${FUZZ_N:-400} 412 > $out/fuzz_parity_wg512.log 2>&1
timeout 900 python3 tools/stress_batch.py 0 1500 4 > $out/stress_batch.log 2>&1
timeout 600 python3 tools/probes/split_probe.py soak 64 40 256 0 > $out/soak_default_wg256.log 2>&1
timeout 600 python3 tools/probes/split_probe.py soak 64 40 512 0 > $out/soak_default_wg512.log 2>&1
BK_FUZZ_FLAGS=1280 timeout 1500 python3 tools/fuzz_parity.py ${FUZZ_N:-400} 413 > $out/fuzz_parity_split_forced.log 2>&1
timeout 900 python3 tools/stress_batch.py 1280 600 3 > $out/stress_batch_split_forced.log 2>&1
tail -n 2 $out/*.log

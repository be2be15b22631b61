#!/bin/bash
# round 6, step 3: k-mer stage with four occurrences per thread in flight and the coalesced compaction: parity, then phases and batch times
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_step3; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_hip_gpu.py -x -q -m gpu -p timeout --timeout 400 --timeout-method thread -k "g4_kmer or g3_assembly or bucket_sort or config4 or config3_regions or noisy_regions_and_both or reads_with_n or windows_with_n or large_windows or edge_cases or batch_vs_oracle or full_size_config2_properties" > $O/pytest_kmer.log 2>&1
echo "pytest rc $?" >> $O/pytest_kmer.log
timeout 200 python tools/phase_probe_k_cfg4.py > $O/k_cfg4.txt 2>&1
timeout 120 python tools/phase_probe_k_noise.py 1 0.005 > $O/k_noise_1.txt 2>&1
timeout 120 python tools/phase_probe_k_noise.py 64 0.005 > $O/k_noise_64.txt 2>&1
timeout 120 python tools/phase_probe_k.py > $O/k_headline.txt 2>&1
timeout 120 python tools/probes/split_probe.py tail 64 > $O/tail64.txt 2>&1
BK_PROBE_KIND=cfg4 BK_PROBE_WGS=0 BK_PROBE_HANDLES=1 timeout 600 python tools/probes/noisy_inflight.py 768 0 2 > $O/cfg4_768.txt 2>&1
tail -n 3 $O/pytest_kmer.log; cat $O/k_cfg4.txt $O/k_noise_1.txt $O/k_headline.txt; grep "^n " $O/tail64.txt; grep "regions/batch" $O/cfg4_768.txt

#!/bin/bash
# round 6, step 12: rows loaded a quad at a time (bk_load_words): parity, k-mer phases, headline A/B (oldk = the tree before)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_step12; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_hip_gpu.py -x -q -m gpu -p timeout --timeout 400 --timeout-method thread -k "g4_kmer or edge_cases or reads_with_n or windows_with_n or batch_vs_oracle or g3_assembly or long_reads or config4_config5 or packed_submit or async_submit or more_regions or config2_properties or large_windows" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
timeout 200 python3 tools/phase_probe_k.py > $O/kmer_phases.txt 2>&1
for rep in 1 2 3; do
  for v in "" oldk; do
    lib=breakmer_amd/libbreakmer_hip${v:+_$v}.so
    timeout 200 python bench.py --lib $lib --cpu-sample 0 --other-configs 0 --steps 100 --warmup 6 > $O/bench_${v:-new}_$rep.json 2> /dev/null
  done
done
tail -n 3 $O/pytest.log; cat $O/kmer_phases.txt
python3 - <<'PY'
import json, glob
for fn in sorted(glob.glob("gpurun_out/r06_step12/bench_*.json")):
    try:
        d = json.loads([l for l in open(fn) if l.startswith("{")][-1])
        print(fn.split("/")[-1], d["value"], d["kernels_ms"], d["kernels_ms_inflight"])
    except Exception as e:
        print(fn, "ERR", e)
PY

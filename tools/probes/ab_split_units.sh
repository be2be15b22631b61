#!/bin/bash
# diagnostic (GPU box): units per split region (-DBK_SPLIT_G=...) on 64 noisy regions, same box; the product library is put back at the end
out=gpurun_out/r4g; mkdir -p $out
cp breakmer_amd/libbreakmer_hip.so $out/product.so
for g in ${UNITS:-16 32 64}; do
  ( cd breakmer_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-result -DBK_SPLIT_G=$g -o ../libbreakmer_hip.so bk_api.hip ) > $out/build_$g.log 2>&1
  echo "== $g units"; python3 tools/probes/split_probe.py tail 64 2>&1 | tail -3
done | tee $out/units.txt
cp $out/product.so breakmer_amd/libbreakmer_hip.so; rm -f $out/product.so

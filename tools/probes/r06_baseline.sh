#!/bin/bash
# round-6 baseline of the noisy path on this round's box: where a noisy batch's time goes (k-mer phases, assembler phases, units)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_base; mkdir -p $O
python tools/phase_probe_k_noise.py 1 0.005 > $O/k_noise_1.txt 2>&1
python tools/phase_probe_k_noise.py 64 0.005 > $O/k_noise_64.txt 2>&1
BK_WG=512 BK_FLAGS=128 python tools/phase_probe_noise.py 0.005 > $O/asm_noise_one_unit.txt 2>&1
BK_VARIANT=diag BK_DEBUG_SPLIT=1 python tools/probes/split_probe.py tail 64 > $O/tail64.txt 2>&1
BK_VARIANT=diag BK_DEBUG_SPLIT=1 BK_PROBE_WG=256 python tools/probes/split_probe.py tail 256 > $O/tail256_wg256.txt 2>&1
BK_VARIANT=diag BK_DEBUG_SPLIT=1 BK_PROBE_WG=512 python tools/probes/split_probe.py tail 256 > $O/tail256_wg512.txt 2>&1
python tools/probes/split_probe.py tail 64 > $O/tail64_product.txt 2>&1
python tools/probes/split_probe.py tail 256 > $O/tail256_product.txt 2>&1
tail -n 4 $O/*.txt

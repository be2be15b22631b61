#!/bin/bash
# Parity evidence of a round on the GPU box: the randomised sweep against the oracle (tools/fuzz_parity.py) through the barrier-check +
# jitter build of the library (five seeds: which wavefronts sleep behind which barrier; both workgroup sizes; split as shipped,
# forced on small graphs, switched off) and through the product build, then run-to-run identity of large batches (stress_batch.py).
#   bash tools/evidence_round.sh r05
# The sweeps spend their time in the CPU oracle, so they run side by side (one process each; the GPU is shared).
tag=${1:-r06}; out=gpurun_out/evidence_$tag; mkdir -p $out
N=${FUZZ_N:-240}
run() { # name variant jitter_seed wg flags n fuzz_seed
  BK_VARIANT=$2 BK_JITTER_SEED=$3 BK_FUZZ_WG=$4 BK_FUZZ_FLAGS=$5 timeout 1700 python3 tools/fuzz_parity.py $6 $7 > $out/fuzz_$1.log 2>&1 &
}
run checkjit_seed1_wg256_default   checkjit 1 256 0   $N 511
run checkjit_seed2_wg512_default   checkjit 2 512 0   $N 512
run checkjit_seed3_wg256_splitall  checkjit 3 256 256 $N 513
run checkjit_seed4_wg512_splitall  checkjit 4 512 256 $N 514
run checkjit_seed5_auto_nosplit    checkjit 5 0   128 $N 515
run checkjit_seed6_wg256_prequeue   checkjit 6 256 16640 $N 518      # round 6: the round-5 unit queue (units queued at launch, BK_CFG_TEST_PREQUEUE_UNITS) with the split forced
run product_wg256_default          ""       0 256 0   300 516
run product_auto_splitall          ""       0 0   256 300 517
wait
# round 6: split batches on several handles at once (the cross-handle wait of the round-5 queue, ADVICE) and the split tests, through the check + jitter build
BK_TEST_VARIANT=checkjit BK_JITTER_SEED=8 timeout 1200 python3 -m pytest tests/test_hip_gpu.py -q -m gpu -p timeout --timeout 900 --timeout-method thread -k "concurrent_handles or split_regions_are_bit_identical or full_size_config3_regions or long_reads or overflow_a_cap" > $out/pytest_checkjit_split_and_long_contigs.log 2>&1
timeout 900 python3 tools/stress_batch.py 0 1500 4 > $out/stress_batch.log 2>&1 &
timeout 900 python3 tools/stress_batch.py 256 600 3 > $out/stress_batch_split_forced.log 2>&1 &
BK_VARIANT=checkjit BK_JITTER_SEED=7 timeout 900 python3 tools/stress_batch.py 0 1500 3 > $out/stress_batch_checkjit.log 2>&1 &
wait
tail -n 2 $out/*.log

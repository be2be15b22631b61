"""Randomised parity sweep (not part of the test suite): many small regions with random shapes through the HIP path
and the C oracle.  python tools/fuzz_parity.py [n_regions] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variant  # noqa: E402
_variant.use()                       # BK_VARIANT / BK_LIB: the build of the library to check (default: the product)
import numpy as np  # noqa: E402
from breakmer_amd import hip_backend as hb, synth  # noqa: E402
from oracle import bk_oracle as bo  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
bad = 0
t0 = time.time()
done = 0
while done < n:
    k = int(rng.choice([int(x) for x in os.environ.get("BK_FUZZ_K", "15,21,31,41").split(",")]))
    rc = int(rng.choice([2, 3]))
    batch = []
    for _ in range(min(16, n - done)):
        L = int(rng.choice([60, 100, 150, 250, 400]))
        W = int(rng.integers(max(2 * L + 100, 400), 2200))
        kw = dict(W=W, L=L, depth=int(rng.integers(15, 160)), sv_type=str(rng.choice(synth.SV_TYPES)),
                  noise=float(rng.choice([0.0, 0.0, 0.003, 0.01, 0.03])), var_len=float(rng.choice([0.0, 0.0, 0.2])),
                  indel_only_frac=float(rng.choice([0.0, 0.0, 0.3])), n_frac=float(rng.choice([0.0, 0.0, 0.1, 0.5])), global_seed=seed)
        kw["sv_size"] = int(rng.choice([20, 60, 120, 200])) if kw["sv_type"] in ("del", "ins") else int(rng.choice([100, 200]))
        if kw["sv_size"] >= W // 2 - L:
            kw["sv_size"] = 20
        batch.append((int(rng.integers(0, 1 << 30)), kw))
    regs = [synth.make_region(rid, **kw) for rid, kw in batch]
    for r in regs:                                      # every 5th region gets a whole-gene sized window (global-memory k-mer set, chunked realign)
        if r.region_id % 5 == 0:
            fl = synth.rand_bases(synth.stream_key(seed, r.region_id, 9), 2 * 15000)
            r.window = np.concatenate([fl[:15000], r.window, fl[15000:]]).astype(np.uint8)
    eng = hb.Engine(kmer_size=k, rc_thresh=rc, wg_threads=int(os.environ.get("BK_FUZZ_WG", "0")), flags=int(os.environ.get("BK_FUZZ_FLAGS", "0")))      # BK_FUZZ_FLAGS=256: every region split into units whatever its size; 128: never
    eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens, indel_only=r.indel_only, partners=[p[4] for p in r.partners]) for r in regs])
    eng.run(hb.BK_STAGE_ALL)
    for i, (r, (rid, kw)) in enumerate(zip(regs, batch)):
        targets = [r.window_str] + [synth.codes_to_str(p[4]) for p in r.partners]
        want, info = bo.assemble_region(r.read_strs(), [r.window_str], k, rc, indel_only=r.indel_only.tolist())
        got = [{kk: v for kk, v in c.items() if kk not in ("total_reads", "n_hits")} for c in eng.contigs(i)]
        ok = got == want
        mers = eng.kmers(i)[0]
        ok = ok and mers == [m for m, _ in sorted(zip(info["mers"], info["counts"].tolist()), key=lambda x: (x[1], x[0]), reverse=True)]
        if ok:
            for ci, c in enumerate(want[:6]):
                if eng.hits(i, ci) != bo.realign(c["seq"], targets):
                    ok = False
                    break
        if not ok:
            bad += 1
            print("MISMATCH", rid, k, rc, kw, flush=True)
    done += len(batch)
    print("done %d / %d  bad %d  (%.0f s)" % (done, n, bad, time.time() - t0), flush=True)
print("FUZZ RESULT: %d regions, %d mismatches" % (n, bad))

"""Barrier-discipline evidence on the GPU box (not imported by the product; tests/test_hip_gpu.py runs it in a child process so
that a device fault is a red test, not a dead suite).

    python tools/race_check.py [--variant checkjit|jitter|check|''] [--seeds 1,2,3] what [what ...]

what:
  g3            the reference-made assembly fixtures (tests/golden/assembly.json), both workgroup sizes, as shipped and with the
                component split forced on their small graphs
  mixed         40 small regions (every SV type, 0-1.5 % noise) against the C oracle, as shipped and split forced, realign included
  shape:N[:R]   the batches that faulted in round 4 (profiles/r04/split_fault): N small regions at 1 % noise, R runs (default 6)
                on both workgroup sizes -- records identical between runs, 8 sampled regions equal to the oracle
  noisy:N       N full-size regions (500x, 150 bp) at 0.5 % noise: split (default) against one unit per region, bit for bit
  redo          regions with contigs beyond the dual / pair kernels' columns (a translocation, a 1,500-base insertion, 250-base reads at 3 % noise), with the
                diagnostic flag that makes the score sweep of a long-contig round flag EVERY read (BK_CFG_DIAG_FORCE_REDO; a diagnostic build): every slot
                of those rounds goes through the full overlap DPs, in as many passes of two wavefronts per read as the round has reads -- the path ~2 % of
                such reads take on real data, a few per round at most -- against the oracle, both workgroup sizes, with and without the flag
  caps          regions that overflow a working cap of the assembler (4,700 candidate reads on one k-mer; a contig of 5,300 bases):
                the give-up paths (bk_fail) and the re-run under larger caps, against the oracle -- the path on which the whole-suite
                run through the check build found a missing barrier in round 5

With a jitter variant every `what` is repeated for every seed (BK_JITTER_SEED: which wavefronts sleep behind which barrier).
With a check variant a barrier divergence surfaces as an error of bk_sync naming both sites.  Last line: RACE CHECK RESULT: ok | FAILED."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from breakmer_amd import build, hip_backend as hb, synth  # noqa: E402
from oracle import bk_oracle as bo  # noqa: E402

STRIP = ("total_reads", "n_hits")
bad = 0
MEMO = {}      # the oracle's side of a check is the same for every jitter seed: computed once per process


def memo(key, fn):
    if key not in MEMO:
        MEMO[key] = fn()
    return MEMO[key]


def note(ok, text):
    global bad
    if not ok:
        bad += 1
    print(("ok    " if ok else "FAILED"), text, flush=True)


def strip(cs):
    return [{k: v for k, v in c.items() if k not in STRIP} for c in cs]


def engine(k, rc=2, **kw):
    return hb.Engine(kmer_size=k, rc_thresh=rc, **kw)


def submit(eng, regions):
    eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens, indel_only=r.indel_only, partners=[p[4] for p in r.partners]) for r in regions])


def g3(tag):
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "assembly.json")))
    by = {}
    for c in d["cases"]:
        by.setdefault((c["k"], c["rc_thresh"]), []).append(c)
    for flags in (0, 256):
        wrong = tot = 0
        for (k, rc), cases in by.items():
            regions = [synth.make_region(**c["gen"]) for c in cases]
            for wg in (512, 256):
                eng = engine(k, rc, wg_threads=wg, flags=flags)
                submit(eng, regions)
                eng.run(3)
                for i, c in enumerate(cases):
                    tot += 1
                    wrong += strip(eng.contigs(i)) != c["contigs"]
                eng.close()
        note(wrong == 0, "%s g3 fixtures x workgroup sizes, flags %d: %d cases, %d wrong" % (tag, flags, tot, wrong))


def mixed(tag):
    regions = [synth.make_region(7000 + i, sv_type=synth.SV_TYPES[i % 5], depth=(60, 120, 200)[i % 3], W=(900, 1200)[i % 2], L=(100, 150)[i % 2],
                                 noise=(0.0, 0.004, 0.008, 0.015)[i % 4], n_frac=(0.0, 0.0, 0.1)[i % 3]) for i in range(40)]
    want = memo("mixed", lambda: [bo.assemble_region(r.read_strs(), [r.window_str], 31, 2, indel_only=r.indel_only.tolist())[0] for r in regions])
    def want_hits(i, ci):
        r = regions[i]
        return memo(("mixed hits", i, ci), lambda: bo.realign(want[i][ci]["seq"], [r.window_str] + [synth.codes_to_str(p[4]) for p in r.partners]))
    for flags in (0, 256):
        for wg in (256, 512):
            eng = engine(31, wg_threads=wg, flags=flags)
            submit(eng, regions)
            eng.run(hb.BK_STAGE_ALL)
            wrong = sum(strip(eng.contigs(i)) != want[i] for i in range(len(regions)))
            hw = 0
            for i in range(0, len(regions), 5):
                for ci, c in enumerate(want[i][:3]):
                    hw += eng.hits(i, ci) != want_hits(i, ci)
            note(wrong == 0 and hw == 0, "%s mixed batch vs oracle, flags %d wg %d: %d regions wrong, %d realign mismatches, split regions %d, repair passes %d"
                 % (tag, flags, wg, wrong, hw, eng.stat(28), eng.stat(27)))
            eng.close()


def shape(tag, n, runs):
    base = [synth.make_region(50000 + i, depth=60, L=150, sv_type="del", noise=0.01) for i in range(min(n, 256))]
    regions = [base[i % len(base)] for i in range(n)]
    sample = list(range(0, min(n, 256), 32))[:8]
    want = {i: bo.assemble_region(regions[i].read_strs(), [regions[i].window_str], 31, 2)[0] for i in sample}
    for wg in (256, 512):
        eng = engine(31, wg_threads=wg)
        submit(eng, regions)
        ref = None
        diff = 0
        t0 = time.time()
        for rep in range(runs):
            eng.run(hb.BK_STAGE_ALL)
            cur = [[(c["seq"], tuple(c["others"]), tuple(c["indel_only"]), tuple(c["reads"])) for c in eng.contigs(i)] for i in range(n)]
            if ref is None:
                ref = cur
            else:
                diff += sum(cur[i] != ref[i] for i in range(n))
        vs = sum(strip(eng.contigs(i)) != want[i] for i in sample)
        note(diff == 0 and vs == 0, "%s %d small regions at 1 %% noise, wg %d, %d runs (%.1f s): %d region-runs differ from run 0, %d of %d sampled regions differ from the oracle"
             % (tag, n, wg, runs, time.time() - t0, diff, vs, len(sample)))
        eng.close()


def caps(tag):
    deep = synth.make_region(21, depth=6000, W=800, var_len=1.0)
    longc = synth.make_region(22, sv_type="ins", sv_size=5000, W=1200, n_reads=1600)
    plain = synth.make_region(3, depth=60, W=1500)
    regions = [deep, plain, longc]
    want = memo("caps", lambda: [bo.assemble_region(r.read_strs(), [r.window_str], 31, 2)[0] for r in regions])
    for wg in (256, 512):
        eng = engine(31, wg_threads=wg)
        submit(eng, regions)
        eng.run(hb.BK_STAGE_ALL, sync=False)
        nf = eng.sync()
        wrong = sum(strip(eng.contigs(i)) != want[i] for i in range(3))
        note(nf == 0 and wrong == 0 and eng.stat(26) == 2, "%s cap overflows, wg %d: %d failed, %d of 3 regions differ from the oracle, %d re-run under larger caps" % (tag, wg, nf, wrong, eng.stat(26)))
        eng.close()
        off = engine(31, wg_threads=wg, no_escalation=1)
        submit(off, regions)
        off.run(hb.BK_STAGE_ALL, sync=False)
        nf = off.sync()
        note(nf == 2 and off.region_status(0)[0] == 4 and off.region_status(2)[0] == 3 and strip(off.contigs(1)) == want[1],
             "%s cap overflows without the re-run, wg %d: %d regions fail with their own status, the neighbour is untouched" % (tag, wg, nf))
        off.close()


def redo(tag):
    regions = [synth.make_region(7700, sv_type="trl", depth=120, W=2400), synth.make_region(7701, sv_type="ins", sv_size=1500, W=1500, depth=100),
               synth.make_region(7702, sv_type="del", depth=150, W=2000, L=250, noise=0.03), synth.make_region(7703, sv_type="inv", depth=100, W=1800, noise=0.01)]
    want = memo("redo", lambda: [bo.assemble_region(r.read_strs(), [r.window_str], 31, 2)[0] for r in regions])
    assert max(len(c["seq"]) for w in want for c in w) > 700, "the check needs long contigs"
    for wg in (256, 512):
        base_redos = None
        for flags in (0, 32768):
            eng = engine(31, wg_threads=wg, flags=flags)
            submit(eng, regions)
            eng.run(hb.BK_STAGE_KMER | hb.BK_STAGE_ASSEMBLE)
            wrong = sum(strip(eng.contigs(i)) != want[i] for i in range(len(regions)))
            redos = eng.stat(31)
            if flags == 0:
                base_redos = redos
            note(wrong == 0 and (flags == 0 or (redos > 4 * base_redos + 100 and redos > 1000)), "%s long-contig rounds, wg %d, flags %d: %d of %d regions differ from the oracle; reads swept again in full %d (of %d through the sweep)"
                 % (tag, wg, flags, wrong, len(regions), redos, eng.stat(30)))
            eng.close()


def noisy(tag, n):
    regions = [synth.make_region(50000 + i, depth=500, L=150, sv_type=("del", "ins", "inv")[i % 3], noise=0.005) for i in range(n)]
    for wg in (256, 512):
        one = engine(31, wg_threads=wg, flags=128)
        many = engine(31, wg_threads=wg)
        for e in (one, many):
            submit(e, regions)
            e.run(hb.BK_STAGE_ALL)
        wrong = 0
        for i in range(n):
            a, b = one.contigs(i), many.contigs(i)
            wrong += a != b
            if a == b:
                wrong += any(one.hits(i, ci) != many.hits(i, ci) for ci in range(0, len(a), 97))
        note(wrong == 0 and many.stat(28) > 0, "%s %d full-size regions at 0.5 %% noise, wg %d: %d differ between split and one unit; split regions %d, repair passes %d, assembler %.1f vs %.1f ms"
             % (tag, n, wg, wrong, many.stat(28), many.stat(27), many.kernel_ms(2), one.kernel_ms(2)))
        one.close(); many.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="")
    ap.add_argument("--seeds", default="1")
    ap.add_argument("what", nargs="+")
    a = ap.parse_args()
    if a.variant:
        hb.load_library(build.lib_path(a.variant))
    seeds = [int(s) for s in a.seeds.split(",")] if "jit" in a.variant else [0]
    for seed in seeds:
        os.environ["BK_JITTER_SEED"] = str(seed)               # read by bk_create of a jitter build (a diagnostic build: the product reads no environment)
        tag = "[%s seed %d]" % (a.variant or "product", seed)
        for w in a.what:
            try:
                if w == "g3":
                    g3(tag)
                elif w == "mixed":
                    mixed(tag)
                elif w.startswith("shape:"):
                    f = w.split(":")
                    shape(tag, int(f[1]), int(f[2]) if len(f) > 2 else 6)
                elif w.startswith("noisy:"):
                    noisy(tag, int(w.split(":")[1]))
                elif w == "caps":
                    caps(tag)
                elif w == "redo":
                    redo(tag)
                else:
                    raise SystemExit("unknown check " + w)
            except hb.BreakmerHipError as e:
                note(False, "%s %s: %s" % (tag, w, e))
                print("RACE CHECK RESULT: FAILED", flush=True)       # the device context may be gone: stop here
                sys.exit(1)
    print("RACE CHECK RESULT: %s" % ("ok" if bad == 0 else "FAILED"), flush=True)
    sys.exit(0 if bad == 0 else 1)


if __name__ == "__main__":
    main()

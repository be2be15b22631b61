"""Sustained throughput of NOISY batches with several batches in flight (one handle = one HIP stream each), the way bench.py times the
headline: a single noisy batch is bound by its slowest region's serial chain (one region in 256 takes 224 ms of the 247 ms, round 6),
batches in flight fill the chip meanwhile.  python tools/probes/noisy_inflight.py <regions per batch> <noise> <steps>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _variant
_variant.use()
import bench
from breakmer_amd import hip_backend as hb, synth

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.005
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    kind = os.environ.get("BK_PROBE_KIND", "noisy")          # noisy | cfg3 | cfg4 (bench.py's side configurations)
    k = bench.SIDE_K[kind]
    regions = bench.make_regions_parallel(kind, n) if kind != "noisy" or noise == 0.005 else [synth.make_region(50000 + i, depth=500, L=150, sv_type="del", noise=noise) for i in range(n)]
    ins = [hb.RegionInput(r.reads, r.window, read_lens=r.read_lens, partners=[p_[4] for p_ in r.partners]) for r in regions]
    ctx = bench.call_context_text(regions, bench.default_opts())
    ref = None
    flags = int(os.environ.get("BK_PROBE_FLAGS", "0"))
    for wg in [int(x) for x in os.environ.get("BK_PROBE_WGS", "0,256").split(",")]:
        for nh in [int(x) for x in os.environ.get("BK_PROBE_HANDLES", "1,2,3,4").split(",")]:
            engs = []
            for _ in range(nh):
                e = hb.Engine(kmer_size=k, rc_thresh=2, wg_threads=wg, flags=flags)
                e.submit(ins); e.set_call_context(ctx); e.run(hb.BK_STAGE_ALL); raw = e.call_blob()
                engs.append(e)
            if ref is None:
                ref = raw
            assert raw == ref
            def run_steps(k):
                launched = 0; raws = []
                for j in range(min(nh, k)):
                    engs[j].run(hb.BK_STAGE_ALL, sync=False); launched += 1
                for s in range(k):
                    e = engs[s % nh]
                    e.fetch()
                    if launched < k:
                        e.run(hb.BK_STAGE_ALL, sync=False); launched += 1
                    raws.append(e.call_blob())
                return raws
            run_steps(nh)
            t0 = time.perf_counter(); raws = run_steps(steps); dt = time.perf_counter() - t0
            ok = all(r == ref for r in raws)
            print(kind, "flags %d regions/batch %d noise %g wg %d handles %d: %.1f ms per batch, %.0f regions/s, rows identical %s" % (flags, n, noise, wg, nh, dt / steps * 1e3, n * steps / dt, ok), flush=True)
            for e in engs:
                e.close()



if __name__ == "__main__":
    main()

"""N4 at size: refseq.GenomeIndex (every 8th 16-mer of the genome, sorted) on a synthetic genome of --mb megabases: build time and
peak RSS, then the look-up of planted contig segments through the host path (numpy.searchsorted) and, with --device, through the
device path (hip_backend.DeviceIndex: the sorted codes in HBM, binary search per query k-mer in bk_index_probe_kernel) -- same loci
required.  Writes a FASTA of --mb MB under --dir (default /tmp).
    python tools/n4_at_size.py --mb 200 [--device 0]"""
import argparse, os, resource, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from breakmer_amd import refseq

ap = argparse.ArgumentParser()
ap.add_argument("--mb", type=int, default=200)
ap.add_argument("--chroms", type=int, default=8)
ap.add_argument("--dir", default="/tmp")
ap.add_argument("--device", type=int, default=-1)
ap.add_argument("--queries", type=int, default=2000)
a = ap.parse_args()
fn = os.path.join(a.dir, "synth_genome_%dmb.fa" % a.mb)
rng = np.random.default_rng(7)
per = a.mb * 1000000 // a.chroms
t0 = time.time()
if not os.path.isfile(fn):
    with open(fn, "wb") as f:
        for c in range(a.chroms):
            f.write(b">chr%d\n" % (c + 1))
            seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, per, dtype=np.uint8)]
            seq[per // 3:per // 3 + 5000] = ord("N")                 # an assembly gap
            rows = seq[:per - per % 60].reshape(-1, 60)
            out = np.empty((rows.shape[0], 61), dtype=np.uint8); out[:, :60] = rows; out[:, 60] = 10
            f.write(out.tobytes()); f.write(seq[per - per % 60:].tobytes() + b"\n")
print("genome: %d chromosomes x %d bases written in %.1f s (%s)" % (a.chroms, per, time.time() - t0, fn), flush=True)
for f_ in (fn + ".fai",):
    pass
fa = refseq.FastaIndex(fn)
t0 = time.time()
gi = refseq.GenomeIndex(fa, cache=False, device=None)
bt = time.time() - t0
rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0
print("index: %d entries (k=%d, step=%d), %.1f MB of arrays, built in %.1f s, peak RSS %.0f MB" % (len(gi.code), gi.k, gi.step, (gi.code.nbytes + gi.seqno.nbytes + gi.pos.nbytes) / 1e6, bt, rss), flush=True)
# planted queries: 60-base segments cut from random places (either strand), as the uncovered half of a translocation contig
qs = []
for i in range(a.queries):
    c = int(rng.integers(0, a.chroms)); p = int(rng.integers(10000, per - 10000))
    if per // 3 - 100 < p < per // 3 + 5100:
        p += 10000
    s = fa.fetch("chr%d" % (c + 1), p, p + 60)
    qs.append((c, p, s if i % 2 == 0 else refseq.revcomp(s)))
t0 = time.time()
host = [gi.find(s) for _c, _p, s in qs]
ht = time.time() - t0
ok = sum(1 for (c, p, s), h in zip(qs, host) if h and h[0][1] == "chr%d" % (c + 1) and h[0][3] >= p and h[0][4] <= p + 60)
print("host look-up: %d segments in %.2f s (%.2f ms each), located %d" % (len(qs), ht, 1e3 * ht / len(qs), ok), flush=True)
if a.device >= 0:
    gd = refseq.GenomeIndex.__new__(refseq.GenomeIndex)
    gd.__dict__.update(gi.__dict__); gd._dev = None; gd._dev_no = a.device; gd.probe_ms = 0.0
    t0 = time.time()
    gd.find(qs[0][2])
    print("device index resident in HBM after %.2f s (%.0f MB)" % (time.time() - t0, gi.code.nbytes / 1e6), flush=True)
    gd.probe_ms = 0.0
    t0 = time.time()
    dev = [gd.find(s) for _c, _p, s in qs]
    dt = time.time() - t0
    print("device look-up: %d segments in %.2f s (%.2f ms each; probe kernels %.2f ms in total), identical loci: %s" % (len(qs), dt, 1e3 * dt / len(qs), gd.probe_ms, dev == host), flush=True)
    # all query k-mers of all segments in ONE probe (what a batched second pass would send)
    codes = np.concatenate([np.concatenate([gi._codes(s)[0], gi._codes(refseq.revcomp(s))[0]]) for _c, _p, s in qs])
    t0 = time.time(); lo, hi = gd._dev.probe(codes); bt2 = time.time() - t0
    t0 = time.time(); lo2 = np.searchsorted(gi.code, codes, side="left"); hi2 = np.searchsorted(gi.code, codes, side="right"); nt = time.time() - t0
    print("one batched probe of %d k-mers: device %.2f ms wall (kernel %.3f ms = %.1f M look-ups/s), numpy.searchsorted %.1f ms; same ranges: %s"
          % (len(codes), bt2 * 1e3, gd._dev.kernel_ms, len(codes) / gd._dev.kernel_ms / 1e3, nt * 1e3, bool((lo == lo2).all() and (hi == hi2).all())), flush=True)

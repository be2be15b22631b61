"""runner.run() (batch lane) over the same 8,192 synthetic targets several times in one process: wall time of each run and the time the
driver's thread spends inside each library call (where a slow run waits)
    python tools/probes/runner_repeat_probe.py [runs]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: F401,E402
from breakmer_amd import hip_backend as hb, synth, sv_processor as sp  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
regions = [synth.make_region(i, depth=500, L=150, sv_type="del") for i in range(512)]
packed = {id(r): hb.pack_reads(r.reads, r.read_lens) for r in regions}
d = tempfile.mkdtemp()
bed, genes, data = [], ["header"], {}
for c in range(16):
    for r in regions:
        name = r.name + ("C%d" % c if c else "")
        bed.append("\t".join([r.chrom, str(r.start), str(r.end), name, "exon"]))
        genes.append("\t".join(["0", name, "chr" + r.chrom, "+", str(r.start), str(r.end)] + ["x"] * 6 + [name]))
        data[name.upper()] = sp.RegionData(r.read_ids, None, None, None, r.window_str, [], r.disc_reads, read_codes=r.reads, read_lens=r.read_lens, read_packed=packed[id(r)])
open(os.path.join(d, "t.bed"), "w").write("\n".join(bed) + "\n")
open(os.path.join(d, "g.txt"), "w").write("\n".join(genes) + "\n")
cfg = {"analysis_name": "x", "targets_bed_file": os.path.join(d, "t.bed"), "gene_annotation_file": os.path.join(d, "g.txt"), "kmer_size": "31", "keep_repeat_regions": True, "batch_regions": 256}
tm = {}
for name in ("submit_packed", "run", "set_call_context", "call_async", "sync", "call", "contig_counts", "stat", "__init__", "trim", "close"):
    f = getattr(hb.Engine, name)
    def make(f, name):
        def g(*a, **k):
            t = time.perf_counter()
            try:
                return f(*a, **k)
            finally:
                tm[name] = tm.get(name, 0.0) + time.perf_counter() - t
        return g
    setattr(hb.Engine, name, make(f, name))
for rep in range(runs):
    tm.clear()
    t0 = time.perf_counter(); run = sp.runner(cfg, region_data=data); rows = run.run(); dt = time.perf_counter() - t0
    print("run %d: %.3f s = %.1f k regions/s, %d rows; driver thread inside library calls (ms): %s; pool %d" % (rep, dt, len(data) / dt / 1e3, len(rows),
          ", ".join("%s %.1f" % (k, v * 1e3) for k, v in sorted(tm.items(), key=lambda kv: -kv[1])), sum(len(v) for v in hb._POOL.values())), flush=True)

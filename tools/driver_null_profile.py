"""cProfile of runner.run with an engine that does nothing (no GPU needed): what the Python driver itself costs per target."""
import cProfile, io, os, pstats, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from breakmer_amd import synth, sv_processor as sp

class NullEngine(object):
    batch_serial = 0
    def __init__(self): self.n = 0
    def submit(self, ins, wait=True):
        self.n = len(ins); self.batch_serial += 1
        for r in ins: r.fill(__import__("breakmer_amd.hip_backend", fromlist=["x"]).BkRegion())
    def run(self, stages, sync=True): pass
    def sync(self): pass
    def fetch(self): pass
    def stat(self, i): return 0
    def region_status(self, i): return 0, "ok"
    def set_call_context(self, text): self.ctx = len(text)
    def call(self): return {}
    def contig_count(self, r): return 1
    def contigs(self, r, lazy_kmers=False): return []
    def hits(self, r, c): return []
    def close(self): pass

regions = [synth.make_region(i, depth=500, L=150, sv_type="del") for i in range(256)]
d = tempfile.mkdtemp()
bed, genes, data = [], ["header"], {}
for c in range(8):
    for r in regions:
        name = r.name + ("C%d" % c if c else "")
        bed.append("\t".join([r.chrom, str(r.start), str(r.end), name, "exon"]))
        genes.append("\t".join(["0", name, "chr" + r.chrom, "+", str(r.start), str(r.end)] + ["x"] * 6 + [name]))
        data[name.upper()] = sp.RegionData(r.read_ids, None, None, None, r.window_str, [], r.disc_reads, read_codes=r.reads, read_lens=r.read_lens)
open(os.path.join(d, "t.bed"), "w").write("\n".join(bed) + "\n")
open(os.path.join(d, "g.txt"), "w").write("\n".join(genes) + "\n")
cfg = {"analysis_name": "x", "targets_bed_file": os.path.join(d, "t.bed"), "gene_annotation_file": os.path.join(d, "g.txt"), "kmer_size": "31", "keep_repeat_regions": True, "batch_regions": 256}
for rep in range(2):
    t0 = time.perf_counter(); sp.runner(cfg, region_data=data, engine_factory=lambda p: NullEngine()).run(); dt = time.perf_counter() - t0
    print("targets", len(data), "seconds %.3f" % dt, "us per target %.1f" % (dt / len(data) * 1e6))
pr = cProfile.Profile(); pr.enable()
sp.runner(cfg, region_data=data, engine_factory=lambda p: NullEngine()).run()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])

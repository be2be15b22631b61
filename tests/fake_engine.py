"""TEST INFRASTRUCTURE: a stand-in for hip_backend.Engine built on the CPU oracle, so the host logic
(sv_processor.runner/target/contig, collate) can be exercised without a GPU.  Never shipped: the
product's runner creates hip_backend.Engine, which fails loudly without the HIP library/GPU."""
import numpy as np

from oracle import bk_oracle as bo


class FakeEngine(object):
    def __init__(self, kmer_size, rc_thresh=2):
        self.k, self.rc = kmer_size, rc_thresh
        self.regions = []
        self.out = []

    def submit(self, ins):
        self.regions = []
        for g in ins:
            if getattr(g, "packed", False):                    # 2-bit rows + N list (hip_backend.pack_reads) back to a code matrix
                w = g.reads
                codes = ((w[:, :, None] >> (30 - 2 * np.arange(16, dtype=np.uint32))[None, None, :]) & 3).astype(np.uint8).reshape(w.shape[0], -1)
                for e in (g.read_n if g.read_n is not None else []):
                    codes[int(e) >> 10, int(e) & 1023] = 4
                asc = np.frombuffer(b"ACGTN", dtype=np.uint8)[codes]
            else:
                asc = np.frombuffer(b"ACGTN", dtype=np.uint8)[g.reads] if getattr(g, "codes", False) and g.reads.size else g.reads
            reads = [bytes(asc[i, :g.lens[i]]).decode() for i in range(g.reads.shape[0])]
            io = g.indel_only if g.indel_only is not None else np.zeros(len(reads), dtype=np.uint8)
            sc = None if g.sc is None else [bytes(g.sc[i, :g.sc_lens[i]]).decode() for i in range(g.sc.shape[0])]
            self.regions.append((reads, io, sc, g.window.decode(), [p.decode() for p in g.partners]))

    def run(self, stages=7, sync=True):
        self.out = []
        for reads, io, sc, window, partners in self.regions:
            contigs, info = bo.assemble_region(reads, [window], self.k, self.rc, indel_only=io, sc_seqs=sc)
            order = sorted(range(len(info["mers"])), key=lambda i: (int(info["counts"][i]), info["mers"][i]), reverse=True)
            hits = [bo.realign(c["seq"], [window] + partners) for c in contigs]
            self.out.append((contigs, [info["mers"][i] for i in order], np.array([info["counts"][i] for i in order]), len(info["rep"]), hits))

    def region_status(self, r):
        return 0, "ok"

    def kmers(self, r):
        return self.out[r][1], self.out[r][2], self.out[r][3]

    def contigs(self, r):
        return self.out[r][0]

    def hits(self, r, ci):
        return self.out[r][4][ci]

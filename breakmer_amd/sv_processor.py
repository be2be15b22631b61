"""Run orchestration and the per-region plugin surface, re-stated for the batched GPU path.

Keeps the reference's surface (sv_processor.py:98-235 `runner`, :244-722 `target`, :730-866 `contig`;
utils.py:535-674 `params`, :727-773 `anno`): the same config keys, the same method sequence
`set_ref_data -> extract_bam_reads -> clean_reads -> compare_kmers -> resolve_sv -> get_summary ->
write_results`, the same state handed between them (`cleaned_read_recs`, `read_len`,
`kmers['clusters']`, `results`, `disc_reads`, `repeat_mask`) and the same output files.

What differs, by design: the reference runs jellyfish / the Python assembler / BLAT once per region;
here `runner.run` gathers every region first and makes ONE batched call into libbreakmer_hip.so
(grouping + k-mer selection + assembly + realignment on the GPU), after which `compare_kmers()` and
`resolve_sv()` of each target only pick up their region's records.  Reads enter from `sample_bam_file`
(read_extraction.py), from the files the reference itself writes at that point (`<name>_sv_reads.fastq`,
`<name>_sv_sc_seqs.fa`, `<name>_forward_refseq.fa`) or as in-memory `RegionData`; adapter trimming (cutadapt)
is out of scope.
"""
from __future__ import annotations

import logging
import os
import shutil
from collections import OrderedDict

from . import call_context as cc, hip_backend, read_extraction, refseq, samio, sv_assembly, sv_caller      # (hip_backend: importing is harmless without the built library)

HEADER_FIELDS = ['genes', 'target_breakpoints', 'align_cigar', 'mismatches', 'strands', 'rep_overlap_segment_len', 'sv_type',
                 'split_read_count', 'nkmers', 'disc_read_count', 'breakpoint_coverages', 'contig_id', 'contig_seq']


# ------------------------------------------------------------------------------------------------ inputs
class RegionData(object):
    """Everything the hot path needs for one target, as the reference has it after clean_reads():
    reads (FASTQ order: id, seq, qual, indel_only), soft-clip sequences (None => case_sc := case),
    the forward window, extra windows with genome coordinates, discordant-pair evidence."""

    def __init__(self, read_ids, read_seqs, indel_only=None, sc_seqs=None, window="", partners=(), disc_reads=None, quals=None,
                 read_codes=None, read_lens=None, read_packed=None):
        # read_codes / read_lens: optional uint8 code matrix [N, L] (0..3 = ACGT, 4 = N) + lengths of the same reads: handed to
        # the library as is (no per-read Python work); read_seqs may then be a lazy sequence of the strings
        # read_packed: optional (words, lens, N list) of the SAME reads as hip_backend.pack_reads makes them (2 bit/base): what goes to
        # the library then (BK_SUBMIT_PACKED: a quarter of the bytes, no packing on the submit path)
        self.read_codes, self.read_lens, self.read_packed = read_codes, read_lens, read_packed
        self.read_ids = read_ids if read_codes is not None else list(read_ids)
        self.read_seqs = read_seqs if read_codes is not None else list(read_seqs)
        self._no_indel_only = indel_only is None and read_codes is not None     # (no read is indel-only: the zero flags are only made when someone looks at them)
        self._indel_only = (indel_only if read_codes is not None else list(indel_only)) if indel_only is not None else ([False] * len(self.read_ids) if read_codes is None else None)
        self.quals = list(quals) if quals is not None else None
        self.sc_seqs = sc_seqs
        self.window = window
        self.partners = list(partners)          # (chrom, start, end, name, seq)
        self.disc_reads = disc_reads or {"disc": {}, "inv": [], "td": [], "other": []}
        self._view = None                       # (the objects the view was made from ..., hip_backend.RegionInput)
        self._refcheck = None                   # (window, number of partners, verdict of target.unsupported_reference)
        self._maxlen = None                     # (read_lens, its maximum)
        self._wbytes = None                     # (window, its bytes)

    @property
    def indel_only(self):
        if self._indel_only is None:
            self._indel_only = _np.zeros(len(self.read_ids), dtype=_np.uint8)
        return self._indel_only

    @indel_only.setter
    def indel_only(self, v):
        self._indel_only, self._no_indel_only = v, False

    def window_bytes(self):
        c = self._wbytes
        if c is None or c[0] is not self.window:
            c = self._wbytes = (self.window, self.window.encode())
        return c[1]

    def checked_at_submit(self):
        """True: the characters of this target's window are checked with the whole batch's (runner._submit_batch: one C-speed scan
        over the concatenated windows) instead of per target -- the targets that go to the library as packed_item()s"""
        return type(self.read_packed) is hip_backend.PackedReads and self.sc_seqs is None and not self.partners and type(self.window) is str

    def packed_item(self):
        """(PackedReads, window bytes, indel_only array or None) for hip_backend.Engine.submit_packed, or None when this target
        needs the general path (no packed reads, soft-clip sequences, partner windows, a window that is not text)"""
        pr = self.read_packed
        if type(pr) is not hip_backend.PackedReads or self.sc_seqs is not None or self.partners or type(self.window) is not str:
            return None
        io = None if self._no_indel_only else hip_backend._as_c(self.indel_only, _np.uint8)
        return (pr, self.window_bytes(), io)

    def device_view(self, use_packed=True):
        """the hip_backend.RegionInput of these inputs.  Made once per state of the inputs and kept with them: a driver that
        runs the same RegionData again (several analyses over one set of extracted reads) does not rebuild it per run.  The
        objects it was made from are kept and compared by identity, so any replacement of an input makes a new view."""
        reads = self.read_codes if self.read_codes is not None else self.read_seqs
        packed = self.read_packed if use_packed else None
        v = self._view
        if v is not None and v[8].packed != (packed is not None):
            v = None
        if v is not None and v[0] is reads and v[1] is self.read_lens and v[2] is self.indel_only and v[3] is self.sc_seqs and v[4] is self.window and v[5] is self.partners and v[6] == len(self.partners) \
                and (self.read_codes is not None or v[7] == len(reads)):
            return v[8]
        io = self.indel_only if type(self.indel_only) is _np.ndarray else _np.asarray(self.indel_only, dtype=_np.uint8)
        ri = hip_backend.RegionInput(reads, self.window, read_lens=self.read_lens, indel_only=io, sc_seqs=self.sc_seqs, partners=[p[4] for p in self.partners], packed=packed)
        self._view = (reads, self.read_lens, self.indel_only, self.sc_seqs, self.window, self.partners, len(self.partners), len(reads), ri)
        return ri

    def max_read_len(self):
        pr = self.read_packed
        if pr is not None and hasattr(pr, "maxlen") and pr[1] is self.read_lens:
            return pr.maxlen
        m = self._maxlen
        if m is None or m[0] is not self.read_lens:
            m = self._maxlen = (self.read_lens, int(_np.asarray(self.read_lens).max()) if len(self.read_lens) else 0)
        return m[1]


def read_fasta_first(fn):
    """the sequence of a one-record FASTA AS WRITTEN: a *_refseq.fa written by the reference from a soft-masked genome keeps
    lower case (utils.py:366-371, str(seq)).  Jellyfish does not care about case; BLAT is run with -repeats=lower
    (sv_processor.py:843) and reports matches on lower-case target bases as repMatches, which the caller reads
    (sv_caller.py:913, 975-986): the library takes the window with its case (include/breakmer_hip.h)."""
    seq = []
    with open(fn) as f:
        for ln in f:
            if not ln.startswith(">"):
                seq.append(ln.strip())
    return "".join(seq)


def read_fastq(fn):
    """FastqFile (utils.py:693-720): 4-line records; the trailing _<0|1> of the id is the indel_only flag
    (utils.py:211-213)."""
    ids, seqs, quals, io = [], [], [], []
    with open(fn) as f:
        while True:
            h = f.readline()
            if not h:
                break
            s, _p, q = f.readline(), f.readline(), f.readline()
            h = h.strip()
            ids.append(h)
            seqs.append(s.strip())
            quals.append(q.strip())
            io.append(h.lstrip("@").split("_")[-1] == "1")
    return ids, seqs, quals, io


class _LazyReads(object):
    """fq_read objects (utils.py:681-688) of a code-matrix region, made when asked for (a contig names ~150 of 10,000)"""

    def __init__(self, d):
        self.d, self.cache = d, {}

    def __len__(self):
        return len(self.d.read_ids)

    def __getitem__(self, i):
        r = self.cache.get(i)
        if r is None:
            d = self.d
            n = int(d.read_lens[i])
            seq = bytes(_CODES[d.read_codes[i, :n]]).decode()
            r = sv_assembly.fq_read(d.read_ids[i], seq, d.quals[i] if d.quals else "I" * n, bool(d.indel_only[i]))
            self.cache[i] = r
        return r

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class _LazyRecs(object):
    """cleaned_read_recs (dict seq -> [fq_read], utils.py:239-244) of a code-matrix region: built when someone iterates it"""

    def __init__(self, reads):
        self.reads, self._d = reads, None

    def _build(self):
        if self._d is None:
            self._d = OrderedDict()
            for r in self.reads:
                self._d.setdefault(r.seq, []).append(r)
        return self._d

    def __len__(self):
        return len(self._build())

    def __iter__(self):
        return iter(self._build())

    def __getitem__(self, k):
        return self._build()[k]

    def items(self):
        return self._build().items()


import numpy as _np  # noqa: E402
_CODES = _np.frombuffer(b"ACGTN", dtype=_np.uint8)


# ------------------------------------------------------------------------------------------------ params / anno
class anno(object):                                                 # utils.py:727-773
    def __init__(self):
        self.genes = OrderedDict()

    def add_genes(self, gene_fn):
        with open(gene_fn) as f:
            lines = f.readlines()
        for ln in lines[1:]:
            p = ln.strip().split()
            chrom, start, end, gid = p[2], int(p[4]), int(p[5]), p[12]
            if gid in self.genes:
                if start <= self.genes[gid][1] and end >= self.genes[gid][2]:
                    self.genes[gid] = [chrom, start, end]
            else:
                self.genes[gid] = [chrom, start, end]

    def add_regions(self, bed_fn):
        with open(bed_fn) as f:
            for ln in f:
                if not ln.strip():
                    continue
                chrom, start, end, name = ln.split()[:4]
                if name not in self.genes:
                    self.genes[name] = [chrom, int(start), int(end)]

    def set_gene(self, chrom, pos):                                  # utils.py:756-773 (table order = insertion order, P4)
        found = []
        if str(chrom).find('chr') == -1:
            chrom = 'chr' + str(chrom)
        for g, (gchrom, gs, ge) in self.genes.items():
            if chrom == gchrom:
                if len(pos) == 1:
                    if gs <= int(pos[0]) <= ge:
                        found.append(g)
                        break
                elif gs <= int(pos[0]) <= ge or gs <= int(pos[1]) <= ge:
                    found.append(g)
        return ",".join(found) if found else 'intergenic'


class params(object):                                               # utils.py:535-674
    DEFAULTS = {'indel_size': 15, 'trl_sr_thresh': 2, 'indel_sr_thresh': 5, 'rearr_sr_thresh': 3, 'rearr_minseg_len': 30,
                'trl_minseg_len': 25, 'keep_intron_vars': False, 'keep_repeat_regions': False, 'var_filter': 'all',
                'no_output_header': False, 'gene_list': None, 'preset_ref_data': False, 'sample_bam_file': None}

    def __init__(self, config_d):
        self.opts = dict(self.DEFAULTS)
        self.opts.update(config_d)
        self.gene_annotations = anno()
        self.targets = {}
        self.paths = {}
        self.logger = logging.getLogger('root')
        self.repeat_mask = None
        self._bams = {}
        self._fasta = None
        self.set_params()

    def set_params(self):                                            # utils.py:574-618
        vf = self.opts['var_filter']
        if vf == 'all':
            self.opts['var_filter'] = ['indel', 'rearrangement', 'trl']
        elif isinstance(vf, str):
            vf = vf.split(",")
            self.opts['var_filter'] = vf if any(x in vf for x in ('indel', 'rearrangement', 'trl')) else ['indel', 'rearrangement', 'trl']
        if self.opts.get('targets_bed_file'):
            self.set_targets(self.opts.get('gene_list'))
        if self.opts.get('gene_annotation_file'):
            self.gene_annotations.add_genes(self.opts['gene_annotation_file'])
        if self.opts.get('other_regions_file'):
            self.gene_annotations.add_regions(self.opts['other_regions_file'])
        if self.opts.get('analysis_dir'):
            self.paths['analysis'] = os.path.abspath(os.path.normpath(self.opts['analysis_dir']))
            self.paths['output'] = os.path.join(self.paths['analysis'], 'output')
            self.paths['targets'] = (os.path.abspath(os.path.normpath(self.opts['targets_dir'])) if 'targets_dir' in self.opts
                                     else os.path.join(self.paths['analysis'], 'targets'))
            if self.opts.get('reference_data_dir'):
                self.paths['ref_data'] = os.path.abspath(os.path.normpath(self.opts['reference_data_dir']))
            for p in self.paths.values():
                os.makedirs(p, exist_ok=True)
        if not self.opts['keep_repeat_regions'] and self.opts.get('repeat_mask_file'):
            self.repeat_mask = setup_rmask_all(self.opts['repeat_mask_file'])

    def set_targets(self, gene_list):                                # utils.py:545-572
        wanted = None
        if gene_list:
            with open(gene_list) as f:
                wanted = [ln.strip().upper() for ln in f]
        with open(self.opts['targets_bed_file']) as f:
            for ln in f:
                p = ln.strip().split()
                if len(p) < 4:
                    continue
                chrm, bp1, bp2, name = p[:4]
                if wanted and name.upper() not in wanted:
                    continue
                self.targets.setdefault(name.upper(), []).append((chrm, int(bp1), int(bp2), name, p[4] if len(p) > 4 else None))

    def open_fasta(self):                                            # the genome reference (`reference_fasta`), indexed once
        fn = self.opts.get('reference_fasta')
        if not fn or not os.path.isfile(fn):
            return None
        if self._fasta is None:
            self._fasta = refseq.FastaIndex(fn)
        return self._fasta

    def genome_index(self, device=None):
        """sampled k-mer index of the whole `reference_fasta` (refseq.GenomeIndex), built on first use (or read from its cache file
        next to the FASTA): the genome-wide part of the realignment (N4) for contigs that neither the target window nor a
        discordant pair explains.  device: the GPU whose HBM holds the sorted codes for the look-ups (None: host numpy)"""
        if getattr(self, '_gindex', None) is None:
            fa = self.open_fasta()
            if fa is None:
                return None
            self.logger.info('indexing %s for the genome-wide realignment of unexplained contigs' % self.opts.get('reference_fasta'))
            self._gindex = refseq.GenomeIndex(fa, device=device)
        return self._gindex

    def open_bam(self, fn):                                          # one parse per alignment file, shared by all targets
        if fn not in self._bams:
            # only the records the targets can use are kept: their [start-200, end+200) windows (sv_processor.py:431) and the mates
            regions = []
            for ivs in self.targets.values():
                regions.append((ivs[0][0], min(int(v[1]) for v in ivs) - 200, max(int(v[2]) for v in ivs) + 200))
            self._bams[fn] = samio.Samfile(fn, regions=regions or None)
        return self._bams[fn]

    def get_kmer_size(self): return int(self.opts['kmer_size'])
    def get_min_segment_length(self, kind): return int(self.opts[kind + '_minseg_len'])

    def get_sr_thresh(self, kind):                                   # utils.py:665-674
        if kind == 'min':
            return min(self.get_sr_thresh('trl'), self.get_sr_thresh('rearrangement'), self.get_sr_thresh('indel'))
        return int(self.opts[{'trl': 'trl_sr_thresh', 'rearrangement': 'rearr_sr_thresh', 'indel': 'indel_sr_thresh'}[kind]])


def setup_rmask_all(fn):                                            # utils.py:302-316
    mask = {}
    with open(fn) as f:
        for ln in f:
            p = ln.strip().split("\t")
            if len(p) < 4:
                continue
            c = p[0].replace('chr', '')
            mask.setdefault(c, []).append((c, int(p[1]), int(p[2]), p[3]))
    return mask


def setup_rmask(gene_coords, ref_path, all_mask):                   # utils.py:320-353
    """Repeats of one target: entries of ITS chromosome -- compared as the reference compares, the mask's name with 'chr'
    stripped against the target's chromosome AS GIVEN (a BED with 'chr1' therefore matches nothing) -- that lie inside
    [start, end]; written to <ref_path>/<name>_rep_mask.bed the first time, read back from it afterwards (marker file)."""
    chrom, s, e, name = gene_coords[:4]
    fn = os.path.join(ref_path, name + '_rep_mask.bed') if ref_path else None
    marker = os.path.join(ref_path, "." + name + '_rep_mask.bed') if ref_path else None
    if marker and os.path.isfile(marker):
        out = []
        with open(fn) as f:
            for ln in f:
                p = ln.strip().split()
                if len(p) >= 4:
                    out.append((p[0], int(p[1]), int(p[2]), p[3]))
        return out
    out = [m for m in all_mask.get(chrom, []) if m[1] >= int(s) and m[2] <= int(e)]
    if fn:
        os.makedirs(ref_path, exist_ok=True)
        with open(fn, 'w') as f:
            for m in out:
                f.write("\t".join(str(x) for x in m) + "\n")
        open(marker, 'w').close()
    return out


# ------------------------------------------------------------------------------------------------ contig
class contig(object):
    """sv_processor.py:730-866: one assembled contig of a target: realign, call, write."""

    def __init__(self, parent_target, contig_id, assembly, hits):
        self.params = parent_target.params
        self.id = contig_id
        self.query_region = parent_target.get_values()
        self.target = parent_target
        self.reads = assembly.reads
        self.kmers = assembly.kmers
        self.contig_seq = assembly.get_contig_seq()
        self.contig_rcounts = assembly.get_contig_counts()
        self.contig_kmer_locs = assembly.get_kmer_locs()
        self.hits = hits                        # realign records of this contig (engine.hits)
        self.result = None
        self.psl_rows = None
        self.path = os.path.join(parent_target.paths['contigs'], contig_id) if 'contigs' in parent_target.paths else None
        if self.path:
            self.setup()

    def setup(self):                                                 # :747-782 cluster file, read fastq, contig fasta
        os.makedirs(self.path, exist_ok=True)
        t = self.target
        if t.files.get('kmer_clusters'):
            with open(t.files['kmer_clusters'], 'w') as f:
                f.write(self.id + " " + str(len(self.kmers)) + "\n")
                f.write(",".join(x[0] for x in self.kmers) + "\n")
                f.write(",".join(x.id for x in self.reads) + "\n\n")
        with open(os.path.join(self.path, self.id + ".fq"), 'w') as f:
            for r in self.reads:
                f.write(r.id + "\n" + r.seq + "\n+\n" + r.qual + "\n")
        with open(os.path.join(self.path, self.id + ".fa"), 'w') as f:
            f.write(">contig1" + "\n" + self.contig_seq)

    def has_result(self): return bool(self.result)

    def query_ref(self):
        """contig.query_ref + check_target_blat (:823-859): records against the target window are tried
        first in window coordinates with the (offset, tname) override; if they do not explain the
        query (`target_hit`), all windows in genome coordinates with chr names are used (Q14)."""
        t = self.target
        qr = self.query_region
        own = [h for h in self.hits if h["t_index"] == 0]
        if own:
            rows = [sv_caller.psl_fields(h, 'contig1', t.name, 0) for h in own]
            meta = {'offset': qr[1] - 200, 'tname': qr[0].replace('chr', ''), 'psl_records': rows, 'sbam': self.params.opts.get('sample_bam_file')}
            am = sv_caller.align_manager(meta)
            if am.bm.target_hit():
                self.psl_rows = [r[3].blat_values for r in am.bm.blat_results]      # the '.mod' rows (:802)
                return
        tinfo = [(qr[0], qr[1] - 200)] + [(p[0], p[1]) for p in t.partner_windows]
        self.psl_rows = [sv_caller.psl_fields(h, 'contig1', 'chr' + str(tinfo[h["t_index"]][0]).replace('chr', ''), tinfo[h["t_index"]][1], repeats_lower=False) for h in self.hits]

    def make_calls(self, disc_reads, rep_mask):                      # :863-866
        meta = {'params': self.params, 'repeat_mask': rep_mask, 'query_region': self.query_region, 'psl_records': self.psl_rows,
                'disc_reads': disc_reads, 'sbam': self.params.opts.get('sample_bam_file'), 'coverage_fn': self.target.coverage_fn,
                'contig_vals': (self.contig_seq, self.contig_rcounts, self.id, self.reads, len(self.kmers), self.contig_kmer_locs)}
        self.result = sv_caller.align_manager(meta).get_result()

    def write_result(self, output_path):                             # :791-799
        if self.result and self.path:
            fn = os.path.join(self.path, self.id + "_svs.out")
            with open(fn, 'w') as f:
                f.write("\t".join(str(x) for x in self.result))
            shutil.copyfile(fn, os.path.join(output_path, self.id + "_svs.out"))


# ------------------------------------------------------------------------------------------------ target
class target(object):                                               # sv_processor.py:244-722
    # defaults of the per-target state that starts out immutable (instances overwrite them; a driver makes tens of thousands of
    # targets per second, and every assignment in __init__ is paid per target)
    name = chrom = start = end = None
    disc_reads = None
    cleaned_read_recs = None
    read_len = 0
    repeat_mask = None
    coverage_fn = None
    region_index = None                         # slot in the batched GPU call
    native_rows = None
    engine = None
    failed = None                               # text of the device cap this target's region hit (it is skipped then)
    n_contigs = None                            # number of contigs of the region when the runner asked for all counts of the batch at once
    check_at_submit = False                     # the window's characters are checked with the whole batch's (runner._submit_batch)
    logger = logging.getLogger('root')

    def __init__(self, intervals, prm, data=None, write_files=True):
        self.params = prm
        self.paths, self.files = {}, {}
        self.kmers = {}
        self.results = []
        self.svs = {'trl': [0, '-'], 'indel': [0, ''], 'rearrangement': [0, '']}
        self.target_intervals = intervals
        self.data = data
        self.partner_windows = []
        self.write_files = write_files and 'targets' in prm.paths
        self.reads = []
        self.setup()

    def setup(self):                                                 # :267-294
        for v in self.target_intervals:
            if not self.name: self.name = v[3]
            if not self.chrom: self.chrom = v[0]
            if not self.start: self.start = int(v[1])
            if not self.end: self.end = int(v[2])
            self.start = min(self.start, int(v[1]))
            self.end = max(self.end, int(v[2]))
        if self.write_files:
            base = os.path.join(self.params.paths['targets'], self.name)
            for key, p in (('base', base), ('data', os.path.join(base, 'data')), ('contigs', os.path.join(base, 'contigs')),
                           ('kmers', os.path.join(base, 'kmers')), ('output', os.path.join(self.params.paths['output'], self.name))):
                self.paths[key] = p
                os.makedirs(p, exist_ok=True)
            if 'ref_data' in self.params.paths:
                self.paths['ref_data'] = os.path.join(self.params.paths['ref_data'], self.name)
                self.files['target_ref_fn'] = [os.path.join(self.paths['ref_data'], self.name + '_forward_refseq.fa'),
                                               os.path.join(self.paths['ref_data'], self.name + '_reverse_refseq.fa')]
            self.files['kmer_clusters'] = os.path.join(self.paths['kmers'], self.name + "_sample_kmers_merged.out")
            self.files['sample_kmers'] = os.path.join(self.paths['kmers'], self.name + "_sample_kmers.out")

    def get_values(self): return (self.chrom, self.start, self.end, self.name, self.target_intervals)
    def has_results(self): return len(self.results) > 0

    def release(self):
        """drop the per-target inputs once its rows are out (the reference discards them too, sv_processor.py:643)"""
        self.data = None
        self.reads = []
        self.engine = None
        if isinstance(self.kmers.get('clusters'), sv_assembly.LazyContigs):
            self.kmers['clusters'].detach()

    def rm_output_dir(self):
        if 'output' in self.paths and os.path.isdir(self.paths['output']):
            shutil.rmtree(self.paths['output'])

    # ---- steps before the hot path (file formats of the reference; BAM extraction itself is out of scope)
    def set_ref_data(self):                                          # :351-364
        fa = self.params.open_fasta()
        if self.data is None and fa is not None and self.files.get('target_ref_fn'):     # utils.extract_refseq_fa, forward + reverse
            for direction in ("forward", "reverse"):
                refseq.extract_refseq_fa(self.get_values(), self.paths['ref_data'], fa, direction)
        if self.data is None and self.files.get('target_ref_fn') and os.path.isfile(self.files['target_ref_fn'][0]):
            self._window = read_fasta_first(self.files['target_ref_fn'][0])
        elif self.data is not None:
            self._window = self.data.window
        else:
            raise RuntimeError("target %s: no reference window (expected %s)" % (self.name, self.files.get('target_ref_fn')))
        if self.params.repeat_mask is not None:                     # sv_processor.py:353-355 -> utils.setup_rmask
            self.repeat_mask = setup_rmask(self.get_values(), self.paths.get('ref_data'), self.params.repeat_mask)

    def extract_bam_reads(self):                                     # :422-540
        bam_fn = self.params.opts.get('sample_bam_file')
        if self.data is None and bam_fn and os.path.isfile(bam_fn):
            # N2: select the evidence reads from the alignment file itself (read_extraction.py, pinned by G6)
            bam = self.params.open_bam(bam_fn)
            self.coverage_fn = sv_caller.bam_coverage_fn(bam)       # sv_caller.py:99-133
            self.sv_reads, fq_text, fa_text, disc = read_extraction.extract_reads(bam, self.chrom, self.start, self.end,
                                                                                  self.params.get_kmer_size())
            d = self.paths.get('data')
            if d:
                self.files['sv_fq'] = os.path.join(d, self.name + "_sv_reads.fastq")
                self.files['sv_sc_unmapped_fa'] = os.path.join(d, self.name + "_sv_sc_seqs.fa")
                with open(self.files['sv_fq'], 'w') as f: f.write(fq_text)
                with open(self.files['sv_sc_unmapped_fa'], 'w') as f: f.write(fa_text)
            recs, _rl = read_extraction.get_fastq_reads(fq_text, self.sv_reads)      # clean_reads' filter (utils.py:203-246)
            sc = [ln for ln in fa_text.split("\n") if ln and not ln.startswith(">")]
            self.data = RegionData([x[0] for x in recs], [x[1] for x in recs], [x[3] for x in recs], sc, self._window,
                                   disc_reads=disc, quals=[x[2] for x in recs])
        if self.data is None:
            d = self.paths.get('data', '')
            fq = os.path.join(d, self.name + "_sv_reads.fastq")
            if not os.path.isfile(fq):
                self.data = RegionData([], [], window=self._window)
                return
            ids, seqs, quals, io = read_fastq(fq)
            sc = None
            scfn = os.path.join(d, self.name + "_sv_sc_seqs.fa")
            if os.path.isfile(scfn):
                sc = [ln.strip() for ln in open(scfn) if ln.strip() and not ln.startswith(">")]
            self.data = RegionData(ids, seqs, io, sc, self._window, quals=quals)
        self.disc_reads = self.data.disc_reads
        if not self.data.partners and self.params.open_fasta() is not None and self.disc_reads.get('disc'):
            # N4: candidate partner windows from the discordant pairs (refseq.discover_partners) stand in for the whole-genome search
            skipped = []
            self.data.partners = refseq.discover_partners(self.disc_reads['disc'], self.params.open_fasta(), self.params.gene_annotations,
                                                          self.chrom, self.start, self.end, min_pairs=self.params.get_sr_thresh('trl'), skipped=skipped)
            for c_, s_, e_ in skipped:
                self.logger.warning('target %s: partner window %s:%d-%d holds characters other than A/C/G/T/N and is not realigned against' % (self.name, c_, s_, e_))
            if self.data.partners:
                self.logger.info('target %s: %d partner window(s) from discordant pairs: %s' % (self.name, len(self.data.partners),
                                 ", ".join("%s:%d-%d" % (p[0], p[1], p[2]) for p in self.data.partners)))
        self.partner_windows = self.data.partners

    def cost_estimate(self):
        """what this target will roughly cost on the device, known on EVERY rank without communication (the ranks deal the
        targets among themselves by it): the number of reads -- given with the region data, counted in the alignment file
        (every rank has parsed it), or the size of an extracted read file -- else the length of the interval"""
        d = self.data
        if d is not None:
            return len(d.read_ids)
        bam_fn = self.params.opts.get('sample_bam_file')
        try:
            if bam_fn and os.path.isfile(bam_fn):
                return sum(1 for _r in self.params.open_bam(bam_fn).fetch(self.chrom, self.start - 200, self.end + 200))
            fq = os.path.join(self.paths.get('data', ''), self.name + "_sv_reads.fastq")
            if os.path.isfile(fq):
                return os.path.getsize(fq) // 300
        except Exception:
            pass
        return max(1, (int(self.end) - int(self.start)) // 100)

    genome_searched = False

    def unsupported_reference(self):
        """None, or why this target cannot go to the device: its window (or a partner window) holds a character other than
        A/C/G/T/N.  (An N -- an assembly gap within 200 bp of the target -- is fine: the device carries the N positions of a
        window as it carries those of the reads; no window k-mer spans one, it matches nothing in the realignment.)  Such a
        target is skipped ALONE, with an error in the log and in runner.failed_targets."""
        d = self.data
        c = d._refcheck
        if c is not None and c[0] is d.window and c[1] is d.partners and c[2] == len(d.partners):
            return c[3]
        why = self._unsupported_reference(d)
        d._refcheck = (d.window, d.partners, len(d.partners), why)
        return why

    @staticmethod
    def _unsupported_reference(d):
        if not d.window:
            return "empty reference window"
        if isinstance(d.window, str) and d.window_bytes().translate(None, b"ACGTNacgtn"):      # C-speed scan (str.strip walks the characters one by one); lower case = soft-masked
            return "reference window holds characters other than A/C/G/T/N (%s)" % ",".join(sorted(set(d.window.strip("ACGTNacgtn")))[:5])
        for p_ in d.partners:
            if isinstance(p_[4], str) and p_[4].encode().translate(None, b"ACGTNacgtn"):
                return "partner window %s:%s-%s holds characters other than A/C/G/T/N" % (p_[0], p_[1], p_[2])
        return None

    def clean_reads(self):                                           # :584-606 (cutadapt out of scope) -> bool
        d = self.data
        # Reads with N calls are kept, as the reference keeps them (utils.py:203-246): the device path carries the N
        # positions next to the 2-bit words.  Any other character (IUPAC codes never occur in the alignment files the
        # reference is run on) cannot be represented: such reads are left out with a warning.
        bad = [] if d.read_codes is not None else [n for n, s_ in enumerate(d.read_seqs) if s_.strip("ACGTN")]
        if bad:
            self.logger.warning('target %s: %d of %d reads contain characters other than A/C/G/T/N and are skipped' % (self.name, len(bad), len(d.read_seqs)))
            drop = set(bad)
            keep = [n for n in range(len(d.read_seqs)) if n not in drop]
            d.read_ids = [d.read_ids[n] for n in keep]; d.read_seqs = [d.read_seqs[n] for n in keep]
            d.indel_only = [d.indel_only[n] for n in keep]
            if d.quals is not None: d.quals = [d.quals[n] for n in keep]
        if d.sc_seqs is not None:
            # case_sc only contributes the PRESENCE of k-mers (sv_processor.py:621) and Jellyfish skips every k-mer that
            # holds a non-ACGT character, so a soft-clip sequence split at those characters has the same k-mer set
            k = self.params.get_kmer_size()
            pieces = []
            for x in d.sc_seqs:
                if not x.strip("ACGT"):
                    pieces.append(x)
                else:
                    cur = []
                    for ch in x:
                        if ch in "ACGT":
                            cur.append(ch)
                        else:
                            if len(cur) >= k: pieces.append("".join(cur))
                            cur = []
                    if len(cur) >= k: pieces.append("".join(cur))
            d.sc_seqs = pieces
        if d.read_codes is not None:
            # code-matrix input (synthetic / pre-packed reads): nothing per read happens on the host; the fq_read objects the
            # writers and the caller need are made on demand (_LazyReads), cleaned_read_recs only exists as a count
            self.reads = _LazyReads(d)
            self.read_len = d.max_read_len()
            self.cleaned_read_recs = _LazyRecs(self.reads)
            return len(d.read_ids) > 0
        q = d.quals
        self.reads = [sv_assembly.fq_read(i, s, (q[n] if q else "I" * len(s)), bool(io)) for n, (i, s, io) in enumerate(zip(d.read_ids, d.read_seqs, d.indel_only))]
        recs = OrderedDict()
        for r in self.reads:                                         # utils.py:239-244
            recs.setdefault(r.seq, []).append(r)
            self.read_len = max(self.read_len, len(r.seq))
        self.cleaned_read_recs = recs
        return len(recs) > 0

    # ---- the hot path: results of the batched GPU call
    def compare_kmers(self):                                         # :609-645
        eng, ri = self.engine, self.region_index
        if self.files.get('sample_kmers'):                          # the k-mer set is only materialised for its file (the assembler consumed it on the device)
            mers, counts, _u = eng.kmers(ri)
            self.kmers['case_only'] = dict(zip(mers, counts.tolist()))
            with open(self.files['sample_kmers'], 'w') as f:
                for m, c in self.kmers['case_only'].items():
                    f.write("\t".join([m, str(c)]) + "\n")
        if self.native_rows is not None and not self.write_files and hasattr(eng, 'contig_count') and hasattr(eng, 'batch_serial'):
            # rows come from the native tail and no per-contig file is written: only the number of contigs is needed now
            self.kmers['clusters'] = sv_assembly.LazyContigs(eng, ri, self.reads, self.params.get_kmer_size(), self.n_contigs)
        else:
            self.kmers['clusters'] = sv_assembly.contigs_from_engine(eng, ri, self.reads, self.params.get_kmer_size())
        self.cleaned_read_recs = None
        self.kmers['case_only'] = {}

    def resolve_sv(self):                                            # :648-665
        if self.native_rows is not None:                             # rows computed by the native tail (csrc/bk_call.h) for the whole batch
            self.results = [list(r) for r in self.native_rows]
            if self.coverage_fn is not None:                         # the native tail has no alignment file: field 10 is filled here
                for r in self.results:
                    r[10] = sv_caller.brkpt_coverages(r[1], self.coverage_fn)
            if self.write_files:                                     # per-contig files (:747-799) are still written
                by_id = {r[11]: r for r in self.results}
                for n, kc in enumerate(self.kmers['clusters'], 1):
                    ctig = contig(self, 'contig' + str(n), kc, [])
                    ctig.result = by_id.get(self.name + "_contig" + str(n))
                    if ctig.has_result():
                        ctig.write_result(self.paths['output'])
            return
        for n, kc in enumerate(self.kmers['clusters'], 1):
            ctig = contig(self, 'contig' + str(n), kc, self.engine.hits(self.region_index, n - 1))
            ctig.query_ref()
            ctig.make_calls(self.disc_reads, self.repeat_mask)
            if ctig.has_result():
                if 'output' in self.paths:
                    ctig.write_result(self.paths['output'])
                self.results.append(ctig.result)

    def write_results(self):                                         # :668-683
        files = {}
        for res in self.results:
            tag = 'rearrangement' if res[6].find('rearrangement') > -1 else res[6]
            if tag not in files:
                files[tag] = open(os.path.join(self.paths['output'], self.name + "_" + tag + "_svs.out"), 'w')
                if not self.params.opts['no_output_header']:
                    files[tag].write("\t".join(HEADER_FIELDS) + "\n")
            files[tag].write("\t".join(str(x) for x in res) + "\n")
        for f in files.values():
            f.close()

    def get_sv_counts(self):                                         # :686-705 incl. the 'rearrangment' misspelling (Q11)
        total = 0
        for res in self.results:
            tag = 'rearrangement' if res[6].find('rearrangement') > -1 else res[6]
            self.svs[tag][0] += 1
            total += 1
        return total

    def get_summary(self):                                           # :708-721
        total = self.get_sv_counts() if self.results else 0
        svs = self.svs
        if len(svs) == 3:                                            # the keys the class starts with, sorted: indel, rearrangement, trl
            return _SUMMARY_HEADER, "%s\t%d\t%d\t%s\t%s\t%s\t-" % (self.name, len(self.kmers['clusters']), total, svs['indel'][0], svs['rearrangement'][0], svs['trl'][0])
        keys = sorted(svs.keys())
        header = ['Target', 'N_contigs', 'Total_variants'] + ['N_' + str(x) for x in keys] + ['Rearrangements']
        out = self.name + '\t' + str(len(self.kmers['clusters'])) + '\t' + str(total) + '\t'
        for t in keys:
            out += str(svs[t][0]) + '\t'
        out += '-'
        return "\t".join(header), out


_SUMMARY_HEADER = "\t".join(['Target', 'N_contigs', 'Total_variants', 'N_indel', 'N_rearrangement', 'N_trl', 'Rearrangements'])


# ------------------------------------------------------------------------------------------------ runner
class _gc_paused(object):
    """The driver makes tens of thousands of small, short-lived, acyclic objects per second (intervals, targets, views, rows), and
    every so many allocations the cyclic collector walks EVERYTHING the process keeps alive to find nothing: in a process that has
    torch imported (bench.py, any torch.distributed launch) that was half of the driver's time -- 15-17 k regions/s with the
    collector on, 30 k with it off, same box (tools/probes/runner_env_probe.py) -- and a full collection landing in the parse of
    the target files doubled the time of one run in four (tools/probes/runner_repeat_probe.py).  It is switched off while the
    target tables are read and for the duration of the run; what they leave behind is reference-counted like everything else."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        if self.was:
            import gc
            gc.enable()


class _Targets(object):
    """runner.targets (sv_processor.py:165-170: name -> target, every target of the run): the same mapping, its objects made when
    someone asks for one.  The batch lane of runner.run never does -- it keeps a target's outcome as (rows, number of contigs)
    -- so a run over tens of thousands of plain targets does not pay for tens of thousands of objects nobody looks at; a target
    looked at afterwards is the object the per-target sequence would have left behind (runner._make_target)."""

    def __init__(self, run):
        self._run, self._made = run, {}

    def __getitem__(self, key):
        t = self._made.get(key)
        if t is None:
            t = self._made[key] = self._run._make_target(key)       # KeyError for a name the run does not have
        return t

    def __setitem__(self, key, t):
        self._made[key] = t

    def __contains__(self, key): return key in self._run.params.targets
    def __len__(self): return len(self._run.params.targets)
    def __iter__(self): return iter(sorted(self._run.params.targets))
    def keys(self): return sorted(self._run.params.targets)
    def values(self): return [self[k] for k in self]
    def items(self): return [(k, self[k]) for k in self]
    def get(self, key, default=None): return self[key] if key in self else default


class _LaneBatch(object):
    """one batch of PLAIN targets (runner._lane_batch) on its way through the library: parallel lists instead of target objects"""
    __slots__ = ("keys", "names", "regions", "data")

    def __init__(self):
        self.keys, self.names, self.regions, self.data = [], [], [], []

    def __len__(self):
        return len(self.keys)


class runner(object):                                               # sv_processor.py:98-235
    def __init__(self, config_d, region_data=None, engine_factory=None, rank=0, world=1, collate=None, native_calls=True, status_exchange=None, batch_lane=True):
        with _gc_paused():
            self.params = params(config_d)
        self.results = []
        self.targets = _Targets(self)
        self.batch_lane = batch_lane            # plain targets go through the library as rows of a batch table (_lane_batch); False: every target as an object
        self._lane_done = {}                    # key -> (rows, number of contigs, why it failed or None) of the targets the batch lane ran
        self.summary = {}
        self.summary_header = ''
        self.logger = logging.getLogger('root')
        self.region_data = region_data or {}
        self.engine_factory = engine_factory
        self.rank, self.world, self.collate = rank, world, collate
        self.status_exchange = status_exchange  # collate.exchange_status: every rank learns whether any rank failed BEFORE the collation
        self.assigned_cost = 0                  # sum of the cost estimates of this rank's targets
        self._ctx_head = None
        self._ctx_token = object()              # which run's tables a handle holds (hip_backend's pool hands handles from run to run)
        self._pooled = []                       # handles to give back to hip_backend's pool when the run is over
        self.native_calls = native_calls        # SV-call tail in C++ (bk_call) instead of breakmer_amd/sv_caller.py; same rows
        self.engine = None
        self._retry = []                        # targets to run again with partner windows from the genome-wide search
        self.failed_targets = {}                # name -> why the target has no result although it had reads (a device cap, an N in
                                                # its window): logged as errors, listed at the end of the run, exit code 3 of breakmer.py

    def create_targets(self):                                        # :165-170
        """the sorted target names; the objects themselves are made by self.targets when asked for"""
        return sorted(self.params.targets.keys())

    def _make_target(self, key):
        """the target object of one name: before its batch has run, the object the per-target sequence starts from; after the
        batch lane ran it, the object that sequence would have left behind (results, contig count, SV counts, inputs released)"""
        done = self._lane_done.get(key)
        t = target(self.params.targets[key], self.params, self.region_data.get(key) if done is None else None)
        if done is not None:
            rows, n_contigs, why = done
            d = self.region_data.get(key)
            t.failed = why
            t.disc_reads, t.partner_windows, t.read_len = d.disc_reads, d.partners, d.max_read_len()
            if why is None:
                t.native_rows = t.results = rows
                t.n_contigs = n_contigs
                t.kmers['clusters'] = sv_assembly.LazyContigs.counted(n_contigs)
                t.kmers['case_only'] = {}
                t.get_sv_counts()
        return t

    # ---- the batch lane: plain targets as rows of a table (no per-target objects, one pass per batch for each step)
    def _lane_ok(self):
        """whether this run may use the batch lane at all: rows from the native call tail, no per-target files, nothing that needs
        a per-target look at the genome or a repeat mask (those runs spend their time elsewhere anyway)"""
        prm = self.params
        return bool(self.batch_lane and self.native_calls and not prm.paths and prm.repeat_mask is None and prm.open_fasta() is None
                    and not (prm.opts.get('sample_bam_file') and os.path.isfile(prm.opts['sample_bam_file'])))

    def _lane_batch(self, eng, keys):
        """the _LaneBatch of these targets if EVERY one of them is plain -- in-memory inputs with 2-bit packed reads
        (hip_backend.PackedReads) as a read extraction that packs as it goes hands them over, a text window, no soft-clip
        sequences, no partner windows, no object made for it yet -- and the engine takes packed batches; else None (the batch
        goes the general way, target by target).  Targets without reads are left out as the general way leaves them out."""
        if not (hasattr(eng, 'submit_packed') and hasattr(eng, 'set_call_context') and hasattr(eng, 'contig_counts')):
            return None
        lb = _LaneBatch()
        rd, tg, made, PR = self.region_data, self.params.targets, self.targets._made, hip_backend.PackedReads
        for key in keys:
            d = rd.get(key)
            if d is None or key in made or type(d.read_packed) is not PR or d.sc_seqs is not None or d.partners or type(d.window) is not str or d.read_codes is None:
                return None
            if not len(d.read_ids):
                continue
            ivs = tg[key]
            v = ivs[0]
            if len(ivs) == 1:
                region = (v[0], v[1], v[2], v[3], ivs)
            else:                                                      # target.setup (:267-294): first name and chromosome, the hull of the intervals
                region = (v[0], min(x[1] for x in ivs), max(x[2] for x in ivs), v[3], ivs)
            lb.keys.append(key); lb.names.append(v[3]); lb.regions.append(region); lb.data.append(d)
        return lb

    def _lane_submit(self, eng, lb):
        """False: a window of the batch is empty -- the batch goes the per-target way, which skips the offender alone.  (The
        CHARACTERS of the windows are not looked at here: the library does when it packs them, on its own thread; a foreign one
        makes its submit fail, which _launch_batch hears of and then sends the batch the per-target way too.)"""
        wins = [d.window_bytes() for d in lb.data]
        if not all(wins):
            return False
        as_c, u8 = hip_backend._as_c, _np.uint8
        eng.submit_packed([(d.read_packed, w, None if d._no_indel_only else as_c(d.indel_only, u8)) for d, w in zip(lb.data, wins)], wait=False)
        return True

    def _lane_finish(self, eng, lb, order):
        rows = eng.call()
        counts = eng.contig_counts()
        failed = {}
        if hasattr(eng, 'region_status') and not (hasattr(eng, 'stat') and eng.stat(22) == 0):
            for i, name in enumerate(lb.names):
                st, text = eng.region_status(i)
                if st != 0:
                    self.logger.error('target %s: not assembled on the device: %s' % (name, text))
                    failed[i] = self.failed_targets[name] = text
        done, summary, results, none = self._lane_done, self.summary, self.results, []
        for i, key in enumerate(lb.keys):
            if failed and i in failed:
                done[key] = (none, 0, failed[i])
                continue
            r = rows.get(i, none)
            ni = nr = nt = 0
            if r:
                oi = order[key]
                for x in r:                                            # get_sv_counts (:686-705)
                    tag = x[6]
                    if tag == 'indel': ni += 1
                    elif tag == 'trl': nt += 1
                    elif tag.find('rearrangement') > -1: nr += 1
                    else: raise KeyError(tag)
                    results.append([oi, x])
            summary[lb.names[i]] = "%s\t%d\t%d\t%d\t%d\t%d\t-" % (lb.names[i], counts[i], len(r), ni, nr, nt)
            done[key] = (r, counts[i], None)
        self.summary_header = _SUMMARY_HEADER

    def _submit_batch(self, eng, live):
        """hand one batch of targets to the library; with a HIP engine the 2-bit packing and the copies run on the library's
        own thread (BK_SUBMIT_ASYNC) while this thread goes on with the previous batch"""
        # (targets whose window was not looked at yet -- RegionData.checked_at_submit -- are checked here: all windows of the batch in
        # one scan; only if that finds a foreign character are they looked at one by one, and the offenders skipped ALONE as ever)
        if type(live) is _LaneBatch:
            return self._lane_submit(eng, live)
        late = [t for t in live if t.check_at_submit]
        if late and b"".join(t.data.window_bytes() for t in late).translate(None, b"ACGTNacgtn") or any(not t.data.window for t in late):
            for t in late:
                why = t.unsupported_reference()
                if why:
                    self.logger.error('target %s: skipped: %s' % (t.name, why))
                    self.failed_targets[t.name] = why
                    t.rm_output_dir()
                    live.remove(t)
            if not live:
                return False
        if hasattr(eng, 'submit_packed'):
            items = [t.data.packed_item() for t in live]
            if None not in items:                                     # the common case of a packing read extraction: one struct.pack per target
                for i, t in enumerate(live):
                    t.region_index, t.engine = i, eng
                eng.submit_packed(items, wait=False)
                return
        ins = []
        allp = all(t.data.read_packed is not None for t in live)          # a batch goes over packed (BK_SUBMIT_PACKED) or not at all
        for i, t in enumerate(live):
            t.region_index, t.engine = i, eng
            ins.append(t.data.device_view(allp))
        try:
            eng.submit(ins, wait=False)
        except TypeError:
            eng.submit(ins)

    def _launch_batch(self, eng, live):
        """start the GPU stages of a submitted batch (asynchronous where the engine supports it) and give the library the
        call context of its regions"""
        try:
            eng.run(hip_backend.BK_STAGE_ALL, sync=False)
        except TypeError:
            eng.run(hip_backend.BK_STAGE_ALL)
        except hip_backend.BreakmerHipError as ex:
            # the (asynchronous) submit of a lane batch was refused (BK_E_ARG: a character other than A/C/G/T/N in a window, ...): the
            # per-target way names the target, skips it ALONE and hands the others over again.  The branch is on the library's error
            # CODE (hip_backend.BreakmerHipError.code), never on the wording of its message
            if type(live) is not _LaneBatch or getattr(ex, "code", 0) != hip_backend.BK_E_ARG:
                raise
            live = self._prepare(live.keys)
            if not live or self._submit_batch(eng, live) is False:
                return []
            self._launch_batch(eng, live)
            return live
        if self.native_calls and hasattr(eng, 'set_call_context'):
            if self._ctx_head is None:                               # options + annotation tables: the same text for every batch
                self._ctx_opts = cc.opts_line(self.params.opts)
                self._ctx_head = "\n".join([self._ctx_opts] + cc.tables_lines(self.params.gene_annotations.genes, self.params.repeat_mask))
            # a handle keeps the tables of its previous context: they are sent (and parsed) once per handle and run
            if getattr(eng, '_ctx_tables_of', None) is self._ctx_token and hasattr(eng, 'h'):
                lines = [self._ctx_opts, "keep_tables"]
            else:
                lines = [self._ctx_head]
                eng._ctx_tables_of = self._ctx_token
            if type(live) is _LaneBatch:
                rl, add = cc.region_lines, lines.append
                for i, d in enumerate(live.data):
                    rg, dr = live.regions[i], d.disc_reads
                    ivs = rg[4]
                    if len(ivs) == 1 and getattr(d.read_ids, "uniform_tag", None) is not None and not (dr.get("inv") or dr.get("td") or dr.get("other") or dr.get("disc")):
                        # the common shape -- one interval, one class of read ids, no discordant pairs -- in one piece (the lines region_lines makes)
                        add("region %d %s %d %d %s\niv %d %d %d\nrtags 0" % (i, rg[0], rg[1], rg[2], rg[3], rg[1], rg[2], 1 if ivs[0][4] == 'exon' else 0))
                    else:
                        lines += rl(i, rg, None, dr, (), d.read_ids)
            else:
                for i, t in enumerate(live):
                    lines += cc.region_lines(i, t.get_values(), t.repeat_mask, t.disc_reads, t.partner_windows, t.data.read_ids)
            eng.set_call_context("\n".join(lines) + "\n")
            if hasattr(eng, 'call_async'):                             # the wait for the GPU, the copy back and the call tail run on the library's thread
                eng.call_async()

    def _prepare(self, keys):
        """the reference's per-target steps before the hot path (sv_processor.py:183-192) for the targets of one batch -> those that go on"""
        live = []
        for n in keys:
            t = self.targets[n]
            t.set_ref_data()
            t.extract_bam_reads()
            if not t.clean_reads():
                t.rm_output_dir()
                continue
            t.check_at_submit = t.data.checked_at_submit()
            why = None if t.check_at_submit else t.unsupported_reference()
            if why:                                           # this target only; the run goes on (summary + exit code report it)
                self.logger.error('target %s: skipped: %s' % (t.name, why))
                self.failed_targets[t.name] = why
                t.rm_output_dir()
                continue
            live.append(t)
        return live

    def _start_batch(self, eng, live):
        self._submit_batch(eng, live)
        self._launch_batch(eng, live)

    def _finish_batch(self, eng, live, order):
        """wait for a batch, then the reference's per-target sequence compare_kmers -> resolve_sv -> summary -> files"""
        if hasattr(eng, 'sync'):
            eng.sync()
        if type(live) is _LaneBatch:
            return self._lane_finish(eng, live, order)
        # a region that hit a device cap fails alone (bk_get_region_status): the target is logged and skipped like a
        # target without reads (sv_processor.py:190-192); the rank still takes part in the collation
        if hasattr(eng, 'region_status') and not (hasattr(eng, 'stat') and eng.stat(22) == 0):      # stat 22: regions that failed in the last run
            for i, t in enumerate(live):
                st, text = eng.region_status(i)
                if st != 0:
                    self.logger.error('target %s: not assembled on the device: %s' % (t.name, text))
                    t.failed = text
                    self.failed_targets[t.name] = text
        if self.native_calls and hasattr(eng, 'set_call_context'):
            rows = eng.call()
            counts = eng.contig_counts() if hasattr(eng, 'contig_counts') else None      # one library call per batch instead of one per target
            for i, t in enumerate(live):
                t.native_rows = rows.get(i, [])
                if counts is not None:
                    t.n_contigs = counts[i]
        for t in live:
            if t.failed:
                t.rm_output_dir()
                continue
            t.compare_kmers()
            t.resolve_sv()
            if not t.has_results() and not t.genome_searched and self._wants_genome_search(t):
                wins = self._genome_partners(eng, t)
                t.genome_searched = True
                if wins:                                              # run the target again, its contigs realigned against these windows too
                    self.logger.info('target %s: %d partner window(s) from the genome-wide search: %s' % (t.name, len(wins), ", ".join("%s:%d-%d" % (w[0], w[1], w[2]) for w in wins)))
                    t.data.partners = list(t.data.partners) + wins
                    t.partner_windows = t.data.partners
                    t.native_rows = None; t.results = []; t.kmers['clusters'] = []
                    self._retry.append(t)
                    continue
            self.summary_header, self.summary[t.name] = t.get_summary()
            if t.has_results():
                if 'output' in t.paths:
                    t.write_results()
                self.results.extend([order[t.name.upper()], r] for r in t.results)      # (the names of the run are the upper-cased BED names, utils.py:545-572)
            else:
                t.rm_output_dir()
            t.release()

    def _wants_genome_search(self, t):
        """the reference hands every contig its target window does not explain to the whole-genome gfServer
        (sv_processor.py:829-831); here that costs an index of the genome, so it is done for targets that came out WITHOUT a
        call and have contigs, when a `reference_fasta` is configured (`genome_search` = False switches it off)"""
        if str(self.params.opts.get('genome_search', True)).lower() in ('false', '0', 'no') or self.params.open_fasta() is None or t.data is None:      # config files hold strings
            return False
        try:
            return len(t.kmers.get('clusters') or []) > 0 and len(t.data.partners) < 8
        except TypeError:
            return False

    def _genome_partners(self, eng, t):
        """partner windows for the contig segments of target t that no record covers (>= trl_minseg_len bases): every k-mer of
        such a segment is looked up in the genome index; loci with >= 2 index hits in one diagonal band, outside the target's
        own window, become windows of +-1,500 bases (at most 4, best supported first)"""
        if not hasattr(eng, 'hits') or self.params.open_fasta() is None:
            return []
        minseg = max(20, self.params.get_min_segment_length('trl'))
        # the contig segments no record covers -- only if there is one is the genome index needed at all (building it, or reading
        # its cache file, is the expensive part: most targets without a call have contigs their window explains completely)
        segs = []
        for ci, c in enumerate(eng.contigs(t.region_index)):
            seq = c["seq"]
            cov = bytearray(len(seq))
            for h in eng.hits(t.region_index, ci):
                cov[h["q_start"]:h["q_end"]] = b"\x01" * max(0, h["q_end"] - h["q_start"])
            a = 0
            while a < len(seq):
                if cov[a]:
                    a += 1
                    continue
                b = a
                while b < len(seq) and not cov[b]:
                    b += 1
                if b - a >= minseg:
                    segs.append(seq[a:b])
                a = b
        if not segs:
            return []
        # (the look-ups run on the device the engine runs on when it is the HIP engine; the test engines use the host path)
        gi = self.params.genome_index(device=getattr(eng, 'device', None) if hasattr(eng, 'h') else None)
        if gi is None:
            return []
        loci = {}
        for sg in segs:
            for nh, name, _strand, s0, e0 in gi.find(sg)[:4]:
                c_ = name.replace("chr", "")
                if c_ == str(t.chrom).replace("chr", "") and e0 >= t.start - 200 and s0 <= t.end + 200:
                    continue
                key = (c_, s0 // 1000)
                if key not in loci or loci[key][0] < nh:
                    loci[key] = (nh, c_, s0, e0)
        fa = self.params.open_fasta()
        out = []
        for nh, c_, s0, e0 in sorted(loci.values(), key=lambda x: (-x[0], x[1], x[2]))[:4]:
            s1, e1 = max(0, s0 - 1500), min(fa.length(c_), e0 + 1500)
            wseq = fa.fetch(c_, s1, e1)
            if len(wseq) < 64 or wseq.strip("ACGTNacgtn"):
                self.logger.warning('target %s: genome window %s:%d-%d holds characters other than A/C/G/T/N and is not realigned against' % (t.name, c_, s1, e1))
                continue
            name = self.params.gene_annotations.set_gene(c_, [(s0 + e0) // 2])
            out.append((c_, s1, e1, name, wseq))
        return out

    def _make_engine(self):
        if self.engine_factory:
            return self.engine_factory(self.params)
        dev = int(os.environ.get("LOCAL_RANK", "0")) if self.world > 1 else 0                # one process per GPU
        tm = str(self.params.opts.get('throughput_mode', '')).lower() in ('1', 'true', 'yes')
        eng = hip_backend.acquire_engine(self.params.get_kmer_size(), self.params.get_sr_thresh('min'), dev,      # kept between runs of one process
                                         flags=hip_backend.BK_CFG_NO_SPLIT if tm else 0, wg_threads=256 if tm else 0)
        self._pooled.append(eng)
        return eng

    def run(self, start_time=None):                                  # :174-209
        with _gc_paused():
            return self._run(start_time)

    def _run(self, start_time=None):
        names = self.create_targets()
        # Partition over the ranks: regions differ in cost by orders of magnitude (depth, noise, SV type), so the
        # targets are dealt heaviest first to the rank with the least load so far (cost = number of reads, known on every
        # rank: no communication); ties and the order inside a rank follow the sorted names.  The rows are put back into
        # target order after the collation.
        order = {n: i for i, n in enumerate(names)}
        if self.world > 1:
            cost = {n: (len(self.region_data[n].read_ids) if n in self.region_data else int(self.targets[n].cost_estimate())) for n in names}
            load = [0] * self.world
            owner = {}
            for n in sorted(names, key=lambda x: (-cost[x], x)):
                rk = min(range(self.world), key=lambda r_: (load[r_], r_))
                owner[n] = rk
                load[rk] += max(1, cost[n])
            mine = [n for n in names if owner[n] == self.rank]
            self.assigned_cost, self.assigned_targets, self.rank_loads = load[self.rank], list(mine), list(load)
            self.logger.info('rank %d of %d: %d of %d targets, estimated cost %d (all ranks: %s)' % (self.rank, self.world, len(mine), len(names), load[self.rank], load))
        else:
            mine = list(names)
        # Batching front-end: the reference handles one target at a time; here bounded batches of targets go through the HIP
        # library on up to three handles.  A batch is handed to the library one iteration before its kernels are launched
        # (its 2-bit packing and H2D copies run on a thread of the library meanwhile); its kernels are launched BEFORE this
        # thread picks up the batch launched one iteration earlier (call tail, per-target objects, files), so the GPU is never
        # waited for.  Batches finish in the order they were started.
        bsz = max(1, int(self.params.opts.get('batch_regions', 256)))
        # submits in flight before a batch's kernels are launched: with one, the launch of a batch of packed reads still waited ~1 ms for
        # the library's thread (row copies + H2D of 100 MB take longer than this thread needs for a batch); a handle per batch in flight
        depth = max(1, int(self.params.opts.get('submit_depth', 2)))
        # launched batches in flight before the oldest is picked up.  1 (default): the batch launched one iteration earlier is picked up right
        # after the next one is launched -- right for clean reads, whose batches take a millisecond.  Realistic (noisy) reads make a batch take
        # a third of a second, bound by the serial chain of its slowest region: with `run_depth` 6-8 and `throughput_mode` (one
        # workgroup per region on 256-thread workgroups: hip_backend.acquire_engine) the chip stays full -- 2,050 against 790 regions/s on the
        # library's own loop (bench.py other_configs.noise_0.5pct_256_regions).  Rows are the same either way.
        run_depth = max(1, int(self.params.opts.get('run_depth', 1)))
        free, pending, running = [], [], []                         # handles; submitted batches; the launched batches, oldest first

        def advance():
            eng, live = pending.pop(0)
            again = self._launch_batch(eng, live)                    # the GPU starts on this batch ...
            if again is not None:                                     # (a lane batch that had to go the per-target way after all)
                live = again
            while len(running) >= run_depth:                          # ... while the oldest one before it (long finished) is picked up
                done = running.pop(0)
                self._finish_batch(done[0], done[1], order)
                free.append(done[0])
            if live:
                running.append((eng, live))
            else:
                free.append(eng)

        ok, failure = False, None
        try:
            lane = self._lane_ok()
            for b0 in range(0, len(mine), bsz):
                eng = None
                if lane:                                               # a batch of plain targets goes through as a table
                    eng = free.pop() if free else self._make_engine()
                    self.engine = eng
                    lb = self._lane_batch(eng, mine[b0:b0 + bsz])
                    if lb is not None and not len(lb):                 # nothing but targets without reads
                        free.append(eng)
                        continue
                    if lb is not None and self._submit_batch(eng, lb) is not False:
                        pending.append((eng, lb))
                        if len(pending) > depth:
                            advance()
                        continue
                live = self._prepare(mine[b0:b0 + bsz])
                if not live:
                    if eng is not None:
                        free.append(eng)
                    continue
                if eng is None:
                    eng = free.pop() if free else self._make_engine()
                self.engine = eng
                if self._submit_batch(eng, live) is False:            # every target of the batch was skipped at the submit-time window check
                    free.append(eng)
                    continue
                pending.append((eng, live))
                if len(pending) > depth:
                    advance()
            while pending:
                advance()
            while running:
                done = running.pop(0)
                self._finish_batch(done[0], done[1], order)
                free.append(done[0])
            while self._retry:                                        # second pass (N4): targets with partner windows from the genome-wide search
                again, self._retry = self._retry[:bsz], self._retry[bsz:]
                eng = free.pop() if free else self._make_engine()
                self.engine = eng
                self._start_batch(eng, again)
                self._finish_batch(eng, again, order)
                free.append(eng)
            ok = True
        except Exception as ex:                                        # with several ranks the others must hear of it (below) before this one stops
            if self.world <= 1 or self.status_exchange is None:
                raise
            failure = ex
            self.logger.error('rank %d failed: %s: %s' % (self.rank, type(ex).__name__, ex))
        finally:
            for eng in self._pooled:                                    # back to the pool after a clean run, closed otherwise
                if ok:
                    hip_backend.release_engine(eng)
                else:
                    eng.close()
            self._pooled = []
            self.engine = None
        if self.world > 1 and self.status_exchange is not None:
            # Every rank says whether it got through its targets BEFORE the collation: a rank that stopped with an exception
            # would otherwise leave the others waiting in the all-gather until the launcher kills them.  One failed rank
            # makes EVERY rank raise (exit code != 0); a fresh launch is the restart.
            errs = self.status_exchange(None if failure is None else "%s: %s" % (type(failure).__name__, failure))
            bad = ["rank %d: %s" % (rk, e) for rk, e in enumerate(errs) if e]
            if failure is not None:
                raise failure
            if bad:
                raise RuntimeError("multi-rank run aborted, " + "; ".join(bad))
        if self.collate is not None and self.world > 1:               # collate per-region rows over ranks (RCCL all-gather)
            self.results, self.summary = self.collate(self.results, self.summary)
            if self.status_exchange is not None:                      # ... and which targets were skipped anywhere (exit code of every rank)
                merged = {}
                for d_ in self.status_exchange(self.failed_targets):
                    merged.update(d_ or {})
                self.failed_targets = merged
        self.results.sort(key=lambda x: x[0])                         # stable: target order (sv_processor.py:175-176), rows of a target as produced
        self.results = [r for _i, r in self.results]
        if self.rank == 0 and 'output' in self.params.paths:
            self.write_output()
        if self.failed_targets:                                      # never silent: the reference has no caps, so these are results that are missing
            self.logger.error('%d target(s) without result on rank %d: %s' % (len(self.failed_targets), self.rank,
                              "; ".join("%s (%s)" % kv for kv in sorted(self.failed_targets.items()))))
        return self.results

    def write_output(self):                                          # :212-234
        files = {}
        out = self.params.paths['output']
        for res in self.results:
            tag = res[6]
            if tag not in files:
                files[tag] = open(os.path.join(out, self.params.opts['analysis_name'] + "_" + tag + "_svs.out"), 'w')
                if not self.params.opts['no_output_header']:
                    files[tag].write("\t".join(HEADER_FIELDS) + "\n")
            files[tag].write("\t".join(str(x) for x in res) + "\n")
        for f in files.values():
            f.close()
        with open(os.path.join(out, self.params.opts['analysis_name'] + "_summary.out"), 'w') as f:
            f.write(self.summary_header + "\n")
            for gene in sorted(self.summary.keys()):
                f.write(self.summary[gene] + "\n")

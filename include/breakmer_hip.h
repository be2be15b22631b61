/* breakmer_hip.h -- C-ABI of libbreakmer_hip.so: the MI355X (gfx950) implementation of
 * BreaKmer's per-target-region hot path, batched over regions.
 *
 * The reference has no FFI: its seams are Python calls and two text-file contracts.
 * Each entry point below names the reference interface it replaces (file:line under the reference
 * tree).  Conventions: plain C types only; every function returns 0 on success or a negative
 * BK_E_* code (bk_last_error() gives the text); the library never frees caller memory; outputs are
 * copied into caller-allocated buffers after a size query and stay valid until the next bk_run on
 * the same handle.  One handle = one HIP device + one stream; handles are independent; calls on
 * one handle are not re-entrant.
 */
#ifndef BREAKMER_HIP_H
#define BREAKMER_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BK_ABI_VERSION 5       /* 2: bk_psl holds 32 blocks; bk_get_hits also returns secondary alignments; bk_sync reports regions that hit a cap (BK_W_REGIONS_FAILED)
                                * 3: the realign stage has no hit / block caps any more (bk_get_hits_flat returns records of any size, bk_call uses them);
                                *    reference / partner windows may be soft-masked (lower case => rep_matches); regions that overflow an assembler cap are
                                *    re-run by the library with larger ones (bk_config.no_escalation); BK_SUBMIT_PACKED
                                * 4: noisy regions are split over up to 16 assembler workgroups by default (same results; BK_CFG_NO_SPLIT = never);
                                *    the realign records pass BLAT's documented output filters (-minScore, -minIdentity default 90: oracle/bk_oracle.h
                                *    step 8); bk_index_set_loci / bk_index_find; the product build reads no environment variable (diagnostics: -DBK_DIAG)
                                * 5: what bk_config.reserved[0..3] carried until ABI 4 has NAMES (flags, asm_wg_threads, no_escalation, submit_threads: same
                                *    offsets, same values); bk_create rejects a non-zero `reserved` word, an unknown flag bit and -- in the product build --
                                *    the diagnostic-only bits (BK_CFG_DIAG_MASK); submit errors that name a region carry its index (bk_last_error_region) */

enum {
    BK_OK = 0,
    BK_E_ARG = -1,          /* bad argument */
    BK_E_HIP = -2,          /* HIP runtime error (text in bk_last_error) */
    BK_E_NOGPU = -3,        /* no usable gfx950 device: there is NO CPU fallback */
    BK_E_LIMIT = -4,        /* a documented device limit was exceeded (contig/read/candidate caps) */
    BK_E_STATE = -5,        /* call out of order (e.g. bk_run before bk_submit_regions) */
    BK_E_NOMEM = -6,
    BK_W_REGIONS_FAILED = 1 /* bk_sync only: the batch is done, but some regions hit a device cap and report no contigs
                             * (bk_get_region_status says which and why; bk_get_stat(22) how many) */
};

/* stage mask for bk_run */
enum {
    BK_STAGE_KMER = 1,      /* T1 read grouping + K1/K2 sample-only k-mer selection            */
    BK_STAGE_ASSEMBLE = 2,  /* A1..A12 init_assembly incl. olc.nw                               */
    BK_STAGE_REALIGN = 4,   /* R2 contig -> reference-window Smith-Waterman (replaces BLAT)     */
    BK_STAGE_ALL = 7
};

typedef struct bk_handle bk_handle;

/* Parameters the reference reads from its config/options (breakmer.py:73-86, utils.py:659-674). */
typedef struct bk_config {
    int32_t abi_version;        /* = BK_ABI_VERSION */
    int32_t kmer_size;          /* kmer_region.config:16 `kmer_size`, utils.py:659 (<= 64) */
    int32_t rc_thresh;          /* params.get_sr_thresh('min'), sv_processor.py:642 */
    int32_t max_contig_len;     /* device cap on contig length (default 4096)      */
    int32_t max_read_len;       /* device cap on read length   (default 1024)      */
    int32_t max_candidates;     /* cap on reads returned by one find_reads (default 2048) */
    int64_t arena_bytes;        /* device scratch arena; 0 = choose from the batch; grown and retried on overflow */
    int32_t sw_min_score;       /* BLAT -minScore=20 analogue for the realign stage (sv_processor.py:843) */
    int32_t out_kbytes;         /* initial result arena in KiB; 0 = choose from the batch; grown and retried on overflow */
    uint32_t flags;             /* BK_CFG_* bits below; 0 in production */
    int32_t asm_wg_threads;     /* assembler workgroup size: 512 = 8 wavefronts / 8 look-ahead slots / 2 per CU (one batch finishes soonest),
                                 * 256 = 4 wavefronts / 8 slots / 4 per CU (most regions/s when batches are in flight), 0 = the library chooses */
    int32_t no_escalation;      /* 1 = do NOT re-run regions that overflow an assembler cap under larger caps (they fail at once: bk_get_region_status) */
    int32_t submit_threads;     /* host threads that fill the staging buffer of one submit (0 = the library chooses; 1 .. 64) */
    int32_t reserved[2];        /* must be 0 (bk_create: BK_E_ARG otherwise) */
} bk_config;

/* bk_config.flags.  Every bit selects another way to the SAME results (the GPU suite holds them against each other); production leaves 0.
 *   option        BK_CFG_NO_SPLIT: noisy regions (>= 1,024 seed k-mers) are NOT split into up to 16 units on 16 workgroups (the split is the
 *                 default since ABI 4: identical results, 2-6x faster on such regions);
 *   test hooks    accepted by every build -- the parity tests run the product's alternative paths through them;
 *   diagnostics   BK_CFG_DIAG_MASK: accepted by the -DBK_DIAG builds only (breakmer_amd/build.py: diag / check / jitter / stamps); the product
 *                 build's bk_create answers BK_E_ARG. */
enum {
    BK_CFG_DIAG_NO_DUAL = 1,              /* diagnostic: one overlap DP per wavefront even for short contigs */
    BK_CFG_DIAG_SPEC4 = 2,                /* diagnostic: at most 4 look-ahead slots */
    BK_CFG_DIAG_DUAL_ALWAYS = 4,          /* diagnostic: both DPs of a slot on one wavefront whatever the number of reads in the round */
    BK_CFG_TEST_NO_XVISIT = 8,            /* test hook: no look-ahead across the k-mer visits of grow */
    BK_CFG_TEST_NO_XSEED = 16,            /* test hook: no look-ahead into the next seeds */
    BK_CFG_TEST_BUCKET_SORT = 32,         /* test hook: bucket sort of the seed k-mers whatever their number (the path of very large noisy regions) */
    BK_CFG_TEST_NO_RUN_RETIRE = 64,       /* test hook: every read retired on its own */
    BK_CFG_NO_SPLIT = 128,                /* option: one assembler workgroup per region, never split */
    BK_CFG_TEST_SPLIT_ALWAYS = 256,       /* test hook: split whatever the size of the region (the split path on small fixtures) */
    BK_CFG_DIAG_SPLIT_NO_LOOKAHEAD = 512, /* diagnostic: no look-ahead inside split regions (the round-4 setting) */
    BK_CFG_DIAG_LEGACY_SPLIT = 1024,      /* diagnostic builds accept it, no effect (it switched the split on while it was opt-in, ABI 3) */
    BK_CFG_TEST_FULL_CALLER = 2048,       /* test hook: bk_call takes every contig through the full caller (no shortcut for single full-span hits) */
    BK_CFG_TEST_HOST_REPAIR = 4096,       /* test hook: repair passes of split regions driven by the host (the fallback when the unit queue is full) */
    BK_CFG_TEST_NO_SCORE_SWEEP = 8192,    /* test hook: every overlap DP is the full sweep with origins (the round-4 DP rounds) */
    BK_CFG_TEST_PREQUEUE_UNITS = 16384,   /* test hook: the units of a split region are queued at launch and wait for unit 0 (the round-5 queue) instead of being appended by it */
    BK_CFG_DIAG_FORCE_REDO = 32768,       /* diagnostic: every read of a long-contig round is swept again in full after the score sweep (the redo passes, which real data takes for ~2 % of such reads) */
    BK_CFG_DIAG_MASK = 1 | 2 | 4 | 512 | 1024 | 32768,
    BK_CFG_KNOWN_MASK = 65535
};

/* One target region = what sv_processor.target hands to compare_kmers()/resolve_sv()
 * (sv_processor.py:609-665): the cleaned reads (self.cleaned_read_recs before grouping,
 * utils.py:203-246), the soft-clip sequences behind case_sc (sv_processor.py:619-620) and the
 * forward reference window FASTA (sv_processor.py:291; the reverse file is derived).
 * Sequences are ASCII A/C/G/T/N (an N in a read or a window matches nothing and no k-mer spans it, as for Jellyfish and BLAT;
 * soft-clip sequences must be A/C/G/T: a caller splits them at any other character, which yields the same k-mer set), rows
 * `stride` bytes apart with explicit lengths.  Windows (target and partners) may be soft-masked: a lower-case base is the same
 * base, and a realign match on it is reported in bk_psl.rep_matches instead of .matches -- what BLAT's -repeats=lower
 * (sv_processor.py:843) does and sv_caller.py:913, 975-986 reads. */
typedef struct bk_region {
    const char *reads;          /* n_reads rows */
    const uint16_t *read_lens;
    const uint8_t *indel_only;  /* fq_read.indel_only per read (utils.py:688); may be NULL = all 0 */
    int32_t n_reads;
    int32_t read_stride;
    const char *sc_seqs;        /* soft-clip/unmapped sequences; n_sc < 0 => case_sc := case */
    const uint16_t *sc_lens;
    int32_t n_sc;
    int32_t sc_stride;
    const char *window;         /* forward target window, [start-200, end+200) (utils.py:367) */
    int32_t window_len;
    int32_t n_partners;         /* extra windows for realignment (whole-genome fallback stand-in, sv_processor.py:829-831) */
    const char *const *partners;   /* n_partners sequences (ASCII) and their lengths; both may be NULL when n_partners == 0 */
    const int32_t *partner_lens;
    const uint32_t *read_n;     /* BK_SUBMIT_PACKED only: the N calls of the reads, (read index << 10 | position), ascending; may be NULL */
    int32_t n_read_n;
} bk_region;

/* Lifetime (replaces: process start + params(), sv_processor.py:99-105). Fails with BK_E_NOGPU when
 * no gfx950 device is visible. */
int bk_create(int device_id, const bk_config *cfg, bk_handle **out);
int bk_destroy(bk_handle *h);
const char *bk_last_error(const bk_handle *h);   /* h may be NULL: last error of bk_create */
/* The region a failed bk_submit_regions[_ex] names in its error text (BK_E_ARG: a character other than A/C/G/T/N in a window or a read,
 * a read longer than max_read_len, ...), or -1 when the last error names none.  A batching driver branches on THIS, not on the wording:
 * it hands the other targets over again and skips that one alone (the per-target analogue: sv_processor.py:190-192). */
int32_t bk_last_error_region(const bk_handle *h);
/* Free every device / pinned buffer of the handle larger than keep_bytes (the scratch arena grows with the largest batch it
 * has seen and is otherwise only freed by bk_destroy); the submitted batch is dropped, the next bk_submit_regions sizes
 * the buffers anew.  For callers that keep handles between runs (the reference creates everything per run). */
int bk_trim(bk_handle *h, uint64_t keep_bytes);
int bk_abi_version(void);

/* Pack the regions' sequences to 2 bit/base and make them resident in HBM
 * (replaces: writing <name>_sv_reads.fastq / *_refseq.fa for jellyfish and the assembler,
 * sv_processor.py:584-606, utils.py:355-381). */
int bk_submit_regions(bk_handle *h, const bk_region *regions, int32_t n_regions);
/* Same, for callers that already hold the reads as base codes (one byte per base: 0..3 = A,C,G,T, 4 = N) instead of the
 * reference's strings: with BK_SUBMIT_READ_CODES the `reads` rows are codes (windows / soft-clip / partner sequences stay ASCII). */
#define BK_SUBMIT_READ_CODES 1u
/* BK_SUBMIT_ASYNC: return at once and pack + copy on a thread of the library, so that a driver can pick up the previous
 * batch of another handle meanwhile.  The bk_region array is copied; the sequences it points to must stay valid until the
 * next call on this handle has returned.  Every later call on the handle first waits for the submit and, if it failed,
 * returns its error code (bk_last_error has the text) instead of doing its own work. */
#define BK_SUBMIT_ASYNC 2u
/* BK_SUBMIT_PACKED: the `reads` rows are ALREADY 2 bit/base as bk_pack_sequence makes them (16 bases per 32-bit word, first base in
 * the most significant bits, an N packed as A), ceil(len / 16) words per row, rows read_stride BYTES apart; the N calls come as
 * bk_region.read_n.  A quarter of the bytes of BK_SUBMIT_READ_CODES (96 MB instead of 384 MB per 256-region batch of the headline
 * shape) and no packing on the submit path: for callers whose read extraction packs as it goes. */
#define BK_SUBMIT_PACKED 4u
int bk_submit_regions_ex(bk_handle *h, const bk_region *regions, int32_t n_regions, uint32_t flags);

/* Run the selected stages on everything submitted (replaces, per region:
 *   BK_STAGE_KMER     run_jellyfish x4 + load_kmers + set algebra, utils.py:151-178,287-296, sv_processor.py:613-631
 *   BK_STAGE_ASSEMBLE init_assembly(mers, fq_recs, kmer_len, rc_thresh, read_len), sv_assembly.py:30
 *   BK_STAGE_REALIGN  contig.query_ref -> blat -> PSL, sv_processor.py:823-851).
 * Asynchronous on the handle's stream; bk_sync() or any bk_get_* waits. */
int bk_run(bk_handle *h, uint32_t stage_mask);
int bk_sync(bk_handle *h);         /* BK_OK, BK_W_REGIONS_FAILED (see above) or a negative BK_E_* code */
/* Wait for the last bk_run and copy its result records to the host.  The device buffers of the handle are free
 * for the next bk_run after this call: a following bk_call() works on the host copy even when a new run has been
 * started in between (the batching analogue of the reference's per-region loop moving on to the next target,
 * sv_processor.py:185-201, while make_calls of the previous one is still being formatted). */
int bk_fetch(bk_handle *h);
/* elapsed device time of the kernels of the last bk_run, measured with HIP events on the handle's
 * stream (ms); which = 0 total, 1 k-mer kernel, 2 assembler kernel, 3 realign kernel */
int bk_last_kernel_ms(bk_handle *h, int which, float *ms);

/* Per-region outcome of the last bk_run.  The reference has no size caps; this implementation sizes its LDS buffers by a few
 * (max_candidates reads per k-mer visit, max_contig_len, the contig k-mer list).  A region that overflows one is RUN AGAIN by the
 * library, alone with the others that did, with caps 4x larger (8,192 candidates, 16,384-base contigs with the defaults; one
 * workgroup per CU; bk_config.reserved[2] = 1 switches that off) -- bk_sync does it before it returns.  Only a region that
 * overflows those too fails: ALONE -- status != 0, text says which cap, it reports zero contigs -- and bk_run/bk_sync still
 * succeed for the batch (the per-target analogue of the reference skipping a target, sv_processor.py:190-192).  status 0 = ok. */
int bk_get_region_status(bk_handle *h, int32_t region, int32_t *status, const char **text);

/* ---- results, region-major -------------------------------------------------------------- */
/* K1/K2: sample-only k-mers of region r in the order init_assembly visits them
 * ((count, mer) descending, sv_assembly.py:281); mers = n*k ASCII bytes. */
int bk_get_kmer_count(bk_handle *h, int32_t region, int32_t *n_mers, int32_t *n_unique_reads);
int bk_get_kmers(bk_handle *h, int32_t region, char *mers, int32_t *counts, int32_t cap);

/* contigs of region r, in the order init_assembly returns them (sv_assembly.py:59) */
typedef struct bk_contig_info {
    int32_t seq_len;        /* len(contig.aseq.seq)                      */
    int32_t counts_len;     /* len(contig.aseq.counts.others) (may differ: set_superseq quirk, sv_assembly.py:190-191) */
    int32_t n_kmers;        /* len(contig.kmers)                         */
    int32_t n_reads;        /* len(contig.reads)                         */
    int32_t total_reads;    /* contig.get_total_read_support()           */
    int32_t n_hits;         /* realign stage: PSL-equivalent records     */
} bk_contig_info;
int bk_get_contig_count(bk_handle *h, int32_t region, int32_t *n_contigs);
int bk_get_contig_counts(bk_handle *h, int32_t *n_contigs, int32_t cap);   /* all regions of the batch at once (cap >= number of regions) */
int bk_get_contig_info(bk_handle *h, int32_t region, int32_t contig, bk_contig_info *info);
/* any pointer may be NULL; sizes from bk_contig_info.
 *   seq        seq_len ASCII bytes            (contig.get_contig_seq(), sv_assembly.py:443)
 *   indel_only counts_len ints, others ditto  (contig.get_contig_counts(), :446)
 *   kmer_locs  seq_len ints                   (contig.get_kmer_locs(), :440)
 *   kmers      n_kmers*k ASCII bytes          (x[0] of contig.kmers, sv_processor.py:760)
 *   reads      n_reads ints: index (FASTQ order within the region) of each supporting read's representative */
int bk_get_contig(bk_handle *h, int32_t region, int32_t contig, char *seq, int32_t *indel_only, int32_t *others,
                  int32_t *kmer_locs, char *kmers, int32_t *reads);

/* realign stage: PSL-equivalent records of one contig (fields consumed by sv_caller.py:911-936): the chained
 * records of the iterated search first (forward query order), then every SECONDARY alignment -- any other gap-free
 * segment scoring >= sw_min_score on another diagonal, the other strand or another window, as BLAT prints every
 * alignment >= -minScore (sv_processor.py:843) and the caller counts them per query base (sv_caller.py:593-594,
 * 616-631, 430-432, 55-72) -- as one-block records ordered by (score desc, window asc, '+' first, query end, target
 * end).  Returns the number of records (>= 0, may exceed cap) or a negative BK_E_* code (BK_E_LIMIT: a chained record of THIS
 * contig has more than BK_MAX_BLOCKS blocks and does not fit bk_psl -- bk_get_hits_flat and bk_call have no such limit).
 * q_starts are in strand coordinates as in PSL. */
#define BK_MAX_BLOCKS 32
typedef struct bk_psl {
    int32_t matches, mismatches, rep_matches, n_count;
    int32_t q_num_insert, q_base_insert, t_num_insert, t_base_insert;
    int32_t strand;             /* '+' or '-' */
    int32_t q_size, q_start, q_end;
    int32_t t_index;            /* 0 = target window, 1.. = partner windows */
    int32_t t_size, t_start, t_end;
    int32_t block_count;
    int32_t block_sizes[BK_MAX_BLOCKS], q_starts[BK_MAX_BLOCKS], t_starts[BK_MAX_BLOCKS];
    int32_t score;
} bk_psl;
int bk_get_hits(bk_handle *h, int32_t region, int32_t contig, bk_psl *hits, int32_t cap);
/* The same records without a block limit, as a flat int32 stream: per record BK_PSL_FLAT_HEAD scalars -- the fields of bk_psl
 * up to t_end in its order, then score and block_count -- followed by block_count block sizes, q starts and t starts.
 * *needed = ints of the whole stream (call with cap = 0 to size the buffer); returns the number of records. */
#define BK_PSL_FLAT_HEAD 18
int bk_get_hits_flat(bk_handle *h, int32_t region, int32_t contig, int32_t *buf, size_t cap, size_t *needed);

/* ---- SV-call tail (C1-C3): contig + PSL-equivalent records -> the 13-field result row -------------------
 * Replaces, per contig, contig.query_ref/check_target_blat/make_calls -> align_manager(meta_dict).get_result()
 * (sv_processor.py:823-866, sv_caller.py:785-833).  Host code; same semantics as breakmer_amd/sv_caller.py.
 * The context is a line-based text (breakmer_amd/call_context.py): options (breakmer.py:73-86), gene table
 * (utils.py:727-773), repeat masks (utils.py:302-353) and per region query_region / discordant pairs
 * (sv_processor.py:376-408) / placement of the partner windows / read-id classes.  A line `keep_tables` in place of the
 * gene / repeat lines keeps the tables of the handle's previous context (they are the same for every batch of a run). */
int bk_set_call_context(bk_handle *h, const char *text);
int bk_call(bk_handle *h);                                      /* after bk_run: calls for every contig of the batch */
int bk_call_async(bk_handle *h);                                /* the same on the handle's own thread (returns at once; every later call on the handle
                                                                 * waits for it first and reports its error; bk_call / bk_get_calls then find the calls made) */
int bk_get_calls(bk_handle *h, char *buf, size_t cap, size_t *needed);   /* "<region>\t<contig>\t<13 fields>\n" ... */
int bk_call_text(const char *text, char *out, size_t cap, int *target_hit);   /* one fully described contig; no GPU needed */

/* ---- genome-wide seed lookup (the step behind the reference's whole-genome fallback: a contig its target window does not explain
 * is handed to a gfServer over the whole genome, sv_processor.py:829-831, utils.py:620-657; BLAT finds its candidate loci through an
 * index of the genome's tiles).  Here: the caller samples the genome's k-mers (k <= 16, 2 bit/base in a uint32, every `step`-th
 * position) and sorts them once (breakmer_amd/refseq.GenomeIndex); the sorted codes live in HBM (1.5 GB for a 3 Gb genome at
 * step 8) and every query k-mer is looked up by binary search on the device: lo[i], hi[i] = the range of index entries equal to
 * queries[i] (lo == hi: absent).  The caller turns ranges into loci (>= 2 hits in one diagonal band = BLAT's -minMatch=2). */
typedef struct bk_index bk_index;
int bk_index_create(int device_id, const uint32_t *sorted_codes, uint64_t n, bk_index **out);
int bk_index_probe(bk_index *ix, const uint32_t *queries, uint64_t n_queries, uint32_t *lo, uint32_t *hi, float *kernel_ms);
/* The whole look-up on the device (round 5): with the sequence number and position of every index entry resident as well
 * (bk_index_set_loci: arrays parallel to sorted_codes), bk_index_find returns the LOCI of one query sequence -- the k-mers of one
 * strand of a contig segment (queries[i] = k-mer at query position i, ok[i] = 0 where it holds an N; at most 32,768) --: index hits
 * of k-mers that occur at most max_occ times, sorted by (sequence, diagonal, position), cut where the sequence changes or the
 * diagonal jumps by more than `band`, kept with >= min_hits hits (2 = BLAT's -minMatch=2), in that order; start / end = first and
 * last index position of the locus.  *n_loci may exceed cap.  Replaces the numpy clustering of refseq.GenomeIndex.find, which
 * stays as the host path it is tested against. */
typedef struct bk_locus { uint32_t hits, seqno, start, end; } bk_locus;
int bk_index_set_loci(bk_index *ix, const uint16_t *seqno, const uint32_t *pos);
int bk_index_find(bk_index *ix, const uint32_t *queries, const uint8_t *ok, uint32_t n_queries, uint32_t max_occ, uint32_t band, uint32_t min_hits,
                  bk_locus *loci, uint32_t cap, uint32_t *n_loci, float *kernel_ms);
int bk_index_destroy(bk_index *ix);

/* The 2-bit packing bk_submit_regions applies to every sequence (16 bases per word, first base in the most significant
 * bits; replaces the FASTA/FASTQ text the reference writes for Jellyfish and the assembler, utils.py:355-381), on one
 * sequence: `words` receives n_words words (zero padded), an 'N' is packed as A and its position appended to n_pos
 * (at most cap; *n_n = how many there are).  flags: BK_SUBMIT_READ_CODES = the bytes are base codes.  Returns BK_E_ARG for
 * any other character.  No GPU needed (host utility; lets the SIMD and the table path be checked against each other). */
int bk_pack_sequence(const char *seq, int32_t len, uint32_t flags, uint32_t *words, int32_t n_words,
                     uint32_t *n_pos, int32_t cap, int32_t *n_n);

/* batched olc.nw on explicit pairs (known-answer tests, DP micro-benchmark): out = 4 ints per pair
 * (j_start, i_end, i_start, score); transposed = 1 uses the sweep the assembler uses for nw(read, contig), 2 the
 * suffix-restricted direct sweep it uses for nw(contig, read); 3 / 4: both DPs on one wavefront (bk_nw_dual), result of nw(seq1, seq2) /
 * nw(seq2, seq1); 5..8: one score matrix with both tie-break orders, two pairs per wavefront (bk_nw_pair): pair i in one half (5: nw(seq1,
 * seq2), 6: nw(seq2, seq1)), pair i+1 in the other (7, 8: ITS two results are returned in out[i]).  9..14: the score sweep of ABI 4
 * (bk_nw_score_c: one plain score matrix per read, the border cell without a traceback where the end cell's score equals its
 * diagonal, the full sweep for the rest) -- 9..12 like 5..8 with two reads per wavefront, 13 / 14 like 3 / 4 with one; 15: the
 * score sweep alone as the assembler calls it: out = (v1.j_start or -1 = needs the full sweep, v1.score, v2.j_start or -1, v2.score);
 * 16 / 17: the score sweep alone (two reads / one read per wavefront) for timing.  Same results by construction; the tests compare them all */
int bk_nw_batch(bk_handle *h, const char *seqs, size_t seq_bytes, const uint32_t *off1, const uint32_t *len1,
                const uint32_t *off2, const uint32_t *len2, int32_t n_pairs, int32_t reps, int32_t transposed, int32_t *out, float *ms);

/* bookkeeping for measurement: algorithmic work of the last bk_run
 *   which = 0: olc.nw DP cells (sum len(seq1)*len(seq2)), 1: olc.nw calls, 2: SW cells,
 *           3: algorithmic HBM bytes (SURVEY 8d formula), 4: unique reads, 5: sample k-mers, 6: contigs,
 *           20/21: host packing / host-to-device copy time of the last bk_submit_regions (microseconds),
 *           22: regions of the last run that failed on a device cap (bk_get_region_status),
 *           26: regions the library re-ran with larger assembler caps (unless bk_config.no_escalation) */
int bk_get_stat(bk_handle *h, int which, uint64_t *value);

#ifdef __cplusplus
}
#endif
#endif

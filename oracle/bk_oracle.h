/* TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of BreaKmer's per-region hot path.
 *
 * Plain C restatement of the reference algorithm, function by function, citing the
 * reference file:line each one follows.  Pinned against golden vectors generated
 * from the real reference (tests/golden/, tools/make_golden.py).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (breakmer_amd/) never does.
 */
#ifndef BK_ORACLE_H
#define BK_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* olc.nw (olc.py:40-107).  seq1/seq2 are ASCII.  out7 = {len(align1), len(align2)=same,
 * m, j_start, i_end, i_start, score}; align1/align2 (optional, may be NULL) receive the
 * gapped strings, caller allocates m+n+1 bytes each. */
void bko_nw(const char *seq1, int m, const char *seq2, int n, int *out7, char *align1, char *align2);
/* number of DP cells computed by bko_nw since the last reset (SURVEY 8d "C") */
uint64_t bko_cells(int reset);
uint64_t bko_nw_calls(int reset);

/* T1: group identical read sequences (utils.py:239-244): out_rep[u] = index of first read with
 * that sequence (FASTQ order), out_n[u] = number of reads; returns U. reads are n fixed-stride
 * ASCII rows (stride bytes apart) with lengths lens[i]. */
int bko_group_reads(const char *reads, int stride, const int *lens, int n, int *out_rep, int *out_n);

/* K1+K2: Jellyfish-style occurrence counts (utils.py:151-178, 287-296; no -C: strand specific)
 * and sample-only set algebra (sv_processor.py:609-631).
 *   case  = k-mers of all n reads (every position, duplicates counted)
 *   sc    = k-mers of the soft-clip sequences; if nsc < 0 then case_sc := case
 *   ref   = k-mers of each of the nref reference strings, forward and reverse complement
 * Result: sample-only k-mers sorted ascending by string, out_mers (cap*k bytes, no NUL),
 * out_counts; returns the number found (may exceed cap: then only cap are written). */
int bko_kmer_select(const char *reads, int stride, const int *lens, int n,
                    const char *sc, int sc_stride, const int *sc_lens, int nsc,
                    const char *const *refs, const int *ref_lens, int nref,
                    int k, char *out_mers, int *out_counts, int cap);

/* A1..A12: sv_assembly.init_assembly (sv_assembly.py:30-63) on grouped reads.
 *   useqs/ulens/unreads/uindel: U unique reads in fq_recs iteration order (first occurrence),
 *   mers/counts: M sample-only k-mers (any order), k, rc_thresh, read_len.
 * Returns an opaque result; query with the accessors; free with bko_asm_free. */
typedef struct bko_asm bko_asm;
bko_asm *bko_init_assembly(const char *useqs, int stride, const int *ulens, const int *unreads,
                           const uint8_t *uindel, int U,
                           const char *mers, const int *counts, int M,
                           int k, int rc_thresh, int read_len);
/* 1: find_reads answers from a per-region k-mer -> (read, first position) index instead of scanning every read per visit
 * (same result, checked in tests; needed for full-size BASELINE configs[4] regions); default 0 = the reference's scan */
void bko_set_find_index(int on);
int bko_asm_ncontigs(const bko_asm *a);
int bko_asm_contig_len(const bko_asm *a, int c);
int bko_asm_contig_clen(const bko_asm *a, int c);          /* len(counts.others) (Q8: may differ) */
int bko_asm_contig_nkmers(const bko_asm *a, int c);
int bko_asm_contig_nreads(const bko_asm *a, int c);
void bko_asm_contig_get(const bko_asm *a, int c, char *seq, int *indel_only, int *others,
                        int *kmer_locs, int *kmer_idx /* index into input mers */, int *read_idx /* sorted unique-read idx */);
/* per-read final flags after assembly: bit0 used, bit1 deleted from fq_recs */
void bko_asm_read_flags(const bko_asm *a, uint8_t *flags);
void bko_asm_free(bko_asm *a);

/* ------------------------------------------------------------------ R2: contig -> window realignment
 * Replaces the external BLAT call (sv_processor.py:835-851).  BLAT's source is not in the reference
 * tree and the binary is absent: PARITY WITH BLAT IS UNPINNED.  This function DEFINES the contract
 * the HIP realign kernel must reproduce bit for bit (DESIGN.md "Realign contract"):
 *   1. iterated gap-free local Smith-Waterman (maximal scoring segment; match +1, mismatch -2) of
 *      the still-unaligned query intervals (>= min_seg bases) against every target on both strands;
 *      best hit by (score desc, target index asc, '+' first); accepted iff score >= min_score;
 *      end cell = max score, then smallest query end, then smallest target end;
 *   2. every hit is one ungapped block (as BLAT's blocks are); gaps arise only from chaining;
 *   2b. BLAT's published seeding rule (sv_processor.py:843: -stepSize=10 -minMatch=2, tile 11): a step-1 hit or a secondary
 *      alignment (step 5) is kept only if it holds TWO index tiles -- target positions 10 j .. 10 j + 10 inside it, all eleven
 *      bases matching the query; the query bases of a dropped step-1 hit stay unaligned (island fill, step 4, may still recover
 *      them next to a chained anchor, as BLAT's extension does).  Without it a 20-29 base stretch copied from elsewhere (a
 *      templated insertion) became a record of its own and split an indel contig into a "rearrangement";
 *   3. hits ordered by query position are chained into one PSL record when they are on the same
 *      target and strand and collinear (a target overlap smaller than half of either hit is trimmed
 *      from the later hit); a chain passes over a hit that is not collinear with it (that hit gets a record of its own), as
 *      BLAT's chaining does;
 *   4. island fill: the unaligned rectangle between two chained blocks (query AND target bases left over: an indel next
 *      to another difference, whose short flank cannot anchor a segment of its own) and the rectangles beyond the first
 *      and the last block are searched for the best gap-free segment on a diagonal within 16 of a neighbouring block's
 *      diagonal; it becomes a block of the record if it scores >= 8, and the two rectangles it leaves are filled the
 *      same way (ties: higher score, smaller query end, smaller target end).  tests/golden/realign_evidence.json holds
 *      what the reference's caller makes of these records next to BLAT-style records built from the known edits.
 *   5. secondary alignments (BLAT prints EVERY alignment scoring >= -minScore, sv_processor.py:843, and the caller
 *      counts them per query base: hit_freq sv_caller.py:593-594, mean_cov :616/:631/:676, check_uniqueness :430-432,
 *      check_previous_add :55-72): every diagonal of every (target, strand) is walked over the WHOLE query with
 *      H = max(0, H + s); each positive excursion (reset to reset / end of the diagonal) yields the segment from its start
 *      to the first position of its highest H; it is reported iff that peak is >= min_score and the segment does not
 *      overlap a step-1 hit on the same target, strand and diagonal (that is the same alignment).  One-block records,
 *      after the chained ones, ordered by (score desc, target index asc, '+' first, query end asc, target end asc).
 *      Secondary segments are NOT chained with each other (BLAT would chain collinear ones): hit_freq is the same
 *      either way, the number of records is not.
 *   6. placement of ambiguous hits (applied before the chaining of step 3): when a step-1 hit has equal alternatives --
 *      a secondary alignment that covers its query interval and scores the same over it, e.g. a duplicated flank -- step
 *      1's tie-break (smallest target end) is arbitrary, while BLAT reports the chain with the smaller gaps.  The hits in
 *      forward query order each choose among {the hit, its alternatives in step-5 order} such that the number of chain
 *      breaks between consecutive hits, then the sum of |diagonal shift| between chained neighbours, is smallest (ties:
 *      the earlier candidate).  A chosen alternative becomes the hit, restricted to the hit's query interval; the hit it
 *      replaces is listed with the secondary alignments, the secondary it came from is dropped.
 *   7. soft-masked windows: see below (rep_matches).
 *   8. BLAT's documented OUTPUT FILTERS (round 5), applied to every record -- chained or secondary -- before it is reported
 *      (bko_psl_passes).  The reference's command lines are `blat -t=dna -q=dna -out=psl -minScore=20 -stepSize=10 -minMatch=2
 *      -repeats=lower` (sv_processor.py:843) and `gfClient -t=dna -q=dna -out=psl -minScore=20` (:840); BLAT's usage text says
 *      "-minScore=N sets minimum score.  This is the matches minus the mismatches minus some sort of gap penalty" and
 *      "-minIdentity=N Sets minimum sequence identity (in percent).  Default is 90 for nucleotide searches" (the reference
 *      passes none, so 90 applies).  Here:
 *          matches + repMatches - misMatches - qNumInsert - tNumInsert >= min_score, and
 *          100 - milliBad / 10 >= 90 with the milliBad the reference's caller computes for a record (sv_caller.py:954-968,
 *          UCSC's pslCalcMilliBad for DNA): 10 * (misMatches + qNumInsert + round(3 ln(1 + max(0, qAli - tAli)))) <= total.
 *      What it changes: a low-identity alignment (a diverged copy of a flank: 80 matches / 20 mismatches scores +40 under
 *      +1/-2 and was reported; BLAT drops it at 80 %) no longer counts in hit_freq / mean_cov / check_uniqueness
 *      (sv_caller.py:593-594, 631, 430-432).  What it cannot change: the step-1 acceptance score.  A segment BLAT can SEED
 *      (step 2b: two perfect 11-base tiles) that passes both filters always scores >= min_score under +1/-2 as well: with s =
 *      m - 2x, identity >= 90 % means 9x <= m; two adjacent tiles are a perfect run of 21 (s >= 21 on its own), two separated
 *      tiles enclose >= 9 bases of which at most x mismatch, so m >= 29 - x + ... and s < 20 would need x <= 2, m <= 23: impossible.
 *      So "+1/-2 score >= 20" is implied for everything BLAT would print, and lowering it would add nothing.
 * An N (in the contig or in a window) matches nothing.  A lower-case (soft-masked) window base is the same base; a match on
 * one is counted in rep_matches instead of matches (BLAT's -repeats=lower, sv_processor.py:843).
 * Output: PSL-equivalent records (fields consumed by sv_caller.py:911-936).  Returns the number of records (may exceed
 * cap), or -2 when a chained record would need more than BKO_MAX_BLOCKS blocks (the library reports that per region). */
#define BKO_MAX_BLOCKS 512     /* the library's bk_call has no limit; its bk_psl (bk_get_hits) holds 32 */
typedef struct bko_psl {
    int32_t matches, mismatches, rep_matches, n_count;
    int32_t q_num_insert, q_base_insert, t_num_insert, t_base_insert;
    int32_t strand;
    int32_t q_size, q_start, q_end;
    int32_t t_index, t_size, t_start, t_end;
    int32_t block_count;
    int32_t block_sizes[BKO_MAX_BLOCKS], q_starts[BKO_MAX_BLOCKS], t_starts[BKO_MAX_BLOCKS];
    int32_t score;
} bko_psl;
int bko_realign(const char *contig, int qlen, const char *const *targets, const int *tlens, int ntargets,
                int min_score, int min_seg, bko_psl *out, int cap);
int bko_psl_passes(const bko_psl *r, int min_score);      /* step 8: would BLAT print this record (-minScore, -minIdentity default 90) */
uint64_t bko_sw_cells(int reset);

#ifdef __cplusplus
}
#endif
#endif

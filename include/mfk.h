/*
 * mfk.h -- C-ABI of libmfk.so: the MI355X-native flank-discovery kernels ("mustache flank kernels").
 *
 * This is the drop-in boundary for the `mustache findflanks` -> `mustache pairflanks` hot path.  The reference
 * (durrantmm/mustache, pure Python) has no FFI for this path; its boundary is the two Python stage drivers
 *     _findflanks(...)   mustache/findflanks.py:18-59
 *     _pairflanks(...)   mustache/pairflanks.py:18-51
 * and the classes they drive (SoftclipParser findflanks.py:62, SoftclipConsensus findflanks.py:862,
 * FlankPairer pairflanks.py:54).  Each entry point below names the reference code it replaces.  A ctypes binding is
 * what a maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no exceptions: every function returns 0 on success, non-zero on error;
 *     mfk_last_error(ctx) gives the message.
 *   - Array pointers inside mfk_reads are DEVICE pointers (e.g. torch.Tensor.data_ptr()), except the two `h_` header
 *     arrays which are host pointers.  The caller owns every input buffer; result buffers are caller-allocated and
 *     filled by the *_fetch call after the compute call has reported their sizes (two-call protocol).
 *   - One ctx per device; calls on one ctx are serialised by the caller.  A ctx owns a HIP stream and a growable
 *     device workspace.
 *   - Positions are 0-based reference coordinates, exactly as the reference prints them.
 */
#ifndef MFK_H
#define MFK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* flags byte of a packed read */
#define MFK_LKIND(f) ((f) & 3)          /* 0 none, 1 CIGAR S, 2 MD terminal mismatch (sctools.py:25-31)  */
#define MFK_RKIND(f) (((f) >> 2) & 3)   /* same for the right end (sctools.py:17-23)                      */
#define MFK_FLAG_COMPLEX 16             /* more than one aligned block: see cx_ / blk_ tables            */
#define MFK_FLAG_POOL 32                /* query sequence + qualities are in the pool                     */

/* One lenient-clipped read (flags & 15 != 0) as findflanks' event kernel wants it: the fields of the nine columns it
 * would otherwise gather one sparse 64-byte sector at a time, in 16 bytes.  Optional, see mfk_reads.clip_rec. */
typedef struct mfk_clip_rec {
    uint32_t read;                 /* index of the read in the columns                                   */
    int32_t ref_start, ref_end;
    uint16_t contig_id;
    uint8_t mapq;
    uint8_t inner;                 /* min(255, qlen - left clip length - right clip length) when both ends are
                                      clipped (sctools.py:166-180, lengths as in sctools.py:111-142), else 255 */
} mfk_clip_rec;
#define MFK_CLIP_TILE 4096         /* reads per entry of mfk_reads.clip_tile                             */

/* Packed alignments, coordinate-sorted, 22 bytes per read in nine SoA columns (see DESIGN.md "Data layout"). */
typedef struct mfk_reads {
    int64_t n_reads;
    const int32_t *ref_start;      /* pysam reference_start                                              */
    const int32_t *ref_end;        /* pysam reference_end (exclusive)                                    */
    const int32_t *link;           /* previous read with the same query name, or -1                      */
    const uint16_t *contig_id;
    const uint16_t *qlen;          /* len(query_sequence)                                                */
    const uint16_t *first_op_len;  /* cigartuples[0][1]  (sctools.py:5-6)                                 */
    const uint16_t *last_op_len;   /* cigartuples[-1][1] (sctools.py:8-9)                                 */
    const uint8_t *mapq;
    const uint8_t *flags;
    int32_t n_contigs;
    int32_t max_span;              /* max(ref_end - ref_start) over all reads                            */
    const int32_t *h_contig_len;         /* HOST, n_contigs   (BAM header LN)                            */
    const int64_t *h_contig_read_start;  /* HOST, n_contigs+1 (first read of each contig)                */
    int64_t n_cx;                  /* reads with FLAG_COMPLEX                                            */
    const uint32_t *cx_read;       /* their read indices, ascending                                      */
    const uint32_t *cx_off;        /* n_cx+1 offsets into blk_*                                          */
    int64_t n_blk;
    const int32_t *blk_start;      /* pysam get_blocks()                                                 */
    const int32_t *blk_end;
    const int32_t *blk_ins;        /* query bases in front of the block that are not in a block: insertions (and, for
                                      the first block, anything but a leading CIGAR S)                    */
    int64_t n_pool;                /* reads with FLAG_POOL                                               */
    const uint32_t *pool_read;     /* their read indices, ascending                                      */
    const uint64_t *pool_off;      /* n_pool+1 base offsets, all even (reads start on byte boundaries)   */
    const uint8_t *pool_bases;     /* 4-bit BAM base codes "=ACMGRSVTWYHKDBN", base k in byte k>>1, even k in the high
                                      nibble (the BAM seq encoding)                                      */
    const uint8_t *pool_quals;     /* one byte per base                                                  */
    /* Optional dense copy of the clipped reads (has no counterpart in the reference; clip_rec == NULL = columns only,
     * same results).  mfk_findflanks / mfk_flankpairs / mfk_count_* read it instead of gathering the columns.  The
     * caller guarantees that it restates the columns: record k is the k-th read with flags & 15 != 0.               */
    int64_t n_clip;                /* number of reads with flags & 15 != 0                               */
    const mfk_clip_rec *clip_rec;  /* DEVICE, n_clip records in read order, 16-byte aligned; or NULL      */
    const uint32_t *clip_tile;     /* DEVICE, (n_reads + 4095) / 4096 + 1 entries: clip_tile[t] = number of such reads
                                      among reads [0, 4096 t)                                            */
} mfk_reads;

/* CLI parameters of both commands (mustache/main.py:68-78, 102-107). */
typedef struct mfk_params {
    int32_t min_softclip_length;         /* -minlen   */
    int32_t min_softclip_count;          /* -mincount */
    int32_t min_alignment_quality;       /* -minq     */
    int32_t min_alignment_inner_length;  /* -minial   */
    int32_t min_distance_to_mate;        /* -mindist  */
    int32_t large_insertion_cutoff;      /* -lins     */
    int32_t min_count_consensus;         /* -mcc      */
    int32_t max_direct_repeat_length;    /* -maxdr    */
    double min_softclip_ratio;           /* -minratio */
    double max_indel_ratio;              /* -maxir    */
    double max_junction_spanning_prop;   /* -maxjsp   */
} mfk_params;

/* One soft-clip site (the reference's SoftclipSite, findflanks.py:684-859), for stage-level inspection. */
typedef struct mfk_site {
    int32_t contig_id;
    int32_t pos;
    int32_t softclip_count_5p, softclip_count_3p, runthrough_count;
    int32_t small_insertion_count_5p, small_insertion_count_3p;
    int32_t large_insertion_count_5p, large_insertion_count_3p;
    int32_t deletion_count, upstream_deletion_count, downstream_deletion_count;
    int32_t keep_5p, keep_3p;
} mfk_site;

/* One row of the findflanks table (make_dataframe, findflanks.py:631-670). */
typedef struct mfk_flank {
    int32_t contig_id;
    int32_t pos;
    int32_t orient;                /* 5 or 3 */
    int32_t softclip_count_5p, softclip_count_3p, runthrough_count;
    int32_t small_insertion_count_5p, small_insertion_count_3p;
    int32_t large_insertion_count_5p, large_insertion_count_3p;
    int32_t deletion_count, upstream_deletion_count, downstream_deletion_count;
    int32_t total_count;
    int32_t consensus_softclip_count;
    int32_t seq_len;
    int64_t seq_off;               /* consensus_seq = seqs[seq_off .. seq_off+seq_len), ASCII */
} mfk_flank;

/* One row of the pairflanks table (get_header_list, pairflanks.py:125-130). */
typedef struct mfk_pair {
    int32_t contig_id;
    int32_t pos_5p, pos_3p;
    int32_t index_5p, index_3p;    /* rows of the flank table (quirk P2) -> seq_5p / seq_3p */
    int32_t softclip_count_5p, softclip_count_3p;
    int32_t total_count_5p, total_count_3p;
    int32_t spanning_count;
    int32_t has_ir;
    int32_t ir_length;
    int32_t ir5_len, ir3_len;      /* -1 = None */
    int32_t dr_len;                /* len(direct_repeat_reads_consensus) = pos_5p - pos_3p - 1 */
    int32_t dr_ref_len;            /* len(direct_repeat_reference): same, unless clamped at a contig end */
    int64_t ir5_off, ir3_off;      /* into strs */
    int64_t dr_ref_off, dr_reads_off;
} mfk_pair;

/* stop_after values for mfk_findflanks (the call sequence of _findflanks, findflanks.py:37-49) */
enum {
    MFK_STAGE_PARSE = 1,      /* parse_softclips                 findflanks.py:100-121 */
    MFK_STAGE_MINLENGTH = 2,  /* filter_softclips_minlength      findflanks.py:124-145 */
    MFK_STAGE_MINCOUNT = 3,   /* filter_softclips_mincount       findflanks.py:148-169 */
    MFK_STAGE_MINDIST1 = 4,   /* filter_softclips_mindistance    findflanks.py:171-200 */
    MFK_STAGE_UNCLIPPED = 5,  /* parse_unclipped_read_info       findflanks.py:349-396 */
    MFK_STAGE_RATIOS = 6,     /* filter_softclips_count_ratios   findflanks.py:203-239 */
    MFK_STAGE_MINDIST2 = 7,   /* filter_softclips_mindistance (second call)            */
    MFK_STAGE_CONSENSUS = 8,  /* make_consensus_sequences + 3 consensus filters, findflanks.py:512-532, 241-342 */
    MFK_STAGE_ALL = 9         /* third filter_softclips_mindistance + make_dataframe   */
};
#define MFK_N_STATS 16
/* stats[0..5]: sites remaining after parse, minlength, mincount, mindistance#1, ratios, mindistance#2;
 * stats[6..8]: consensus sequences after the minlength, mincount and best-of filters; stats[9]: sites after
 * mindistance#3; stats[10]: soft-clip events; stats[11]: flank rows. (the reference logs these, findflanks.py:144...)
 * stats[12]: work counter for bench.py: columns x 64-bit limbs of the bit-vector edit-distance recurrences that ran. */

typedef struct mfk_ctx mfk_ctx;

/* lifecycle ------------------------------------------------------------------------------------------------ */
mfk_ctx *mfk_create(int device_id);           /* NULL if no HIP device / out of memory */
void mfk_destroy(mfk_ctx *ctx);
const char *mfk_last_error(mfk_ctx *ctx);
int mfk_version(void);

/* findflanks: replaces SoftclipParser driven as in _findflanks (findflanks.py:24-51).
 * Runs stages 1..stop_after on the resident arrays.  On return *n_sites is the number of surviving sites, and (when
 * stop_after == MFK_STAGE_ALL) *n_flanks / *seq_bytes size the flank table. */
int mfk_findflanks(mfk_ctx *ctx, const mfk_reads *reads, const mfk_params *params, int stop_after,
                   int64_t *n_sites, int64_t *n_flanks, int64_t *seq_bytes);
/* copies results of the last mfk_findflanks; any pointer may be NULL */
int mfk_findflanks_fetch(mfk_ctx *ctx, mfk_site *sites, mfk_flank *flanks, char *seqs, int64_t *stats);

/* pairflanks: replaces FlankPairer.run_pair_flanks (pairflanks.py:96-122).
 * flanks / flank_seqs: HOST flank table (rows in file order).  h_contig_rank[c] = rank of contig c's name in string
 * order (the final sort_values(['contig', ...]) of pairflanks.py:248 sorts names).  genome: DEVICE bytes of the FASTA
 * sequences concatenated in contig order (case preserved, quirk P10), h_genome_off: HOST n_contigs+1 offsets. */
int mfk_pairflanks(mfk_ctx *ctx, const mfk_reads *reads, const mfk_params *params, const mfk_flank *flanks,
                   const char *flank_seqs, int64_t n_flanks, const int32_t *h_contig_rank, const uint8_t *genome,
                   const int64_t *h_genome_off, int64_t *n_pairs, int64_t *str_bytes);
int mfk_pairflanks_fetch(mfk_ctx *ctx, mfk_pair *pairs, char *strs, int64_t *stats);
/* pairflanks stats[0]: candidate pairs (pair_all_nearby_flanks), [1]: assigned, [2]: with IR, [3]: spanning_count>0,
 * [4]: after filter_junction_spanning. */

/* findflanks followed by pairflanks on the same resident reads in ONE call: what the workflow does per sample with two
 * processes and a TSV in between (Snakefile:64-109: rule findflanks, then rule pairflanks on its output).  The flank
 * table of the first stage goes to the second without leaving the library; both results are then fetched with
 * mfk_findflanks_fetch / mfk_pairflanks_fetch exactly as after the separate calls, and are the same bytes.  The reads
 * must carry the sequence pool both stages need (pool = clipped reads and the reads over direct-repeat windows, or all). */
int mfk_flankpairs(mfk_ctx *ctx, const mfk_reads *reads, const mfk_params *params, const int32_t *h_contig_rank,
                   const uint8_t *genome, const int64_t *h_genome_off, int64_t *n_sites, int64_t *n_flanks,
                   int64_t *seq_bytes, int64_t *n_pairs, int64_t *str_bytes);

/* Inverted-repeat results from an external detector for the NEXT mfk_pairflanks call on this ctx -- the escape hatch
 * for sites that have EMBOSS einverted, which is what the reference runs (embosstools.py:8-26, pairflanks.py:172-219).
 * One row per candidate pair (pair_all_nearby_flanks, pairflanks.py:138-169): flank row indices and ir = has_ir,
 * arm1_start, arm1_end, arm2_start, arm2_end (1-based inclusive, after the acceptance loop of pairflanks.py:196-205;
 * arm1 = IR_3p, arm2 = IR_5p).  A candidate without a row makes mfk_pairflanks fail with code 8.  n = 0 clears. */
int mfk_pairflanks_set_ir(mfk_ctx *ctx, int64_t n, const int32_t *index_5p, const int32_t *index_3p, const int32_t *ir);

/* inverted-repeat detection on its own (replaces one einverted run, embosstools.py:8-26 + pairflanks.py:183-213).
 * seqs: HOST concatenated upper-case sequences, off: n+1 offsets.  out: n x 6 int32:
 * has_ir, arm1_start, arm1_end, arm2_start, arm2_end (1-based inclusive, einverted convention), n_reported. */
int mfk_ir_detect(mfk_ctx *ctx, const char *seqs, const int64_t *off, int64_t n, int32_t *out);

/* recall: per-site read tallies at GIVEN positions (replaces Recaller.parse_clipped_and_unclipped_read_info,
 * recall.py:50-112, with get_reads_at_site / get_unclipped_read_info_at_site of findflanks.py:399-452).  Sites are host
 * arrays (contig id, 0-based position, -1 .. contig length); counts_out: n_sites x 10 int32 in the order softclip_5p,
 * softclip_3p, runthrough, small_insertion_5p, small_insertion_3p, large_insertion_5p, large_insertion_3p, deletion,
 * upstream_deletion, downstream_deletion (distinct query names each).  Uses min_alignment_quality,
 * min_alignment_inner_length and large_insertion_cutoff of params. */
int mfk_recall(mfk_ctx *ctx, const mfk_reads *reads, const mfk_params *params, const int32_t *h_contig_id, const int32_t *h_pos,
               int64_t n_sites, int32_t *h_counts_out);

/* mfk_recall for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  NO COUNTERPART IN THE REFERENCE (which has no device memory to manage); same arguments, meaning and results
 * as mfk_recall.  The call allocates nothing on the device or the host (no hipMalloc / hipFree / hipHostMalloc, no arena
 * growth): the contig header, the staged site list and the device-side counts live in a fixed array in the library's
 * code object (852,224 bytes per GPU, held until the process ends), the pinned staging block is allocated by mfk_create.
 * Limits: n_sites <= 16384 and reads->n_contigs <= 4096 per call, otherwise return code 9 with a message.  All calls
 * on one GPU share that array, whatever ctx they use: calls that overlap on one GPU race; serialise them. */
int mfk_recall_noalloc(mfk_ctx *ctx, const mfk_reads *reads, const mfk_params *params, const int32_t *h_contig_id, const int32_t *h_pos,
                       int64_t n_sites, int32_t *h_counts_out);

/* inferseq-overlap: best ungapped sliding overlap of the end of the 5p flank with the start of the 3p flank (replaces
 * get_best_sliding_alignment, inferseqoverlap.py:94-164, called from find_overlap 73-91).  seqs5 / seqs3: HOST
 * concatenated sequences with n+1 offsets each (bytes are compared as they are).  out: n x 6 int32 = best_score,
 * best_score_mismatches, best_r_start, best_r_end, best_q_start, best_q_end (q = 5p flank, r = 3p flank; half-open).
 * An empty sequence is the reference's IndexError: return code 7. */
int mfk_overlap(mfk_ctx *ctx, const char *seqs5, const int64_t *off5, const char *seqs3, const int64_t *off3, int64_t n, int32_t *out);

/* inferseq-reference / inferseq-assembly: pairing of the flank alignments (replaces what InferSequence does after the
 * aligner has run: __prefilter_reads inferseq.py:103-126, AlignedPairs.add_read / match_forward_reverse 214-290,
 * filter_smallest_overlapping 321-362, filter_pairs_max_internal_softclip_prop 364-380, filter_best_alignments 305-318,
 * filter_pairs_size 382-390, in the order of infer_sequences 50-70).  The aligner (bowtie2 --local --all) stays
 * external, as in the reference.  One record per alignment of a flank, HOST array in BAM order: */
typedef struct mfk_aln {
    int32_t group;                 /* dense id of query_name.split('_')[0] (the pair)                    */
    int32_t name;                  /* dense id of the whole query_name                                   */
    int32_t mate;                  /* 1 if query_name.split('_')[-1] == '1', else 2                      */
    int32_t is_reverse;
    int32_t contig_id;
    int32_t ref_start, ref_end;
    int32_t contig_len;            /* len(genome_dict[reference_name]) (inferseq.py:38, 121)             */
    int32_t qlen;                  /* len(query_sequence)                                                */
    int32_t left_clip, right_clip; /* CIGAR S length at that end, 0 if none (sctools.py:57-67, 78-97)     */
    int32_t aln_len;               /* query_alignment_length                                             */
    int32_t nm, as;                /* NM and AS tags                                                     */
} mfk_aln;
/* *n_pairs = surviving (forward, reverse) alignment pairs; fetch gives their indices into alns, ordered by group and,
 * inside a group, by creation order (forward reads of mate 1, of mate 2, then reverse reads of mate 1, of mate 2, each
 * in BAM order).  Two behaviours of the reference depend on set iteration order; the build fixes them: duplicates keep
 * the pair created first, ties of get_smallest_window_pair go to the pair created first.  A read with
 * aln_len == 0 is the reference's ZeroDivisionError: return code 7. */
int mfk_inferseq_pairs(mfk_ctx *ctx, const mfk_aln *alns, int64_t n, double min_perc_identity, double max_internal_softclip_prop,
                       int64_t max_inferseq_size, int64_t min_inferseq_size, int64_t *n_pairs);
int mfk_inferseq_pairs_fetch(mfk_ctx *ctx, int32_t *forward, int32_t *reverse);

/* inferseq-database: the pairing of flank alignments against a database of known elements (replaces what
 * infer_sequences_database does after the aligner has run, inferseqdatabase.py:116-139: prefilter_reads 142-180, get_pairs
 * 183-208 with reads_mapped_both_ends 211-232, match_pairs 269-300 and pair_mapped_both_ends 234-266,
 * keep_best_alignment_score 303-318).  Same records as mfk_inferseq_pairs (mate = 1 for flank id '1', else 2).
 * *n_pairs = surviving couples; mfk_inferseq_pairs_fetch gives (forward-strand read, reverse-strand read) indices into alns,
 * ordered by group, inside a group by the contig's first kept alignment, then in the order match_pairs takes couples
 * (distance of the positions descending, flank-1 read, flank-2 read).  An alignment with aln_len == 0 or qlen == 0 is the
 * reference's ZeroDivisionError: return code 7. */
int mfk_inferseq_db_pairs(mfk_ctx *ctx, const mfk_aln *alns, int64_t n, double min_perc_identity, double max_internal_softclip_prop,
                          int64_t max_edge_distance, int64_t *n_pairs);

/* multi-GPU merge: all-gather of one table per rank over RCCL (xGMI inside a node).  NO COUNTERPART IN THE REFERENCE: its
 * processes never communicate (the Snakemake workflow schedules independent jobs, workflow/Snakefile); this is the one
 * exchange step of the sharded path: the flank tables of the coordinate / contig shards before
 * pairflanks, the pair tables after it, or the per-sample site tables of a cohort.  RCCL is bound at run time (dlopen; a copy
 * already in the process, e.g. PyTorch's, is reused), so libmfk.so has no link-time dependency on it.
 *   mfk_comm_unique_id: rank 0 makes the 128-byte id; the caller hands it to every rank by its own means
 *                       (torch.distributed broadcast, MPI, a file).
 *   mfk_comm_init:      collective; one communicator per ctx, on the ctx's device and stream.  world = 1 is allowed.
 *   mfk_allgather_sites: collective.  rows: n_rows records of row_bytes bytes (any fixed-width table: mfk_site, mfk_flank,
 *                       mfk_pair), blob: their string bytes; both HOST buffers (the tables of this ABI are host tables).
 *                       Every rank sends a 16-byte header + rows + blob padded to pad_bytes in ONE
 *                       ncclAllGather; a second one with the true maximum follows only if some table did not fit.  pad_bytes <= 0 lets the
 *                       size follow the tables (16 KB at first, then the largest table of the previous gather + 25 %; the same on
 *                       every rank because every rank sees every header).
 *                       counts (2 x world, may be NULL): rows and blob bytes of every rank.
 *   mfk_allgather_fetch: copies rank r's rows / blob of the last gather (either pointer may be NULL).
 * Return code 101 = an RCCL call failed, 102 = librccl.so could not be loaded (message in mfk_last_error). */
#define MFK_COMM_ID_BYTES 128
int mfk_comm_unique_id(mfk_ctx *ctx, void *id128);
int mfk_comm_init(mfk_ctx *ctx, const void *id128, int rank, int world);
int mfk_comm_free(mfk_ctx *ctx);
int mfk_allgather_sites(mfk_ctx *ctx, const void *rows, int64_t n_rows, int32_t row_bytes, const void *blob, int64_t blob_bytes,
                        int64_t pad_bytes, int64_t *counts);
int mfk_allgather_fetch(mfk_ctx *ctx, int rank, void *rows, void *blob);
/* all ranks at once: rows of rank 0, 1, ... back to back, and their blobs back to back (sizes: the counts of the gather) */
int mfk_allgather_fetch_all(mfk_ctx *ctx, void *rows, void *blob);
/* bytes per rank the next mfk_allgather_sites with pad_bytes <= 0 will exchange.  The hint behind it belongs to the
 * communicator: mfk_comm_init / mfk_comm_free reset it to 16 KB, so ranks that join a new communicator with different
 * gather histories still agree on the size (they must: ncclAllGather with unequal counts is undefined). */
int64_t mfk_allgather_next_bytes(mfk_ctx *ctx);

/* Sharded path, host tables (no ctx, no device work; NO COUNTERPART IN THE REFERENCE): the selection and merge loops around
 * mfk_allgather_sites (mustache_amd/dist.py states the same in numpy; tests compare the two).  A shard owns, per contig, the
 * positions own_lo[c] <= pos < own_hi[c] (an empty interval where it owns nothing).
 *   mfk_flanks_owned      rows of a flank table the shard owns, their sequences compacted in row order -> row count
 *   mfk_flank_runs_owned  rows of the MERGED table the shard pairs: every run (same contig, gaps <= reach = maxdr + 1,
 *                         pairflanks.py:155-157) that holds an owned position -> indices
 *   mfk_pairs_owned       pairs whose 5p flank the shard owns, flank indices mapped back to the merged table (index_map),
 *                         the four strings of a pair compacted in the library's order
 *   mfk_flanks_merge / mfk_pairs_merge   tables of all ranks back to back (as mfk_allgather_fetch_all delivers them): string
 *                         offsets re-based in place; returns 1 if the rows are then in table order ((contig, pos); (contig name
 *                         rank, pos_5p, pos_3p), pairflanks.py:248), 0 if the caller still has to sort (whole-contig shards). */
int64_t mfk_flanks_owned(const mfk_flank *rows, int64_t n, const char *seqs, const int32_t *own_lo, const int32_t *own_hi, int32_t n_contigs,
                         mfk_flank *out, char *out_seqs, int64_t *out_seq_bytes);
int64_t mfk_flank_runs_owned(const mfk_flank *rows, int64_t n, const int32_t *own_lo, const int32_t *own_hi, int32_t n_contigs, int32_t reach, int64_t *sel);
int64_t mfk_pairs_owned(const mfk_pair *rows, int64_t n, const char *strs, const int32_t *own_lo, const int32_t *own_hi, int32_t n_contigs,
                        const int64_t *index_map, mfk_pair *out, char *out_strs, int64_t *out_str_bytes);
int mfk_flanks_merge(mfk_flank *rows, const int64_t *counts, int32_t world);
int mfk_pairs_merge(mfk_pair *rows, const int64_t *counts, int32_t world, const int32_t *contig_rank, int32_t n_contigs);

/* time spent in device work of the last compute call, milliseconds (HIP events on the ctx stream) */
double mfk_last_device_ms(mfk_ctx *ctx);
/* on (the default): findflanks / pairflanks / recall record an event before their first and behind their last kernel, which is
 * what mfk_last_device_ms reports.  off: those two records per stage are skipped (a few microseconds of host time each, which a
 * loop over sub-millisecond calls notices) and mfk_last_device_ms returns 0. */
int mfk_span_timing(mfk_ctx *ctx, int on);

/* per-kernel timing for bench.py: when enabled, every kernel launch of the compute calls is bracketed by HIP events on
 * the ctx stream and the durations are accumulated per kernel name.  enable(…, on) also clears the totals.
 * fetch: names = '\n'-separated kernel names in the order of ms[] / calls[]. */
int mfk_profile_enable(mfk_ctx *ctx, int on);
int mfk_profile_fetch(mfk_ctx *ctx, char *names, int64_t names_cap, double *ms, int64_t *calls, int64_t cap, int64_t *n);
/* restrict the bracketing to the kernels named in `names` ('\n'-separated; NULL or "" = every kernel), so that a timed
 * region pays for two event records per launch only where a duration is wanted.  Also clears the totals. */
int mfk_profile_select(mfk_ctx *ctx, const char *names);

/* ---- BGZF inflate on the device (SURVEY 8f N3 "optionally GPU inflate"; the reference inflates inside pysam/htslib,
 * findflanks.py:22, 100-121).  One wave per block; compressed bytes in device memory -> inflated bytes in device memory.
 * First stage of a decode that stays on the device; the shipped stage drivers still use the host reader (libmfkio.so). ---- */
typedef struct mfk_bgzf_block {
    uint64_t comp_off;   /* offset of the block's raw deflate stream in the compressed buffer          */
    uint64_t out_off;    /* where its bytes go in the output buffer                                    */
    uint32_t comp_len;   /* length of the deflate stream (BSIZE - XLEN - 19)                           */
    uint32_t out_len;    /* ISIZE                                                                      */
} mfk_bgzf_block;
/* blocks: HOST array.  d_comp / d_out: DEVICE.  h_status: NULL or n_blocks codes (0 = ok).  Returns 0, 12 when some block
 * does not inflate to exactly out_len bytes (*n_bad: how many; nothing is read or written outside a block's own ranges,
 * whatever the data), 1 for bad arguments.  The blocks' CRC32 is not checked here. */
int mfk_bgzf_inflate(mfk_ctx *ctx, const void *d_comp, int64_t comp_bytes, const mfk_bgzf_block *blocks, int64_t n_blocks,
                     void *d_out, int64_t out_bytes, int32_t *h_status, int64_t *n_bad);

/* ---- BAM decode on the device (mfk_decode.hip): BGZF file bytes in device memory -> the packed reads above in device
 * memory, what libmfkio.so does on host threads (include/mfk_bamio.h) and the reference gets from pysam while it iterates
 * the BAM (findflanks.py:100-121, sctools.py:5-31).  Two calls: sizes, then the caller allocates and the columns are written
 * into its buffers.  Results equal mfkio_pack_bam_filtered's arrays (tests/test_gpu_decode.py). ---- */
/* BGZF block table of a file in HOST memory; blocks == NULL only counts; file_off: NULL or the file offset of every block. */
int mfk_bgzf_block_table(mfk_ctx *ctx, const void *file, int64_t size, mfk_bgzf_block *blocks, uint64_t *file_off, uint32_t *crc,
                         int64_t cap, int64_t *n_blocks, int64_t *inflated_bytes);   /* crc: NULL or the CRC-32 of every block's trailer */
/* CRC-32 (gzip's) of the byte ranges out_off .. out_off + out_len of d_data (DEVICE), one per block of the HOST table. */
int mfk_bgzf_crc32(mfk_ctx *ctx, const void *d_data, int64_t data_bytes, const mfk_bgzf_block *blocks, int64_t n_blocks, uint32_t *h_crc);
/* Stage 1.  d_comp: the file in DEVICE memory; blocks: HOST; first_record_off: offset of the first alignment record in the
 * inflated stream (the caller reads the header); pool_mode 0 none / 1 clipped / 2 all; exclude_flags: FLAG bits to skip.
 * sizes[8] <- n_reads, n_cx, n_blk, n_pool, pool bytes, 0, n_records, inflated bytes.  Returns 0; 13 = a guessed record
 * boundary did not verify (use the host reader); 14 = a record the packer rejects (sizes[0] = its index among all records,
 * sizes[1] = reason); 15 = damaged stream; 12 = a block does not inflate, or (block_crc: HOST, NULL = not checked) its
 * inflated bytes do not have the CRC-32 of its trailer. */
int mfk_decode_bam(mfk_ctx *ctx, const void *d_comp, int64_t comp_bytes, const mfk_bgzf_block *blocks, const uint32_t *block_crc,
                   int64_t n_blocks, int64_t first_record_off, int32_t n_contigs, int32_t pool_mode, uint32_t exclude_flags,
                   int64_t *sizes);
/* Stage 2.  dst: DEVICE buffers sized from stage 1 (cx_off n_cx + 1, pool_off n_pool + 1 entries; the h_ fields are ignored).
 * h_contig_read_start: HOST, n_contigs + 1.  marks / mark_ref_pos: NULL, or HOST room for ceil(inflated / granule) entries
 * of the record index (stream offset of the first record starting in the granule, ~0 = none; its reference id and position). */
int mfk_decode_bam_fetch(mfk_ctx *ctx, const mfk_reads *dst, int64_t *h_contig_read_start, int32_t *max_span, int64_t granule,
                         uint64_t *marks, int32_t *mark_ref_pos);

#ifdef __cplusplus
}
#endif
#endif /* MFK_H */

/*
 * mfk_bamio.h -- C-ABI of libmfkio.so: native BAM ingestion for the flank-discovery path.
 *
 * Host-only code (no HIP): multi-threaded BGZF inflate, BAM record parsing and packing into the structure-of-arrays
 * that include/mfk.h's mfk_reads describes.  It replaces, for this path, what the reference gets from pysam/htslib
 * while it iterates the BAM and evaluates the per-read predicates of mustache/sctools.py:
 *     for read in bam                                  findflanks.py:100-121
 *     sctools.is_{left,right}_softclipped_lenient      sctools.py:17-31
 *     sctools.{left,right}_softclipped_site_lenient    sctools.py:99-109
 *     read.get_blocks() / cigartuples / get_tag('MD')  pysamtools.py:186-209, findflanks.py:455-509
 *     bam.fetch(contig, start, end) for pairflanks     pairflanks.py:393-409 (region mode below)
 * The Python packer mustache_amd/pack.py::pack_reads is the readable statement of the same rules; the two are
 * compared array for array in tests/test_bamio_native.py.
 *
 * Conventions: extern "C", plain pointers and sizes.  mfkio_pack_bam returns 0 on success; on failure it returns a
 * non-zero code and *out still holds an object whose mfkio_pack_error() gives the message (free it as usual).
 * The arrays of a pack stay valid until mfkio_pack_free.
 */
#ifndef MFK_BAMIO_H
#define MFK_BAMIO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* which reads carry their bases + qualities in the pool */
enum {
    MFKIO_POOL_NONE = 0,
    MFKIO_POOL_CLIPPED = 1,  /* lenient-clipped reads (findflanks) */
    MFKIO_POOL_ALL = 2,
    MFKIO_POOL_REGIONS = 3   /* pairflanks: only reads overlapping [lo,hi) of a region are packed at all; those
                                overlapping its [es,ee) window are pooled */
};

/* error codes */
enum {
    MFKIO_OK = 0,
    MFKIO_ERR_IO = 1,        /* cannot open / map the file                                  */
    MFKIO_ERR_FORMAT = 2,    /* not BGZF / not BAM / truncated / checksum mismatch          */
    MFKIO_ERR_PACK = 3,      /* a record the packer rejects (same cases as pack.PackError)  */
    MFKIO_ERR_ARG = 4,
    MFKIO_ERR_INDEX = 5      /* mfkio_pack_bam_regions_indexed: no usable record index (missing, stale, another FLAG
                                filter): run mfkio_pack_bam_filtered instead                 */
};

/* One candidate region of pairflanks (mustache_amd/pairflanks.py::candidate_regions): reads with
 * ref_start < hi && ref_end > lo on contig_id are kept, those with ref_start < ee && ref_end > es are pooled. */
typedef struct mfkio_region {
    int32_t contig_id;
    int32_t lo, hi;
    int32_t es, ee;
} mfkio_region;

/* The packed arrays (HOST pointers): same columns, element types and meaning as mfk_reads in mfk.h. */
typedef struct mfkio_arrays {
    int64_t n_reads;
    int32_t n_contigs;
    int32_t max_span;
    int64_t n_cx, n_blk, n_pool;
    int64_t pool_bases_bytes, pool_quals_bytes;
    const int32_t *ref_start, *ref_end, *link;
    const uint16_t *contig_id, *qlen, *first_op_len, *last_op_len;
    const uint8_t *mapq, *flags;
    const int32_t *contig_len;          /* n_contigs   */
    const int64_t *contig_read_start;   /* n_contigs+1 */
    const uint32_t *cx_read, *cx_off;   /* n_cx, n_cx+1 */
    const int32_t *blk_start, *blk_end, *blk_ins;
    const uint32_t *pool_read;          /* n_pool   */
    const uint64_t *pool_off;           /* n_pool+1 */
    const uint8_t *pool_bases, *pool_quals;
    const char *contig_names;           /* n_contigs NUL-terminated names back to back */
    int64_t contig_names_bytes;
    /* what the run did (for logs and bench.py) */
    int64_t n_records;                  /* alignment records in the file, unmapped ones included */
    int64_t compressed_bytes, uncompressed_bytes;
    int32_t n_threads;
    int32_t inflate_backend;            /* 0 = zlib, 1 = libdeflate (found at run time) */
    double seconds_inflate, seconds_parse, seconds_link, seconds_total;
    int64_t n_redone_passes;            /* passes walked a second time because a guessed share start was wrong */
} mfkio_arrays;

typedef struct mfkio_pack mfkio_pack;

/* Decode `path` (a coordinate-sorted BAM) and pack its mapped reads.
 * regions / n_regions: only for MFKIO_POOL_REGIONS.  n_threads <= 0: min(hardware threads, 16), or $MFK_THREADS.
 * chunk_bytes <= 0: 256 MiB of uncompressed data per pass (tests pass small values to exercise records that straddle
 * passes). */
int mfkio_pack_bam(const char *path, int pool_mode, const mfkio_region *regions, int64_t n_regions, int n_threads,
                   int64_t chunk_bytes, mfkio_pack **out);
/* The same with a FLAG filter: records that have any bit of exclude_flags set are skipped, as `samtools view -F` does.
 * 0x900 (secondary | supplementary) is what the reference's formatbam removes before findflanks ever sees the file
 * (mustache/samtools.py:9-18, formatbam.py:63-66): with it a BAM straight from the aligner does not trip the
 * "more than two reads of one query name" abort of findflanks.py:971-978. */
int mfkio_pack_bam_filtered(const char *path, int pool_mode, const mfkio_region *regions, int64_t n_regions, int n_threads,
                            int64_t chunk_bytes, uint32_t exclude_flags, mfkio_pack **out);

/* Record index: what lets pairflanks read only the reads over its candidate regions, as the reference does through the BAI
 * and bam.fetch (pairflanks.py:393-409) -- findflanks has to inflate every block anyway (findflanks.py:100-121) and leaves,
 * per 32 KiB of the uncompressed stream, the offset of a record and its (reference id, position), plus the BGZF block table.
 *   mfkio_pack_bam_index:           mfkio_pack_bam_filtered (pool_mode NONE / CLIPPED / ALL) that also writes the index
 *                                   to `index_out` (best effort: a path that cannot be written is not an error).
 *   mfkio_pack_bam_regions_indexed: mfkio_pack_bam_filtered(MFKIO_POOL_REGIONS) through the index: only the BGZF blocks
 *                                   between the marks around each region are inflated.  Same arrays, byte for byte.
 *                                   MFKIO_ERR_INDEX when the index is missing or was made for another file (size, mtime)
 *                                   or FLAG filter: the caller falls back to the full pass. */
int mfkio_pack_bam_index(const char *path, int pool_mode, int n_threads, int64_t chunk_bytes, uint32_t exclude_flags, const char *index_out,
                         mfkio_pack **out);
int mfkio_pack_bam_regions_indexed(const char *path, const char *index_path, const mfkio_region *regions, int64_t n_regions, int n_threads,
                                   uint32_t exclude_flags, mfkio_pack **out);
const mfkio_arrays *mfkio_pack_arrays(const mfkio_pack *p);
const char *mfkio_pack_error(const mfkio_pack *p);
void mfkio_pack_free(mfkio_pack *p);
int mfkio_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MFK_BAMIO_H */

// mfk_tables.hip -- host-side table work of the sharded path: which rows of a flank / pair table a
// shard owns, with their strings compacted, and the merge of the tables the ranks exchanged.  No device work and no ctx:
// these are the loops mustache_amd/dist.py states in numpy (ShardPlan.owns, _compact, _flank_rows_for, merge_flanks,
// merge_pairs), which cost 2.8 ms per step at 20 000 flank rows / 10 000 pairs and bounded the strong-scaling curve.
// The reference has no counterpart (one process per sample, no exchange).
#include <stdint.h>
#include <string.h>

#include "../../include/mfk.h"

namespace {
inline bool owned(const int32_t *lo, const int32_t *hi, int32_t nc, int32_t c, int32_t pos) { return c >= 0 && c < nc && pos >= lo[c] && pos < hi[c]; }
}

extern "C" {

// rows with own_lo[contig] <= pos < own_hi[contig] (a contig the shard owns nothing of has an empty interval), strings in row
// order at the front of out_seqs.  out / out_seqs: room for n rows / all of the input's string bytes.  Returns the row count.
int64_t mfk_flanks_owned(const mfk_flank *rows, int64_t n, const char *seqs, const int32_t *own_lo, const int32_t *own_hi, int32_t n_contigs,
                         mfk_flank *out, char *out_seqs, int64_t *out_seq_bytes) {
    int64_t k = 0, b = 0;
    for (int64_t i = 0; i < n; i++) {
        const mfk_flank &f = rows[i];
        if (!owned(own_lo, own_hi, n_contigs, f.contig_id, f.pos)) continue;
        out[k] = f;
        const int64_t len = f.seq_len > 0 ? f.seq_len : 0;
        if (len) memcpy(out_seqs + b, seqs + f.seq_off, (size_t)len);
        out[k].seq_off = b;
        b += len; k++;
    }
    if (out_seq_bytes) *out_seq_bytes = b;
    return k;
}

// rows of the merged flank table a shard hands to mfk_pairflanks: every run of rows (same contig, gaps <= reach) that holds an
// owned position (candidate pairs, and through shared flanks the greedy assignment, never cross a wider gap,
// pairflanks.py:155-157, 229-245).  sel: room for n indices.  Returns how many.
int64_t mfk_flank_runs_owned(const mfk_flank *rows, int64_t n, const int32_t *own_lo, const int32_t *own_hi, int32_t n_contigs, int32_t reach, int64_t *sel) {
    int64_t k = 0, i = 0;
    while (i < n) {
        int64_t j = i;
        bool any = owned(own_lo, own_hi, n_contigs, rows[i].contig_id, rows[i].pos);
        while (j + 1 < n && rows[j + 1].contig_id == rows[j].contig_id && (int64_t)rows[j + 1].pos - (int64_t)rows[j].pos <= reach) {
            j++;
            any = any || owned(own_lo, own_hi, n_contigs, rows[j].contig_id, rows[j].pos);
        }
        if (any) for (int64_t q = i; q <= j; q++) sel[k++] = q;
        i = j + 1;
    }
    return k;
}

// pairs whose 5p flank the shard owns; index_5p / index_3p mapped through `index_map` (rows of the table the shard paired ->
// rows of the merged table; NULL = as they are); the four strings of a pair in the library's order at the front of out_strs
int64_t mfk_pairs_owned(const mfk_pair *rows, int64_t n, const char *strs, const int32_t *own_lo, const int32_t *own_hi, int32_t n_contigs,
                        const int64_t *index_map, mfk_pair *out, char *out_strs, int64_t *out_str_bytes) {
    int64_t k = 0, b = 0;
    for (int64_t i = 0; i < n; i++) {
        const mfk_pair &p = rows[i];
        if (!owned(own_lo, own_hi, n_contigs, p.contig_id, p.pos_5p)) continue;
        mfk_pair o = p;
        if (index_map) { o.index_5p = (int32_t)index_map[p.index_5p]; o.index_3p = (int32_t)index_map[p.index_3p]; }
        auto put = [&](int64_t off, int32_t len, int64_t *new_off) {
            const int64_t l = len > 0 ? len : 0;          // a negative length is None: no bytes
            if (l) memcpy(out_strs + b, strs + off, (size_t)l);
            *new_off = b;
            b += l;
        };
        put(p.ir3_off, p.ir3_len, &o.ir3_off); put(p.ir5_off, p.ir5_len, &o.ir5_off);
        put(p.dr_ref_off, p.dr_ref_len, &o.dr_ref_off); put(p.dr_reads_off, p.dr_len, &o.dr_reads_off);
        out[k++] = o;
    }
    if (out_str_bytes) *out_str_bytes = b;
    return k;
}

// The tables of all ranks back to back (rank order, each with its strings compact and in row order, as mfk_flanks_owned /
// mfk_pairs_owned make them and mfk_allgather_fetch_all delivers them): string offsets re-based onto the concatenated blob,
// in place.  Returns 1 when the rows are then in table order -- (contig, pos) for flanks -- which is the case for coordinate
// shards; 0 when the caller still has to sort (whole-contig shards).
int mfk_flanks_merge(mfk_flank *rows, const int64_t *counts /* rows, bytes per rank */, int32_t world) {
    int64_t r0 = 0, b0 = 0;
    for (int32_t r = 0; r < world; r++) {
        for (int64_t i = r0; i < r0 + counts[2 * r]; i++) rows[i].seq_off += b0;
        r0 += counts[2 * r]; b0 += counts[2 * r + 1];
    }
    for (int64_t i = 1; i < r0; i++)
        if (rows[i].contig_id < rows[i - 1].contig_id || (rows[i].contig_id == rows[i - 1].contig_id && rows[i].pos < rows[i - 1].pos)) return 0;
    return 1;
}

// the same for pair tables; table order = (rank of the contig's name, pos_5p, pos_3p) (pairflanks.py:248)
int mfk_pairs_merge(mfk_pair *rows, const int64_t *counts, int32_t world, const int32_t *contig_rank, int32_t n_contigs) {
    int64_t r0 = 0, b0 = 0;
    for (int32_t r = 0; r < world; r++) {
        for (int64_t i = r0; i < r0 + counts[2 * r]; i++) { rows[i].ir3_off += b0; rows[i].ir5_off += b0; rows[i].dr_ref_off += b0; rows[i].dr_reads_off += b0; }
        r0 += counts[2 * r]; b0 += counts[2 * r + 1];
    }
    for (int64_t i = 1; i < r0; i++) {
        const mfk_pair &a = rows[i - 1], &b = rows[i];
        if (a.contig_id < 0 || a.contig_id >= n_contigs || b.contig_id < 0 || b.contig_id >= n_contigs) return 0;
        const int32_t ra = contig_rank[a.contig_id], rb = contig_rank[b.contig_id];
        if (rb < ra || (rb == ra && (b.pos_5p < a.pos_5p || (b.pos_5p == a.pos_5p && b.pos_3p < a.pos_3p)))) return 0;
    }
    return 1;
}

}  // extern "C"

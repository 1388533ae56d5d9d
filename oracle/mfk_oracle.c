/*
 * mfk_oracle.c -- TEST INFRASTRUCTURE ONLY.  Sequential CPU restatement of the reference's flank-discovery path
 * (durrantmm/mustache: findflanks.py, flanktrie.py, pairflanks.py, sctools.py, pysamtools.py, misc.py) on the packed
 * arrays of include/mfk.h.  It is the checker for the HIP library and the `cpu_baseline` ("port") of bench.py.
 * Nothing in the product (mustache_amd/) may call it.
 *
 * Parity status: PINNED for findflanks and for pairflanks except the inverted-repeat columns -- the golden vectors in
 * tests/golden/ were produced by running the reference itself (oracle/ref_harness.py) and this file reproduces them.
 * PARITY UNPINNED for has_IR / IR_length / IR_5p / IR_3p: the reference shells out to EMBOSS 6.6.0 `einverted`
 * (embosstools.py:8-14), which is neither in the reference tree nor in this image; ir_detect() below restates
 * einverted's published model (local alignment of a sequence against its reverse complement, match +3, mismatch -4,
 * gap 12, threshold 15) with tie and traceback rules of our own (DESIGN.md "IR definition").
 *
 * It deliberately uses different algorithms from the HIP kernels (explicit per-site event lists, dense name ids
 * with sort+unique, a pointer trie with insertion-ordered children, a full Levenshtein matrix, union-find) so that
 * agreement between the two means something.
 *
 * All pointers in mfk_reads are HOST pointers here.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mfk.h"

typedef struct { int32_t *v; int64_t n, cap; } ivec;
static void iv_push(ivec *a, int32_t x) {
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 8; a->v = (int32_t *)realloc(a->v, a->cap * sizeof(int32_t)); }
    a->v[a->n++] = x;
}
static void iv_free(ivec *a) { free(a->v); a->v = NULL; a->n = a->cap = 0; }

static const char BASES[] = "=ACMGRSVTWYHKDBN";

/* ---------------------------------------------------------------------------------------------------------- */
/* per-read predicates: sctools.py                                                                             */
static inline int lkind(const mfk_reads *R, int64_t r) { return MFK_LKIND(R->flags[r]); }
static inline int rkind(const mfk_reads *R, int64_t r) { return MFK_RKIND(R->flags[r]); }
/* len(left_softclipped_sequence(read)) sctools.py:135-142 */
static inline int llen(const mfk_reads *R, int64_t r) { int k = lkind(R, r); return k == 1 ? R->first_op_len[r] : (k == 2 ? 1 : 0); }
static inline int rlen(const mfk_reads *R, int64_t r) { int k = rkind(R, r); return k == 1 ? R->last_op_len[r] : (k == 2 ? 1 : 0); }
/* left_softclipped_site_lenient / right_softclipped_site_lenient sctools.py:99-109 */
static inline int32_t lsite(const mfk_reads *R, int64_t r) { return lkind(R, r) == 1 ? R->ref_start[r] - 1 : R->ref_start[r]; }
static inline int32_t rsite(const mfk_reads *R, int64_t r) { return rkind(R, r) == 1 ? R->ref_end[r] : R->ref_end[r] - 1; }

/* passes_read_filters findflanks.py:586-592 + read_meets_min_alignment_inner_length sctools.py:166-180 */
static int passes_read_filters(const mfk_reads *R, const mfk_params *P, int64_t r) {
    if ((int)R->mapq[r] < P->min_alignment_quality) return 0;
    if (lkind(R, r) && rkind(R, r)) {
        int aligned = (int)R->qlen[r] - rlen(R, r) - llen(R, r);
        if (aligned < P->min_alignment_inner_length) return 0;
    }
    return 1;
}
/* is_softclipped_lenient_at_site sctools.py:33-55 (contig equality is implied by the caller's fetch) */
static int clipped_at_site(const mfk_reads *R, int64_t r, int32_t pos) {
    if (rkind(R, r) && rsite(R, r) == pos) return 1;
    if (lkind(R, r) && lsite(R, r) == pos) return 1;
    return 0;
}

/* aligned blocks of a read (pysam get_blocks) */
typedef struct { int n; const int32_t *s, *e, *ins; int32_t s1, e1, i1; } blocks_t;
static blocks_t get_blocks(const mfk_reads *R, int64_t r) {
    blocks_t b;
    if (!(R->flags[r] & MFK_FLAG_COMPLEX)) {
        /* single block: kept inline (s == NULL), read through blk_s/blk_e/blk_i below */
        b.n = 1; b.s1 = R->ref_start[r]; b.e1 = R->ref_end[r]; b.i1 = 0; b.s = b.e = b.ins = NULL;
        return b;
    }
    int64_t lo = 0, hi = R->n_cx;
    while (lo < hi) { int64_t m = (lo + hi) / 2; if ((int64_t)R->cx_read[m] < r) lo = m + 1; else hi = m; }
    uint32_t o0 = R->cx_off[lo], o1 = R->cx_off[lo + 1];
    b.n = (int)(o1 - o0); b.s = R->blk_start + o0; b.e = R->blk_end + o0; b.ins = R->blk_ins + o0;
    b.s1 = b.e1 = b.i1 = 0;
    return b;
}
static inline int32_t blk_s(const blocks_t *b, int i) { return b->s ? b->s[i] : b->s1; }
static inline int32_t blk_e(const blocks_t *b, int i) { return b->s ? b->e[i] : b->e1; }
static inline int32_t blk_i(const blocks_t *b, int i) { return b->s ? b->ins[i] : b->i1; }

enum { CAT_NONE = 0, CAT_RUNTHROUGH, CAT_SMALL5, CAT_SMALL3, CAT_LARGE5, CAT_LARGE3, CAT_DELETION };

/* process_aligned_blocks_at_site findflanks.py:455-509 */
static int classify_read(const mfk_reads *R, const mfk_params *P, int64_t r, int32_t position, int deletions_only) {
    blocks_t b = get_blocks(R, r);
    int runthrough = 0, s5 = 0, s3 = 0, l5 = 0, l3 = 0, del = 0;
    for (int i = 0; i < b.n; i++)
        if (blk_s(&b, i) <= position && blk_e(&b, i) > position) runthrough = 1;
    for (int i = 0; i + 1 < b.n; i++) {
        int32_t e1 = blk_e(&b, i), s2 = blk_s(&b, i + 1);
        if (e1 == s2) {
            if (deletions_only) continue;
            int32_t ins = blk_i(&b, i + 1); /* get_insertion_length pysamtools.py:186-209 */
            if (position == e1 - 1) { if (ins >= P->large_insertion_cutoff) l3 = 1; else s3 = 1; }
            else if (position == s2) { if (ins >= P->large_insertion_cutoff) l5 = 1; else s5 = 1; }
        } else if (e1 <= position && s2 > position) del = 1;
    }
    if (s5) return CAT_SMALL5;
    if (s3) return CAT_SMALL3;
    if (l5) return CAT_LARGE5;
    if (l3) return CAT_LARGE3;
    if (del) return CAT_DELETION;
    if (runthrough) return CAT_RUNTHROUGH;
    return CAT_NONE;
}

/* ---------------------------------------------------------------------------------------------------------- */
/* misc.takeClosestSmaller / takeClosestLarger, ported branch by branch (misc.py:29-76); -1 stands for None    */
static int64_t bisect_left64(const int32_t *a, int64_t n, int32_t x) { int64_t lo = 0, hi = n; while (lo < hi) { int64_t m = (lo + hi) / 2; if (a[m] < x) lo = m + 1; else hi = m; } return lo; }
static int64_t bisect_right64(const int32_t *a, int64_t n, int32_t x) { int64_t lo = 0, hi = n; while (lo < hi) { int64_t m = (lo + hi) / 2; if (x < a[m]) hi = m; else lo = m + 1; } return lo; }
int mfko_take_closest_smaller(const int32_t *a, int64_t n, int32_t number, int32_t *out) {
    if (n == 0) return 0;
    int64_t pos = bisect_left64(a, n, number);
    if (pos == 0) { if (a[0] >= number) return 0; *out = a[0]; return 1; }
    if (pos == n) { if (a[n - 1] < number) { *out = a[n - 1]; return 1; } *out = a[n - 2]; return 1; }
    if (a[pos] < number) { *out = a[pos]; return 1; }
    *out = a[pos - 1]; return 1;
}
int mfko_take_closest_larger(const int32_t *a, int64_t n, int32_t number, int32_t *out) {
    if (n == 0) return 0;
    int64_t pos = bisect_right64(a, n, number);
    if (pos == n) { if (a[n - 1] <= number) return 0; *out = a[n - 1]; return 1; }
    if (pos == 0) { if (a[0] > number) { *out = a[0]; return 1; } *out = a[1]; return 1; }
    if (a[pos] > number) { *out = a[pos]; return 1; }
    *out = a[pos + 1]; return 1;
}

/* ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t contig, pos;
    ivec ev5, ev3;              /* read indices clipped here, BAM order (softclip_5p_reads / softclip_3p_reads) */
    int meets5, meets3, keep5, keep3, alive;
    int32_t c[10];              /* sc5 sc3 runthrough si5 si3 li5 li3 del updel downdel */
    /* consensus results: list of (count, seq) per side */
    int ncons5, ncons3;
    int32_t *cons_count5, *cons_count3;
    char **cons_seq5, **cons_seq3;
} site_t;

typedef struct {
    site_t *sites; int64_t n_sites;
    int32_t *name_id;           /* dense query-name ids rebuilt from link chains */
    int64_t stats[MFK_N_STATS];
    mfk_flank *flanks; int64_t n_flanks; char *seqs; int64_t seq_bytes;
    mfk_pair *pairs; int64_t n_pairs; char *strs; int64_t str_bytes;
    int64_t pstats[MFK_N_STATS];
    int32_t *isq_f, *isq_r; int64_t n_isq;   /* result of the last mfko_inferseq_pairs */
    int32_t *ir_ov; int64_t n_ir_ov;   /* external inverted-repeat table for the next pairflanks call: i5 i3 has a1s a1e a2s a2e */
    char err[512];
} octx;

static int cmp_i32(const void *a, const void *b) { int32_t x = *(const int32_t *)a, y = *(const int32_t *)b; return (x > y) - (x < y); }
static int32_t distinct_names(const octx *C, const ivec *reads) {
    if (!reads->n) return 0;
    int32_t *t = (int32_t *)malloc(reads->n * sizeof(int32_t));
    for (int64_t i = 0; i < reads->n; i++) t[i] = C->name_id[reads->v[i]];
    qsort(t, reads->n, sizeof(int32_t), cmp_i32);
    int32_t d = 1;
    for (int64_t i = 1; i < reads->n; i++) if (t[i] != t[i - 1]) d++;
    free(t);
    return d;
}

static int32_t total_count(const site_t *s) { return s->c[0] + s->c[1] + s->c[2] + s->c[3] + s->c[4] + s->c[5] + s->c[6] + s->c[7]; }

typedef struct { int32_t contig, pos, side; int32_t read; } event_t;
static int cmp_event(const void *a, const void *b) {
    const event_t *x = (const event_t *)a, *y = (const event_t *)b;
    if (x->contig != y->contig) return x->contig < y->contig ? -1 : 1;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    if (x->read != y->read) return x->read < y->read ? -1 : 1;
    return x->side - y->side;
}

static int64_t count_alive(const octx *C) { int64_t n = 0; for (int64_t i = 0; i < C->n_sites; i++) n += C->sites[i].alive; return n; }

/* filter_softclips_mindistance findflanks.py:171-200 with has_nearby_{3p,5p}_mate 542-558 */
static void filter_mindistance(octx *C, const mfk_params *P) {
    int64_t n = C->n_sites;
    int32_t *p5 = (int32_t *)malloc((n + 1) * sizeof(int32_t)), *p3 = (int32_t *)malloc((n + 1) * sizeof(int32_t));
    int64_t i = 0;
    while (i < n) {
        int64_t j = i; int32_t contig = C->sites[i].contig; int64_t n5 = 0, n3 = 0;
        while (j < n && C->sites[j].contig == contig) {
            site_t *s = &C->sites[j];
            if (s->alive) { if (s->c[0] > 0) p5[n5++] = s->pos; if (s->c[1] > 0) p3[n3++] = s->pos; }
            j++;
        }
        for (int64_t k = i; k < j; k++) {
            site_t *s = &C->sites[k];
            if (!s->alive) continue;
            int32_t v;
            if (s->c[0] > 0) { int ok = mfko_take_closest_smaller(p3, n3, s->pos, &v) && (s->pos - v <= P->min_distance_to_mate); if (!ok) s->keep5 = 0; }
            if (s->c[1] > 0) { int ok = mfko_take_closest_larger(p5, n5, s->pos, &v) && (v - s->pos <= P->min_distance_to_mate); if (!ok) s->keep3 = 0; }
        }
        for (int64_t k = i; k < j; k++) { site_t *s = &C->sites[k]; if (s->alive && !(s->keep5 || s->keep3)) s->alive = 0; }
        i = j;
    }
    free(p5); free(p3);
}

/* get_reads_at_site findflanks.py:399-415: reads overlapping [start,end) that pass the filters, BAM order */
static void fetch_reads(const mfk_reads *R, int32_t contig, int32_t start, int32_t end, ivec *out) {
    int64_t r0 = R->h_contig_read_start[contig], r1 = R->h_contig_read_start[contig + 1];
    /* first read that can overlap: ref_start >= start - max_span */
    int64_t lo = r0, hi = r1; int32_t lim = start - R->max_span;
    while (lo < hi) { int64_t m = (lo + hi) / 2; if (R->ref_start[m] < lim) lo = m + 1; else hi = m; }
    for (int64_t r = lo; r < r1 && R->ref_start[r] < end; r++) {
        int32_t re = R->ref_end[r] > R->ref_start[r] ? R->ref_end[r] : R->ref_start[r] + 1;
        if (re > start) iv_push(out, (int32_t)r);
    }
}

/* one site: the reads of get_reads_at_site, sorted into the sets of get_unclipped_read_info_at_site (findflanks.py:418-452)
 * at pos, the deletion sets at pos-1 / pos+1 (380-396) and, as recall.py:96-112 does, the reads lenient-clipped at pos.
 * c10: softclip_5p softclip_3p runthrough si5 si3 li5 li3 del updel downdel (distinct query names) */
static void tally_site(const octx *C, const mfk_reads *R, const mfk_params *P, int32_t contig, int32_t pos, int32_t *c10) {
    int32_t clen = R->h_contig_len[contig];
    int32_t start = pos - 1, end = pos + 2;
    if (start < 0) start = 0;
    if (end > clen) end = clen;
    ivec fetched = {0}, reads = {0};
    if (start < end) fetch_reads(R, contig, start, end, &fetched);
    for (int64_t k = 0; k < fetched.n; k++) if (passes_read_filters(R, P, fetched.v[k])) iv_push(&reads, fetched.v[k]);
    ivec sets[10]; memset(sets, 0, sizeof sets);  /* sc5 sc3 runthrough si5 si3 li5 li3 del up down */
    for (int64_t k = 0; k < reads.n; k++) {
        int64_t r = reads.v[k];
        if (rkind(R, r) && rsite(R, r) == pos) iv_push(&sets[0], (int32_t)r);   /* is_right_softclipped_lenient_at_site -> 5p */
        if (lkind(R, r) && lsite(R, r) == pos) iv_push(&sets[1], (int32_t)r);   /* is_left_softclipped_lenient_at_site -> 3p */
        if (clipped_at_site(R, r, pos)) continue;
        int cat = classify_read(R, P, r, pos, 0);
        if (cat != CAT_NONE) iv_push(&sets[2 + cat - 1], (int32_t)r);
    }
    if (pos - 1 >= 0) for (int64_t k = 0; k < reads.n; k++) {
        int64_t r = reads.v[k];
        if (clipped_at_site(R, r, pos - 1)) continue;
        if (classify_read(R, P, r, pos - 1, 1) == CAT_DELETION) iv_push(&sets[8], (int32_t)r);
    }
    if (pos + 1 < clen) for (int64_t k = 0; k < reads.n; k++) {
        int64_t r = reads.v[k];
        if (clipped_at_site(R, r, pos + 1)) continue;
        if (classify_read(R, P, r, pos + 1, 1) == CAT_DELETION) iv_push(&sets[9], (int32_t)r);
    }
    for (int k = 0; k < 10; k++) { c10[k] = distinct_names(C, &sets[k]); iv_free(&sets[k]); }
    iv_free(&fetched); iv_free(&reads);
}

/* ---- trie (flanktrie.py) ------------------------------------------------------------------------------------ */
typedef struct { int32_t child[16]; int32_t first_child, last_child, next_sib, parent; int32_t ch; int32_t counter, word_count; int64_t qual; } tnode;
typedef struct { tnode *n; int32_t cnt, cap; } trie_t;
static int32_t trie_new(trie_t *T, int ch, int32_t parent) {
    if (T->cnt == T->cap) { T->cap = T->cap ? T->cap * 2 : 256; T->n = (tnode *)realloc(T->n, T->cap * sizeof(tnode)); }
    tnode *x = &T->n[T->cnt];
    for (int i = 0; i < 16; i++) x->child[i] = -1;
    x->first_child = x->last_child = x->next_sib = -1; x->parent = parent; x->ch = ch; x->counter = 1; x->word_count = 0; x->qual = 0;
    return T->cnt++;
}
/* Trie.add flanktrie.py:31-59 */
static void trie_add(trie_t *T, const uint8_t *w, const uint8_t *q, int len) {
    int32_t node = 0;
    for (int i = 0; i < len; i++) {
        int ch = w[i];
        int32_t c = T->n[node].child[ch];
        if (c >= 0) { T->n[c].counter += 1; T->n[c].qual += q[i]; node = c; }
        else {
            c = trie_new(T, ch, node);
            T->n[c].qual += q[i];
            T->n[node].child[ch] = c;
            if (T->n[node].last_child < 0) T->n[node].first_child = c; else T->n[T->n[node].last_child].next_sib = c;
            T->n[node].last_child = c;
            node = c;
        }
    }
    T->n[node].word_count += 1;
}
/* traverse_seqs flanktrie.py:157-170: leaves in DFS order, children in insertion order */
static void trie_leaves(const trie_t *T, int32_t node, ivec *out) {
    if (T->n[node].first_child < 0) { iv_push(out, node); return; }
    for (int32_t c = T->n[node].first_child; c >= 0; c = T->n[c].next_sib) trie_leaves(T, c, out);
}
static int node_depth(const trie_t *T, int32_t node) { int d = 0; while (T->n[node].parent >= 0) { d++; node = T->n[node].parent; } return d; }
static void node_word(const trie_t *T, int32_t node, uint8_t *out, int len) { for (int i = len - 1; i >= 0; i--) { out[i] = (uint8_t)T->n[node].ch; node = T->n[node].parent; } }

/* jellyfish.levenshtein_distance: unit-cost edit distance */
static int levenshtein(const uint8_t *a, int n, const uint8_t *b, int m) {
    int *prev = (int *)malloc((m + 1) * sizeof(int)), *cur = (int *)malloc((m + 1) * sizeof(int));
    for (int j = 0; j <= m; j++) prev[j] = j;
    for (int i = 1; i <= n; i++) {
        cur[0] = i;
        for (int j = 1; j <= m; j++) {
            int c = prev[j - 1] + (a[i - 1] != b[j - 1]);
            if (prev[j] + 1 < c) c = prev[j] + 1;
            if (cur[j - 1] + 1 < c) c = cur[j - 1] + 1;
            cur[j] = c;
        }
        int *t = prev; prev = cur; cur = t;
    }
    int d = prev[m];
    free(prev); free(cur);
    return d;
}
int mfko_levenshtein(const uint8_t *a, int n, const uint8_t *b, int m) { return levenshtein(a, n, b, m); }

static int uf_find(int *p, int x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }

static uint8_t pool_base(const mfk_reads *R, uint64_t k) { return (R->pool_bases[k >> 1] >> ((~k & 1) * 4)) & 15; }
static int64_t pool_index(const mfk_reads *R, int64_t r) {
    int64_t lo = 0, hi = R->n_pool;
    while (lo < hi) { int64_t m = (lo + hi) / 2; if ((int64_t)R->pool_read[m] < r) lo = m + 1; else hi = m; }
    if (lo >= R->n_pool || (int64_t)R->pool_read[lo] != r) return -1;
    return lo;
}

/* clipped sequence / qualities of one read for one side, in trie orientation (3p reversed, findflanks.py:936) */
static int clip_word(const mfk_reads *R, int64_t r, int side5, uint8_t *w, uint8_t *q) {
    int64_t pi = pool_index(R, r);
    if (pi < 0) return -1;
    uint64_t o = R->pool_off[pi]; int ql = R->qlen[r];
    if (side5) { int n = rlen(R, r); for (int i = 0; i < n; i++) { w[i] = pool_base(R, o + ql - n + i); q[i] = R->pool_quals[o + ql - n + i]; } return n; }
    int n = llen(R, r);
    for (int i = 0; i < n; i++) { w[i] = pool_base(R, o + n - 1 - i); q[i] = R->pool_quals[o + n - 1 - i]; }
    return n;
}

/* SoftclipConsensus findflanks.py:862-1104 for one site-side.  Returns 0 ok, 1 = ">2 reads per name" abort. */
static int make_consensus(octx *C, const mfk_reads *R, const mfk_params *P, const ivec *ev, int side5,
                          int *ncons, int32_t **ccount, char ***cseq) {
    int64_t n = ev->n; int maxlen = 0;
    for (int64_t i = 0; i < n; i++) { int l = side5 ? rlen(R, ev->v[i]) : llen(R, ev->v[i]); if (l > maxlen) maxlen = l; }
    /* one word per name, first-seen order (dict insertion order of softclip_*_reads) */
    int64_t *first = (int64_t *)malloc(n * sizeof(int64_t)), *second = (int64_t *)malloc(n * sizeof(int64_t)); int64_t nw = 0;
    for (int64_t i = 0; i < n; i++) {
        int32_t nm = C->name_id[ev->v[i]]; int64_t k;
        for (k = 0; k < nw; k++) if (C->name_id[first[k]] == nm) break;
        if (k == nw) { first[nw] = ev->v[i]; second[nw] = -1; nw++; }
        else if (second[k] < 0) second[k] = ev->v[i];
        else { free(first); free(second); return 1; }   /* findflanks.py:976-978 / 1025-1027 */
    }
    trie_t T = {0}; trie_new(&T, 0, -1);
    uint8_t *w1 = (uint8_t *)malloc(4 * (maxlen + 1)), *q1 = w1 + maxlen + 1, *w2 = q1 + maxlen + 1, *q2 = w2 + maxlen + 1;
    uint8_t *wm = (uint8_t *)malloc(2 * (maxlen + 1)), *qm = wm + maxlen + 1;
    for (int64_t k = 0; k < nw; k++) {
        int l1 = clip_word(R, first[k], side5, w1, q1);
        if (l1 < 0) { snprintf(C->err, sizeof C->err, "clipped read %lld has no sequence in the pool", (long long)first[k]); free(first); free(second); free(w1); free(wm); free(T.n); return 2; }
        if (second[k] < 0) { trie_add(&T, w1, q1, l1); continue; }
        /* combine_softclip_seq_quals findflanks.py:932-1027; read1 = first in BAM order (quirk Q4 build rule) */
        int l2 = clip_word(R, second[k], side5, w2, q2);
        if (l2 < 0) { snprintf(C->err, sizeof C->err, "clipped read %lld has no sequence in the pool", (long long)second[k]); free(first); free(second); free(w1); free(wm); free(T.n); return 2; }
        int L = l1 > l2 ? l1 : l2;
        for (int p = 0; p < L; p++) {
            if (p < l1 && p < l2) { if (q1[p] >= q2[p]) { qm[p] = q1[p]; wm[p] = w1[p]; } else { qm[p] = q2[p]; wm[p] = w2[p]; } }
            else if (p < l1) { qm[p] = q1[p]; wm[p] = w1[p]; }
            else { qm[p] = q2[p]; wm[p] = w2[p]; }
        }
        trie_add(&T, wm, qm, L);
    }
    free(w1); free(wm); free(first); free(second);
    /* traverse_all */
    ivec leaves = {0}; trie_leaves(&T, 0, &leaves);
    int nl = (int)leaves.n;
    int *ldepth = (int *)malloc(nl * sizeof(int)); uint8_t **lword = (uint8_t **)malloc(nl * sizeof(uint8_t *));
    for (int i = 0; i < nl; i++) { ldepth[i] = node_depth(&T, leaves.v[i]); lword[i] = (uint8_t *)malloc(ldepth[i] + 1); node_word(&T, leaves.v[i], lword[i], ldepth[i]); }
    /* get_sequence_clusters findflanks.py:1031-1077 */
    int *clus_of = (int *)malloc(nl * sizeof(int)); int nclus = 0;      /* cluster index per leaf, list order */
    if (nl == 1) { clus_of[0] = 0; nclus = 1; }
    else {
        int *par = (int *)malloc(nl * sizeof(int)), *has = (int *)calloc(nl, sizeof(int));
        for (int i = 0; i < nl; i++) par[i] = i;
        for (int i = 0; i < nl; i++) for (int j = i + 1; j < nl; j++) {
            int m = ldepth[i] < ldepth[j] ? ldepth[i] : ldepth[j];
            int d = levenshtein(lword[i], m, lword[j], m);
            double similarity = 1 - ((double)d / (double)m);
            if (similarity >= 0.75) { has[i] = has[j] = 1; int a = uf_find(par, i), b = uf_find(par, j); if (a != b) { if (a < b) par[b] = a; else par[a] = b; } }
        }
        /* singletons first, in leaf order (quirk Q12 build rule), then components in order of their smallest leaf */
        for (int i = 0; i < nl; i++) if (!has[i]) clus_of[i] = nclus++;
        int *label = (int *)malloc(nl * sizeof(int)); for (int i = 0; i < nl; i++) label[i] = -1;
        for (int i = 0; i < nl; i++) if (has[i]) { int root = uf_find(par, i); if (label[root] < 0) label[root] = nclus++; clus_of[i] = label[root]; }
        free(par); free(has); free(label);
    }
    /* get_cluster_consensus_seqs + merge_cluster_counts */
    *ncons = nclus; *ccount = (int32_t *)calloc(nclus, sizeof(int32_t)); *cseq = (char **)calloc(nclus, sizeof(char *));
    int32_t *mark = (int32_t *)malloc(T.cnt * sizeof(int32_t));
    for (int c = 0; c < nclus; c++) {
        for (int32_t i = 0; i < T.cnt; i++) mark[i] = 0;
        int32_t words = 0; int maxd = 0;
        for (int i = 0; i < nl; i++) if (clus_of[i] == c) {
            if (ldepth[i] > maxd) maxd = ldepth[i];
            for (int32_t nd = leaves.v[i]; nd > 0 && !mark[nd]; nd = T.n[nd].parent) { mark[nd] = 1; words += T.n[nd].word_count; }  /* make_subtrie flanktrie.py:89-118 */
        }
        (*ccount)[c] = words;
        /* make_consensus_word flanktrie.py:306-342: level-by-level vote over the marked sub-trie */
        char *cons = (char *)malloc(maxd + 1); int clen = 0;
        ivec cur = {0}, nxt = {0};
        for (int32_t ch = T.n[0].first_child; ch >= 0; ch = T.n[ch].next_sib) if (mark[ch]) iv_push(&cur, ch);
        while (cur.n > 0) {
            int64_t tot = 0, cq[16]; int present[16];
            for (int k = 0; k < 16; k++) { cq[k] = 0; present[k] = 0; }
            for (int64_t k = 0; k < cur.n; k++) { tnode *x = &T.n[cur.v[k]]; cq[x->ch] += x->qual; present[x->ch] = 1; tot += x->counter; }
            if (tot < P->min_count_consensus) break;
            int best = -1, nbest = 0;
            for (int k = 0; k < 16; k++) if (present[k]) { if (best < 0 || cq[k] > cq[best]) { best = k; nbest = 1; } else if (cq[k] == cq[best]) nbest++; }
            cons[clen++] = nbest > 1 ? 'N' : BASES[best];
            nxt.n = 0;
            for (int64_t k = 0; k < cur.n; k++) for (int32_t ch = T.n[cur.v[k]].first_child; ch >= 0; ch = T.n[ch].next_sib) if (mark[ch]) iv_push(&nxt, ch);
            ivec t = cur; cur = nxt; nxt = t;
        }
        cons[clen] = 0;
        if (!side5) for (int a = 0, b = clen - 1; a < b; a++, b--) { char t = cons[a]; cons[a] = cons[b]; cons[b] = t; }   /* findflanks.py:894-895 */
        (*cseq)[c] = cons;
        iv_free(&cur); iv_free(&nxt);
    }
    free(mark);
    for (int i = 0; i < nl; i++) free(lword[i]);
    free(lword); free(ldepth); free(clus_of); iv_free(&leaves); free(T.n);
    return 0;
}

/* filter_consensus_sequences_minlength / _mincount findflanks.py:241-323 (mode 0: length, 1: count) */
static void filter_cons_side(int *keep, int *ncons, int32_t *cnt, char **seq, int mode, const mfk_params *P) {
    if (!*keep) return;
    int k = 0;
    for (int i = 0; i < *ncons; i++) {
        int ok = mode == 0 ? ((int)strlen(seq[i]) >= P->min_softclip_length) : (cnt[i] >= P->min_softclip_count);
        if (ok) { cnt[k] = cnt[i]; seq[k] = seq[i]; k++; } else free(seq[i]);
    }
    *ncons = k;
    if (k == 0) *keep = 0;
}
/* filter_multiple_consensus_sequences + get_best_consensus_sequence findflanks.py:325-346: stable sort, take last */
static void best_cons_side(int *ncons, int32_t *cnt, char **seq) {
    if (*ncons <= 1) return;
    int best = 0;
    for (int i = 1; i < *ncons; i++) {
        int li = (int)strlen(seq[i]), lb = (int)strlen(seq[best]);
        if (cnt[i] > cnt[best] || (cnt[i] == cnt[best] && li >= lb)) best = i;
    }
    for (int i = 0; i < *ncons; i++) if (i != best) free(seq[i]);
    cnt[0] = cnt[best]; seq[0] = seq[best]; *ncons = 1;
}

static void free_sites(octx *C) {
    for (int64_t i = 0; i < C->n_sites; i++) {
        site_t *s = &C->sites[i];
        iv_free(&s->ev5); iv_free(&s->ev3);
        for (int k = 0; k < s->ncons5; k++) free(s->cons_seq5[k]);
        for (int k = 0; k < s->ncons3; k++) free(s->cons_seq3[k]);
        free(s->cons_seq5); free(s->cons_seq3); free(s->cons_count5); free(s->cons_count3);
    }
    free(C->sites); C->sites = NULL; C->n_sites = 0;
    free(C->name_id); C->name_id = NULL;
    free(C->flanks); C->flanks = NULL; free(C->seqs); C->seqs = NULL; C->n_flanks = 0; C->seq_bytes = 0;
}

static void build_name_ids(octx *C, const mfk_reads *R) {
    free(C->name_id);
    C->name_id = (int32_t *)malloc((R->n_reads + 1) * sizeof(int32_t));
    int32_t next = 0;
    for (int64_t r = 0; r < R->n_reads; r++) C->name_id[r] = R->link[r] < 0 ? next++ : C->name_id[R->link[r]];
}

void *mfko_create(int device_id) { (void)device_id; return calloc(1, sizeof(octx)); }
void mfko_destroy(void *c) { octx *C = (octx *)c; if (!C) return; free_sites(C); free(C->pairs); free(C->strs); free(C); }
const char *mfko_last_error(void *c) { return ((octx *)c)->err; }
int mfko_version(void) { return 1; }
double mfko_last_device_ms(void *c) { (void)c; return 0.0; }

int mfko_findflanks(void *c, const mfk_reads *R, const mfk_params *P, int stop_after, int64_t *n_sites,
                    int64_t *n_flanks, int64_t *seq_bytes) {
    octx *C = (octx *)c;
    free_sites(C);
    memset(C->stats, 0, sizeof C->stats);
    build_name_ids(C, R);
    /* ---- parse_softclips findflanks.py:100-121 ---- */
    event_t *ev = NULL; int64_t nev = 0, cap = 0;
    for (int64_t r = 0; r < R->n_reads; r++) {
        if (!(R->flags[r] & 15)) continue;
        if (!passes_read_filters(R, P, r)) continue;
        if (nev + 2 > cap) { cap = cap ? cap * 2 : 1024; ev = (event_t *)realloc(ev, cap * sizeof(event_t)); }
        if (rkind(R, r)) { ev[nev].contig = R->contig_id[r]; ev[nev].pos = rsite(R, r); ev[nev].side = 5; ev[nev].read = (int32_t)r; nev++; }
        if (lkind(R, r)) { ev[nev].contig = R->contig_id[r]; ev[nev].pos = lsite(R, r); ev[nev].side = 3; ev[nev].read = (int32_t)r; nev++; }
    }
    C->stats[10] = nev;
    if (nev) qsort(ev, nev, sizeof(event_t), cmp_event);
    int64_t ns = 0;
    for (int64_t i = 0; i < nev; i++) if (i == 0 || ev[i].contig != ev[i - 1].contig || ev[i].pos != ev[i - 1].pos) ns++;
    C->sites = (site_t *)calloc(ns ? ns : 1, sizeof(site_t)); C->n_sites = ns;
    int64_t si = -1;
    for (int64_t i = 0; i < nev; i++) {
        if (i == 0 || ev[i].contig != ev[i - 1].contig || ev[i].pos != ev[i - 1].pos) {
            si++; site_t *s = &C->sites[si]; s->contig = ev[i].contig; s->pos = ev[i].pos; s->keep5 = s->keep3 = 1; s->alive = 1;
        }
        site_t *s = &C->sites[si]; int64_t r = ev[i].read;
        if (ev[i].side == 5) { iv_push(&s->ev5, (int32_t)r); if ((int)R->last_op_len[r] >= P->min_softclip_length) s->meets5 = 1; }
        else { iv_push(&s->ev3, (int32_t)r); if ((int)R->first_op_len[r] >= P->min_softclip_length) s->meets3 = 1; }
    }
    free(ev);
    for (int64_t i = 0; i < ns; i++) { site_t *s = &C->sites[i]; s->c[0] = distinct_names(C, &s->ev5); s->c[1] = distinct_names(C, &s->ev3); }
    C->stats[0] = count_alive(C);
    if (stop_after <= MFK_STAGE_PARSE) goto done;
    /* ---- filter_softclips_minlength findflanks.py:124-145 ---- */
    for (int64_t i = 0; i < ns; i++) { site_t *s = &C->sites[i]; if (!s->meets5) s->keep5 = 0; if (!s->meets3) s->keep3 = 0; if (!(s->keep5 || s->keep3)) s->alive = 0; }
    C->stats[1] = count_alive(C);
    if (stop_after <= MFK_STAGE_MINLENGTH) goto done;
    /* ---- filter_softclips_mincount findflanks.py:148-169 ---- */
    for (int64_t i = 0; i < ns; i++) { site_t *s = &C->sites[i]; if (!s->alive) continue; if (s->c[0] < P->min_softclip_count) s->keep5 = 0; if (s->c[1] < P->min_softclip_count) s->keep3 = 0; if (!(s->keep5 || s->keep3)) s->alive = 0; }
    C->stats[2] = count_alive(C);
    if (stop_after <= MFK_STAGE_MINCOUNT) goto done;
    filter_mindistance(C, P);
    C->stats[3] = count_alive(C);
    if (stop_after <= MFK_STAGE_MINDIST1) goto done;
    /* ---- parse_unclipped_read_info findflanks.py:349-396 ---- */
    for (int64_t i = 0; i < ns; i++) {
        site_t *s = &C->sites[i];
        if (!s->alive) continue;
        int32_t c10[10];
        tally_site(C, R, P, s->contig, s->pos, c10);
        for (int k = 0; k < 8; k++) s->c[2 + k] = c10[2 + k];
    }
    if (stop_after <= MFK_STAGE_UNCLIPPED) goto done;
    /* ---- filter_softclips_count_ratios findflanks.py:203-239; ratios findflanks.py:827-845 ---- */
    for (int64_t i = 0; i < ns; i++) {
        site_t *s = &C->sites[i];
        if (!s->alive) continue;
        double tot = (double)total_count(s);
        if (s->keep5) { double ratio = (double)(s->c[0] + s->c[5] + s->c[6]) / tot, indel = (double)(s->c[3] + s->c[7]) / tot; if (ratio < P->min_softclip_ratio || indel > P->max_indel_ratio) s->keep5 = 0; }
        if (s->keep3) { double ratio = (double)(s->c[1] + s->c[5] + s->c[6]) / tot, indel = (double)(s->c[4] + s->c[7]) / tot; if (ratio < P->min_softclip_ratio || indel > P->max_indel_ratio) s->keep3 = 0; }
        if (s->keep5 || s->keep3) { double down = (double)s->c[9] / tot, up = (double)s->c[8] / tot; if (down > P->max_indel_ratio || up > P->max_indel_ratio) { s->keep5 = 0; s->keep3 = 0; } }
        if (!(s->keep5 || s->keep3)) s->alive = 0;
    }
    C->stats[4] = count_alive(C);
    if (stop_after <= MFK_STAGE_RATIOS) goto done;
    filter_mindistance(C, P);
    C->stats[5] = count_alive(C);
    if (stop_after <= MFK_STAGE_MINDIST2) goto done;
    /* ---- make_consensus_sequences findflanks.py:512-532 ---- */
    for (int64_t i = 0; i < ns; i++) {
        site_t *s = &C->sites[i];
        if (!s->alive) continue;
        int rc = 0;
        if (s->keep5) rc = make_consensus(C, R, P, &s->ev5, 1, &s->ncons5, &s->cons_count5, &s->cons_seq5);
        if (!rc && s->keep3) rc = make_consensus(C, R, P, &s->ev3, 0, &s->ncons3, &s->cons_count3, &s->cons_seq3);
        if (rc == 1) { snprintf(C->err, sizeof C->err, "READ NUMBER ERROR - ZERO OR MORE THAN TWO READS ASSOCIATED WITH QUERY NAME"); return 3; }
        if (rc) return rc;
    }
    for (int mode = 0; mode < 2; mode++) {
        int64_t nseq = 0;
        for (int64_t i = 0; i < ns; i++) {
            site_t *s = &C->sites[i];
            if (!s->alive) continue;
            filter_cons_side(&s->keep5, &s->ncons5, s->cons_count5, s->cons_seq5, mode, P);
            filter_cons_side(&s->keep3, &s->ncons3, s->cons_count3, s->cons_seq3, mode, P);
            if (!(s->keep5 || s->keep3)) s->alive = 0; else nseq += s->ncons5 + s->ncons3;
        }
        C->stats[6 + mode] = nseq;
    }
    { int64_t nseq = 0; for (int64_t i = 0; i < ns; i++) { site_t *s = &C->sites[i]; if (!s->alive) continue; best_cons_side(&s->ncons5, s->cons_count5, s->cons_seq5); best_cons_side(&s->ncons3, s->cons_count3, s->cons_seq3); nseq += s->ncons5 + s->ncons3; } C->stats[8] = nseq; }
    if (stop_after <= MFK_STAGE_CONSENSUS) goto done;
    filter_mindistance(C, P);
    C->stats[9] = count_alive(C);
    /* ---- make_dataframe findflanks.py:631-670 ---- */
    {
        int64_t nf = 0, sb = 0;
        for (int64_t i = 0; i < ns; i++) { site_t *s = &C->sites[i]; if (!s->alive) continue; if (s->keep5) for (int k = 0; k < s->ncons5; k++) { nf++; sb += strlen(s->cons_seq5[k]); } if (s->keep3) for (int k = 0; k < s->ncons3; k++) { nf++; sb += strlen(s->cons_seq3[k]); } }
        C->flanks = (mfk_flank *)calloc(nf ? nf : 1, sizeof(mfk_flank)); C->seqs = (char *)malloc(sb + 1); C->n_flanks = nf; C->seq_bytes = sb;
        int64_t f = 0, o = 0;
        for (int64_t i = 0; i < ns; i++) {
            site_t *s = &C->sites[i];
            if (!s->alive) continue;
            for (int side = 0; side < 2; side++) {
                int keep = side == 0 ? s->keep5 : s->keep3, nc = side == 0 ? s->ncons5 : s->ncons3;
                if (!keep) continue;
                for (int k = 0; k < nc; k++) {
                    mfk_flank *F = &C->flanks[f++];
                    const char *sq = side == 0 ? s->cons_seq5[k] : s->cons_seq3[k];
                    F->contig_id = s->contig; F->pos = s->pos; F->orient = side == 0 ? 5 : 3;
                    F->softclip_count_5p = s->c[0]; F->softclip_count_3p = s->c[1]; F->runthrough_count = s->c[2];
                    F->small_insertion_count_5p = s->c[3]; F->small_insertion_count_3p = s->c[4];
                    F->large_insertion_count_5p = s->c[5]; F->large_insertion_count_3p = s->c[6];
                    F->deletion_count = s->c[7]; F->upstream_deletion_count = s->c[8]; F->downstream_deletion_count = s->c[9];
                    F->total_count = total_count(s);
                    F->consensus_softclip_count = side == 0 ? s->cons_count5[k] : s->cons_count3[k];
                    F->seq_len = (int32_t)strlen(sq); F->seq_off = o; memcpy(C->seqs + o, sq, F->seq_len); o += F->seq_len;
                }
            }
        }
        C->stats[11] = nf;
    }
done:
    if (n_sites) *n_sites = count_alive(C);
    if (n_flanks) *n_flanks = C->n_flanks;
    if (seq_bytes) *seq_bytes = C->seq_bytes;
    return 0;
}

int mfko_findflanks_fetch(void *c, mfk_site *sites, mfk_flank *flanks, char *seqs, int64_t *stats) {
    octx *C = (octx *)c;
    if (sites) {
        int64_t k = 0;
        for (int64_t i = 0; i < C->n_sites; i++) {
            site_t *s = &C->sites[i];
            if (!s->alive) continue;
            mfk_site *o = &sites[k++];
            o->contig_id = s->contig; o->pos = s->pos;
            o->softclip_count_5p = s->c[0]; o->softclip_count_3p = s->c[1]; o->runthrough_count = s->c[2];
            o->small_insertion_count_5p = s->c[3]; o->small_insertion_count_3p = s->c[4];
            o->large_insertion_count_5p = s->c[5]; o->large_insertion_count_3p = s->c[6];
            o->deletion_count = s->c[7]; o->upstream_deletion_count = s->c[8]; o->downstream_deletion_count = s->c[9];
            o->keep_5p = s->keep5; o->keep_3p = s->keep3;
        }
    }
    if (flanks && C->n_flanks) memcpy(flanks, C->flanks, C->n_flanks * sizeof(mfk_flank));
    if (seqs && C->seq_bytes) memcpy(seqs, C->seqs, C->seq_bytes);
    if (stats) memcpy(stats, C->stats, sizeof C->stats);
    return 0;
}

/* ========================================================================================================== */
/* inverted repeats: restatement of einverted's model (PARITY UNPINNED, see header)                            */
#define IR_MAXLEN 256
#define IR_MAXREP 8
typedef struct { int a1s, a1e, a2s, a2e; } ir_t;   /* 0-based inclusive */
static int ir_code(char ch) { switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; } }

static int ir_detect(const char *s, int L, ir_t *out) {
    const int MATCH = 3, MISMATCH = -4, GAP = 12, THRESHOLD = 15;
    static __thread int16_t S[IR_MAXLEN][IR_MAXLEN];
    uint8_t code[IR_MAXLEN], maskI[IR_MAXLEN];
    if (L > IR_MAXLEN) L = IR_MAXLEN;
    for (int i = 0; i < L; i++) { code[i] = (uint8_t)ir_code(s[i]); maskI[i] = 0; }
    int nrep = 0;
    while (nrep < IR_MAXREP) {
        int best = 0, bi = -1, bj = -1;
        /* spans d = j - i ascending: S(i,j) depends on S(i+1,j-1), S(i+1,j), S(i,j-1) */
        for (int d = 1; d < L; d++) for (int i = 0; i + d < L; i++) {
            int j = i + d, v = 0;
            if (!maskI[i] && !maskI[j]) {
                int diag = (d >= 2) ? S[i + 1][j - 1] : 0;
                int m = (code[i] < 4 && code[j] < 4 && code[i] + code[j] == 3) ? MATCH : MISMATCH;
                v = diag + m;
                int up = S[i + 1][j] - GAP, left = S[i][j - 1] - GAP;
                if (up > v) v = up;
                if (left > v) v = left;
                if (v < 0) v = 0;
            }
            S[i][j] = (int16_t)v;
        }
        for (int i = 0; i < L; i++) S[i][i] = 0;
        /* best cell: highest score; ties -> smallest j, then largest i (first met scanning j up, i down) */
        for (int j = 1; j < L; j++) for (int i = j - 1; i >= 0; i--) if (S[i][j] > best) { best = S[i][j]; bi = i; bj = j; }
        if (best < THRESHOLD) break;
        int i = bi, j = bj, ii = bi, jj = bj;
        while (i < j && S[i][j] > 0) {
            int diag = (j - i >= 2) ? S[i + 1][j - 1] : 0;
            int m = (code[i] < 4 && code[j] < 4 && code[i] + code[j] == 3) ? MATCH : MISMATCH;
            ii = i; jj = j;
            if (S[i][j] == diag + m) { i++; j--; }
            else if (S[i][j] == S[i][j - 1] - GAP) j--;
            else i++;
        }
        out[nrep].a1s = bi; out[nrep].a1e = ii; out[nrep].a2s = jj; out[nrep].a2e = bj;
        nrep++;
        for (int k = bi; k <= ii; k++) maskI[k] = 1;
        for (int k = jj; k <= bj; k++) maskI[k] = 1;
    }
    /* report order: scan order of the outer end (a2e ascending, then a1s descending) */
    for (int a = 1; a < nrep; a++) { ir_t t = out[a]; int b = a - 1; while (b >= 0 && (out[b].a2e > t.a2e || (out[b].a2e == t.a2e && out[b].a1s < t.a1s))) { out[b + 1] = out[b]; b--; } out[b + 1] = t; }
    return nrep;
}

/* the reference's acceptance loop pairflanks.py:196-205 with pair_has_ir 253-266 over the reported repeats */
static int ir_accept(const char *s, int L, int32_t *res) {
    ir_t rep[IR_MAXREP]; int n = ir_detect(s, L, rep);
    int has = 0, k1s = 0, k1e = 0, k2s = 0, k2e = 0;
    for (int k = 0; k < n; k++) {
        int ir1_pos_5p = rep[k].a1s + 1, ir2_pos_3p = rep[k].a2e + 1;
        if (ir1_pos_5p <= 15 && ir2_pos_3p >= L - 15) { has = 1; if (rep[k].a1e - rep[k].a1s + 1 > 0) { k1s = rep[k].a1s + 1; k1e = rep[k].a1e + 1; k2s = rep[k].a2s + 1; k2e = rep[k].a2e + 1; } }
    }
    res[0] = has; res[1] = k1s; res[2] = k1e; res[3] = k2s; res[4] = k2e; res[5] = n;
    return has;
}
int mfko_ir_detect(void *c, const char *seqs, const int64_t *off, int64_t n, int32_t *out) {
    (void)c;
    for (int64_t i = 0; i < n; i++) ir_accept(seqs + off[i], (int)(off[i + 1] - off[i]), out + 6 * i);
    return 0;
}

/* ========================================================================================================== */
typedef struct { int32_t i5, i3; int has_ir, irlen, diffcount, difflength; int32_t ir[4]; int order; } cand_t;
static int cmp_cand(const void *a, const void *b) {   /* sort_values(['IR_length','diffcount','difflength'], [False,True,True]), stable */
    const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
    if (x->irlen != y->irlen) return x->irlen > y->irlen ? -1 : 1;
    if (x->diffcount != y->diffcount) return x->diffcount < y->diffcount ? -1 : 1;
    if (x->difflength != y->difflength) return x->difflength < y->difflength ? -1 : 1;
    return x->order - y->order;
}
typedef struct { int32_t rank, pos5, pos3; int order; cand_t c; } kept_t;
static int cmp_kept(const void *a, const void *b) {   /* sort_values(['contig','pos_5p','pos_3p']) pairflanks.py:248 */
    const kept_t *x = (const kept_t *)a, *y = (const kept_t *)b;
    if (x->rank != y->rank) return x->rank < y->rank ? -1 : 1;
    if (x->pos5 != y->pos5) return x->pos5 < y->pos5 ? -1 : 1;
    if (x->pos3 != y->pos3) return x->pos3 < y->pos3 ? -1 : 1;
    return x->order - y->order;
}

/* identify_large_insertion_at_site pairflanks.py:446-470 (quirk P8) */
static int large_insertion_at(const mfk_reads *R, const mfk_params *P, int64_t r, int32_t position) {
    blocks_t b = get_blocks(R, r);
    for (int i = 0; i + 1 < b.n; i++) {
        int32_t e1 = blk_e(&b, i), s2 = blk_s(&b, i + 1);
        if (e1 == s2) {
            if (position == e1 - 1 || position == s2) return blk_i(&b, i + 1) >= P->large_insertion_cutoff;
        } else return 0;
    }
    return 0;
}

int mfko_pairflanks(void *c, const mfk_reads *R, const mfk_params *P, const mfk_flank *F, const char *fseq, int64_t nF,
                    const int32_t *contig_rank, const uint8_t *genome, const int64_t *goff, int64_t *n_pairs, int64_t *str_bytes) {
    octx *C = (octx *)c;
    free(C->pairs); free(C->strs); C->pairs = NULL; C->strs = NULL; C->n_pairs = 0; C->str_bytes = 0;
    memset(C->pstats, 0, sizeof C->pstats);
    build_name_ids(C, R);
    /* ---- pair_all_nearby_flanks pairflanks.py:138-169 ---- */
    cand_t *cand = NULL; int64_t nc = 0, cap = 0;
    for (int64_t a = 0; a < nF; a++) {
        if (F[a].orient != 5) continue;
        int32_t min_pos = F[a].pos - P->max_direct_repeat_length - 1;
        for (int64_t b = 0; b < nF; b++) {
            if (F[b].orient != 3 || F[b].contig_id != F[a].contig_id || F[b].pos < min_pos || F[b].pos >= F[a].pos) continue;
            if (nc == cap) { cap = cap ? cap * 2 : 64; cand = (cand_t *)realloc(cand, cap * sizeof(cand_t)); }
            cand_t *k = &cand[nc]; memset(k, 0, sizeof *k); k->i5 = (int32_t)a; k->i3 = (int32_t)b; k->order = (int)nc; nc++;
        }
    }
    C->pstats[0] = nc;
    /* ---- check_pairs_for_ir pairflanks.py:172-219 ---- */
    for (int64_t k = 0; k < nc; k++) {
        const mfk_flank *f5 = &F[cand[k].i5], *f3 = &F[cand[k].i3];
        char comb[IR_MAXLEN]; int L = 0;
        int n5 = f5->seq_len > 40 ? 40 : f5->seq_len, n3 = f3->seq_len > 40 ? 40 : f3->seq_len;   /* truncate_sequence 268-275 */
        memcpy(comb, fseq + f5->seq_off, n5); L = n5;
        memset(comb + L, 'N', 20); L += 20;
        memcpy(comb + L, fseq + f3->seq_off + (f3->seq_len - n3), n3); L += n3;
        int32_t res[6];
        if (C->n_ir_ov) {   /* table from an external detector (mfk_pairflanks_set_ir) */
            int64_t q = 0;
            for (; q < C->n_ir_ov; q++) if (C->ir_ov[7 * q] == cand[k].i5 && C->ir_ov[7 * q + 1] == cand[k].i3) break;
            if (q == C->n_ir_ov) { snprintf(C->err, sizeof C->err, "candidate pair (%d, %d) has no entry in the inverted-repeat table", cand[k].i5, cand[k].i3); free(cand); free(C->ir_ov); C->ir_ov = NULL; C->n_ir_ov = 0; return 8; }
            for (int z = 0; z < 5; z++) res[z] = C->ir_ov[7 * q + 2 + z];
            res[5] = -1;
        } else ir_accept(comb, L, res);
        cand[k].has_ir = res[0];
        /* IR_3p = ir1.seq, IR_5p = ir2.seq (pairflanks.py:212-213); IR_length = len(IR_5p) (225) */
        cand[k].ir[0] = res[1]; cand[k].ir[1] = res[2]; cand[k].ir[2] = res[3]; cand[k].ir[3] = res[4];
        cand[k].irlen = res[1] ? res[4] - res[3] + 1 : 0;
        int dc = f5->consensus_softclip_count - f3->consensus_softclip_count; cand[k].diffcount = dc < 0 ? -dc : dc;
        int dl = f5->seq_len - f3->seq_len; cand[k].difflength = dl < 0 ? -dl : dl;
    }
    /* ---- assign_pairs pairflanks.py:222-250 ---- */
    if (nc) qsort(cand, nc, sizeof(cand_t), cmp_cand);
    uint8_t *used5 = (uint8_t *)calloc(nF + 1, 1), *used3 = (uint8_t *)calloc(nF + 1, 1);
    kept_t *kept = (kept_t *)malloc((nc + 1) * sizeof(kept_t)); int64_t nk = 0;
    for (int64_t k = 0; k < nc; k++) {
        if (used5[cand[k].i5] || used3[cand[k].i3]) continue;
        used5[cand[k].i5] = 1; used3[cand[k].i3] = 1;
        kept[nk].c = cand[k]; kept[nk].rank = contig_rank[F[cand[k].i5].contig_id]; kept[nk].pos5 = F[cand[k].i5].pos; kept[nk].pos3 = F[cand[k].i3].pos; kept[nk].order = (int)nk; nk++;
    }
    free(used5); free(used3); free(cand);
    if (nk) qsort(kept, nk, sizeof(kept_t), cmp_kept);
    C->pstats[1] = nk;
    /* ---- count_insertion_spanning_reads pairflanks.py:370-391 ---- */
    int32_t *spanning = (int32_t *)calloc(nk + 1, sizeof(int32_t));
    for (int64_t k = 0; k < nk; k++) {
        int32_t contig = F[kept[k].c.i5].contig_id, pos3 = kept[k].pos3, pos5 = kept[k].pos5, clen = R->h_contig_len[contig];
        int32_t start = pos3 - 1, end = pos5 + 2;
        if (start < 0) start = 0;
        if (end > clen) end = clen;
        ivec fetched = {0}, names = {0};
        fetch_reads(R, contig, start, end, &fetched);
        for (int64_t q = 0; q < fetched.n; q++) {
            int64_t r = fetched.v[q];
            if (!passes_read_filters(R, P, r)) continue;
            if (R->ref_start[r] < pos3 - 10 && R->ref_end[r] > pos5 + 10 + 1 &&
                !(large_insertion_at(R, P, r, pos3) || large_insertion_at(R, P, r, pos5))) iv_push(&names, (int32_t)r);
        }
        spanning[k] = distinct_names(C, &names);
        if (kept[k].c.has_ir) C->pstats[2]++;
        if (spanning[k] > 0) C->pstats[3]++;
        iv_free(&fetched); iv_free(&names);
    }
    /* ---- filter_junction_spanning pairflanks.py:421-431 ---- */
    int64_t nfin = 0;
    for (int64_t k = 0; k < nk; k++) {
        double total = (double)(F[kept[k].c.i5].total_count + F[kept[k].c.i3].total_count) / 2;
        double prop = (double)spanning[k] / total;
        if (prop < P->max_junction_spanning_prop) { kept[nfin] = kept[k]; spanning[nfin] = spanning[k]; nfin++; }
    }
    C->pstats[4] = nfin;
    /* ---- get_direct_repeats pairflanks.py:277-367 ---- */
    int64_t sb = 0;
    for (int64_t k = 0; k < nfin; k++) { sb += 2 * (kept[k].pos5 - kept[k].pos3); if (kept[k].c.ir[0]) sb += (kept[k].c.ir[1] - kept[k].c.ir[0] + 1) + (kept[k].c.ir[3] - kept[k].c.ir[2] + 1); }
    C->pairs = (mfk_pair *)calloc(nfin ? nfin : 1, sizeof(mfk_pair)); C->strs = (char *)malloc(sb + 16); int64_t so = 0;
    for (int64_t k = 0; k < nfin; k++) {
        const mfk_flank *f5 = &F[kept[k].c.i5], *f3 = &F[kept[k].c.i3];
        mfk_pair *o = &C->pairs[k];
        int32_t contig = f5->contig_id, start = kept[k].pos3, end = kept[k].pos5;
        int64_t glen = goff[contig + 1] - goff[contig]; const uint8_t *g = genome + goff[contig];
        o->contig_id = contig; o->pos_5p = end; o->pos_3p = start; o->index_5p = kept[k].c.i5; o->index_3p = kept[k].c.i3;
        o->softclip_count_5p = f5->consensus_softclip_count; o->softclip_count_3p = f3->consensus_softclip_count;
        o->total_count_5p = f5->total_count; o->total_count_3p = f3->total_count; o->spanning_count = spanning[k];
        o->has_ir = kept[k].c.has_ir; o->ir_length = kept[k].c.irlen; o->ir5_len = o->ir3_len = -1;
        if (kept[k].c.ir[0]) {
            /* rebuild the combined sequence to slice the arms */
            char comb[IR_MAXLEN]; int L = 0; int n5 = f5->seq_len > 40 ? 40 : f5->seq_len, n3 = f3->seq_len > 40 ? 40 : f3->seq_len;
            memcpy(comb, fseq + f5->seq_off, n5); L = n5; memset(comb + L, 'N', 20); L += 20; memcpy(comb + L, fseq + f3->seq_off + (f3->seq_len - n3), n3); L += n3;
            o->ir3_len = kept[k].c.ir[1] - kept[k].c.ir[0] + 1; o->ir3_off = so; memcpy(C->strs + so, comb + kept[k].c.ir[0] - 1, o->ir3_len); so += o->ir3_len;
            o->ir5_len = kept[k].c.ir[3] - kept[k].c.ir[2] + 1; o->ir5_off = so; memcpy(C->strs + so, comb + kept[k].c.ir[2] - 1, o->ir5_len); so += o->ir5_len;
        }
        /* get_reference_direct_repeats pairflanks.py:356-367: genome[start+1:end] with Python slice clamping */
        int64_t a = start + 1, b = end; if (a < 0) a = 0; if (b > glen) b = glen; if (b < a) b = a;
        o->dr_ref_off = so; int dr_ref_len = (int)(b - a); memcpy(C->strs + so, g + a, dr_ref_len); so += dr_ref_len;
        /* get_read_direct_repeats pairflanks.py:288-329 */
        /* round((end + start) / 2): Python 3 rounds x.5 to the even neighbour; sum >= -1 here */
        int64_t sum = (int64_t)end + start, center;
        if (sum % 2 == 0) center = sum / 2;
        else { int64_t lo = (sum - 1) / 2; center = (lo % 2 == 0) ? lo : lo + 1; }
        int64_t es = center - 25, ee = center + 25, add_start_n = 0;
        if (es < 0) { add_start_n = -es; es = 0; }
        if (ee > glen) ee = glen;
        int W = (int)(ee - es); if (W < 0) W = 0;
        /* per position: ordered key list (char, sum); ref base seeded with weight 1 (initialize_target_region_reads 347-353) */
        typedef struct { int nk; uint8_t key[18]; int64_t sum[18]; } col_t;
        col_t *cols = (col_t *)calloc(W ? W : 1, sizeof(col_t));
        for (int p = 0; p < W; p++) {
            int64_t idx = p;   /* target_region[i] where target_region = 'N'*add_start_n + genome[es:ee] + ... (quirk P10) */
            uint8_t ch = idx < add_start_n ? (uint8_t)'N' : ((es + idx - add_start_n) < glen ? g[es + idx - add_start_n] : (uint8_t)'N');
            cols[p].nk = 1; cols[p].key[0] = ch; cols[p].sum[0] = 1;
        }
        ivec fetched = {0};
        if (W > 0) fetch_reads(R, contig, (int32_t)es, (int32_t)ee, &fetched);
        for (int64_t q = 0; q < fetched.n; q++) {
            int64_t r = fetched.v[q]; int64_t pi = pool_index(R, r);
            if (R->qlen[r] == 0) continue;
            if (pi < 0) { snprintf(C->err, sizeof C->err, "read %lld overlaps a direct-repeat window but has no sequence in the pool", (long long)r); free(cols); iv_free(&fetched); free(kept); free(spanning); return 4; }
            uint64_t o0 = R->pool_off[pi];
            blocks_t bl = get_blocks(R, r);
            int64_t qpos = (lkind(R, r) == 1) ? R->first_op_len[r] : 0;   /* + blk_ins[0]: other query bases in front */
            for (int bi = 0; bi < bl.n; bi++) {
                qpos += blk_i(&bl, bi);
                for (int32_t rp = blk_s(&bl, bi); rp < blk_e(&bl, bi); rp++, qpos++) {
                    if (rp < es || rp >= ee) continue;
                    col_t *cl = &cols[rp - es]; uint8_t ch = (uint8_t)BASES[pool_base(R, o0 + qpos)]; int kk;
                    for (kk = 0; kk < cl->nk; kk++) if (cl->key[kk] == ch) break;
                    if (kk == cl->nk) { cl->key[kk] = ch; cl->sum[kk] = 0; cl->nk++; }
                    cl->sum[kk] += R->pool_quals[o0 + qpos];
                }
            }
        }
        iv_free(&fetched);
        /* consensus slice [index(start+1), index(end)) -- .index raises outside the window (pairflanks.py:323) */
        if (start + 1 < es || start + 1 >= ee || end < es || end >= ee) {
            snprintf(C->err, sizeof C->err, "ValueError: direct-repeat bounds %d..%d fall outside the target window %lld..%lld (pairflanks.py:323)", start + 1, end, (long long)es, (long long)ee);
            free(cols); free(kept); free(spanning); return 5;
        }
        o->dr_reads_off = so; o->dr_len = end - start - 1;
        for (int64_t p = start + 1; p < end; p++) {
            col_t *cl = &cols[p - es]; int64_t bq = 0; uint8_t bn = 0;
            for (int kk = 0; kk < cl->nk; kk++) if (cl->sum[kk] > bq) { bq = cl->sum[kk]; bn = cl->key[kk]; }   /* get_consensus_target_region 332-344 */
            C->strs[so++] = (char)bn;
        }
        o->dr_ref_len = dr_ref_len;   /* differs from dr_len only where the slice is clamped at a contig end */
        free(cols);
    }
    free(kept); free(spanning);
    free(C->ir_ov); C->ir_ov = NULL; C->n_ir_ov = 0;   /* the table serves one call */
    C->n_pairs = nfin; C->str_bytes = so;
    if (n_pairs) *n_pairs = nfin;
    if (str_bytes) *str_bytes = so;
    return 0;
}

int mfko_pairflanks_fetch(void *c, mfk_pair *pairs, char *strs, int64_t *stats) {
    octx *C = (octx *)c;
    if (pairs && C->n_pairs) memcpy(pairs, C->pairs, C->n_pairs * sizeof(mfk_pair));
    if (strs && C->str_bytes) memcpy(strs, C->strs, C->str_bytes);
    if (stats) memcpy(stats, C->pstats, sizeof C->pstats);
    return 0;
}

/* every repeat ir_detect reports for one sequence, in report order; out: n x 4 (1-based inclusive a1s a1e a2s a2e).
 * Used by oracle/ref_harness.py to stand in for the einverted binary when the reference itself is run. */
int mfko_ir_report(const char *s, int L, int32_t *out) {
    ir_t rep[IR_MAXREP]; int n = ir_detect(s, L, rep);
    for (int k = 0; k < n; k++) { out[4 * k] = rep[k].a1s + 1; out[4 * k + 1] = rep[k].a1e + 1; out[4 * k + 2] = rep[k].a2s + 1; out[4 * k + 3] = rep[k].a2e + 1; }
    return n;
}

/* ---- recall (recall.py:50-112) -------------------------------------------------------------------------------------- */
int mfko_recall(void *c, const mfk_reads *R, const mfk_params *P, const int32_t *contig_id, const int32_t *pos, int64_t n_sites, int32_t *counts) {
    octx *C = (octx *)c;
    if (!C || !R || !P || n_sites < 0) return 1;
    C->err[0] = 0;
    for (int64_t i = 0; i < n_sites; i++) {
        if (contig_id[i] < 0 || contig_id[i] >= R->n_contigs) { snprintf(C->err, sizeof C->err, "site %lld: contig id %d out of range", (long long)i, contig_id[i]); return 1; }
        if (pos[i] < -1 || pos[i] > R->h_contig_len[contig_id[i]]) { snprintf(C->err, sizeof C->err, "site %lld: position %d outside contig", (long long)i, pos[i]); return 1; }
    }
    build_name_ids(C, R);
    for (int64_t i = 0; i < n_sites; i++) tally_site(C, R, P, contig_id[i], pos[i], counts + 10 * i);
    return 0;
}

/* ---- inferseq-overlap: get_best_sliding_alignment inferseqoverlap.py:94-164, the window state machine as written ---- */
int mfko_overlap(void *c, const char *seqs5, const int64_t *off5, const char *seqs3, const int64_t *off3, int64_t n, int32_t *out) {
    octx *C = (octx *)c;
    if (!C || n < 0) return 1;
    C->err[0] = 0;
    for (int64_t k = 0; k < n; k++) {
        const char *query = seqs5 + off5[k], *ref = seqs3 + off3[k];
        int64_t lq = off5[k + 1] - off5[k], lr = off3[k + 1] - off3[k];
        if (lq <= 0 || lr <= 0) { snprintf(C->err, sizeof C->err, "IndexError: pair %lld has an empty flank sequence (inferseqoverlap.py:126)", (long long)k); return 7; }
        int64_t max_len = lq < lr ? lq : lr;
        int max_len_query = max_len == lq, max_len_ref = !max_len_query;
        int64_t best_score = INT64_MIN, best_mm = 0, best_rs = 0, best_re = 0, best_qs = 0, best_qe = 0;
        int64_t q_start = lq - 1, q_end = lq, r_start = 0, r_end = 1;
        int reached_max = 0;
        while (q_start != q_end) {
            int64_t score = 0, mismatches = 0;
            for (int64_t i = 0; i < q_end - q_start; i++) {
                if (query[q_start + i] == ref[r_start + i]) score += 1;
                else { score -= 1; mismatches += 1; }
            }
            if (score > best_score) { best_score = score; best_mm = mismatches; best_rs = r_start; best_re = r_end; best_qs = q_start; best_qe = q_end; }
            if (q_end - q_start == max_len) reached_max = 1;
            if (!reached_max) { q_start -= 1; r_end += 1; }
            else if (max_len_query) { if (r_end == lr) { r_start += 1; q_end -= 1; } else { r_start += 1; r_end += 1; } }
            else if (max_len_ref) { if (q_start == 0) { r_start += 1; q_end -= 1; } else { q_start -= 1; q_end -= 1; } }
        }
        int32_t *o = out + 6 * k;
        o[0] = (int32_t)best_score; o[1] = (int32_t)best_mm; o[2] = (int32_t)best_rs; o[3] = (int32_t)best_re; o[4] = (int32_t)best_qs; o[5] = (int32_t)best_qe;
    }
    return 0;
}

/* ---- inferseq: pairing of flank alignments, inferseq.py:50-70 with AlignedPairs 185-395, written the way the reference
 * does it: four lists and four "position -> read" maps per pair id, sorted position lists with strict neighbours, a list
 * of pairs that is filtered four times.  Set-iteration order of the reference is replaced by creation order. ---- */
typedef struct { int32_t contig, pos, read; } posmap_t;          /* dict[reference_name][position] = read (last one wins) */
static void pm_set(posmap_t **m, int *n, int32_t contig, int32_t pos, int32_t read) {
    for (int i = 0; i < *n; i++) if ((*m)[i].contig == contig && (*m)[i].pos == pos) { (*m)[i].read = read; return; }
    *m = (posmap_t *)realloc(*m, (size_t)(*n + 1) * sizeof(posmap_t));
    (*m)[*n].contig = contig; (*m)[*n].pos = pos; (*m)[*n].read = read; (*n)++;
}
/* misc.takeClosestLarger / takeClosestSmaller over the keys of one contig: smallest key > x / largest key < x */
static int32_t pm_closest(const posmap_t *m, int n, int32_t contig, int32_t x, int larger) {
    int32_t best = -1, bestpos = 0;
    for (int i = 0; i < n; i++) {
        if (m[i].contig != contig) continue;
        if (larger ? m[i].pos <= x : m[i].pos >= x) continue;
        if (best < 0 || (larger ? m[i].pos < bestpos : m[i].pos > bestpos)) { best = m[i].read; bestpos = m[i].pos; }
    }
    return best;
}
typedef struct { int32_t f, r; int alive; } ipair_t;

int mfko_inferseq_pairs(void *c, const mfk_aln *A, int64_t n, double min_perc_identity, double max_internal_softclip_prop,
                        int64_t max_size, int64_t min_size, int64_t *n_pairs) {
    octx *C = (octx *)c;
    if (!C || n < 0 || !n_pairs) return 1;
    C->err[0] = 0;
    free(C->isq_f); free(C->isq_r); C->isq_f = C->isq_r = NULL; C->n_isq = 0;
    *n_pairs = 0;
    if (n == 0) return 0;
    int32_t n_groups = 0;
    for (int64_t i = 0; i < n; i++) if (A[i].group + 1 > n_groups) n_groups = A[i].group + 1;
    char *keep = (char *)calloc((size_t)n, 1);
    /* __prefilter_reads inferseq.py:103-126 */
    for (int64_t i = 0; i < n; i++) {
        if (A[i].aln_len == 0) { snprintf(C->err, sizeof C->err, "ZeroDivisionError: alignment %lld has query_alignment_length 0 (pysamtools.py:176)", (long long)i); free(keep); return 7; }
        double identity = (double)(A[i].aln_len - A[i].nm) / (double)A[i].aln_len;
        if (identity < min_perc_identity) continue;
        if (!A[i].is_reverse) { if (A[i].left_clip > 0 && A[i].left_clip > 1 && A[i].ref_start - 1 >= 0) continue; }
        else { if (A[i].right_clip > 0 && A[i].right_clip > 1 && A[i].ref_end < A[i].contig_len) continue; }
        keep[i] = 1;
    }
    C->isq_f = (int32_t *)malloc((size_t)(2 * n + 1) * sizeof(int32_t)); C->isq_r = (int32_t *)malloc((size_t)(2 * n + 1) * sizeof(int32_t));
    for (int32_t g = 0; g < n_groups; g++) {
        /* add_read 214-230: lists in BAM order + position maps */
        ivec lists[4] = {{0}}; posmap_t *maps[4] = {0}; int nmap[4] = {0};   /* 0 fwd mate1, 1 fwd mate2, 2 rev mate1, 3 rev mate2 */
        for (int64_t i = 0; i < n; i++) {
            if (A[i].group != g || !keep[i]) continue;
            int k = (A[i].is_reverse ? 2 : 0) + (A[i].mate == 1 ? 0 : 1);
            iv_push(&lists[k], (int32_t)i);
            pm_set(&maps[k], &nmap[k], A[i].contig_id, A[i].is_reverse ? A[i].ref_end : A[i].ref_start, (int32_t)i);
        }
        /* match_forward_reverse 232-256 */
        ipair_t *P = (ipair_t *)malloc((size_t)(lists[0].n + lists[1].n + lists[2].n + lists[3].n + 1) * sizeof(ipair_t));
        int np = 0;
        for (int k = 0; k < 4; k++)
            for (int64_t t = 0; t < lists[k].n; t++) {
                int32_t rd = lists[k].v[t], f, r;
                if (k < 2) {   /* forward read: closest reverse read of the other mate, 258-272 */
                    int other = A[rd].mate == 1 ? 3 : 2;
                    r = pm_closest(maps[other], nmap[other], A[rd].contig_id, A[rd].ref_start, 1); f = rd;
                } else {       /* reverse read: closest forward read of the other mate, 275-289 */
                    int other = A[rd].mate == 1 ? 1 : 0;
                    f = pm_closest(maps[other], nmap[other], A[rd].contig_id, A[rd].ref_end, 0); r = rd;
                }
                if (f < 0 || r < 0) continue;
                int dup = 0;   /* the `added` set of name tuples */
                for (int q = 0; q < np && !dup; q++)
                    dup = A[P[q].f].name == A[f].name && A[P[q].f].contig_id == A[f].contig_id && A[P[q].f].ref_start == A[f].ref_start && A[P[q].f].ref_end == A[f].ref_end &&
                          A[P[q].r].name == A[r].name && A[P[q].r].contig_id == A[r].contig_id && A[P[q].r].ref_start == A[r].ref_start && A[P[q].r].ref_end == A[r].ref_end;
                if (!dup) { P[np].f = f; P[np].r = r; P[np].alive = 1; np++; }
            }
        /* filter_smallest_overlapping 321-362 */
        char *kp = (char *)calloc((size_t)np + 1, 1);
        for (int p = 0; p < np; p++) {
            int nf = 0, nr = 0;
            for (int q = 0; q < np; q++) {
                if (A[P[q].f].name == A[P[p].f].name && A[P[q].f].ref_start == A[P[p].f].ref_start) nf++;
                if (A[P[q].r].name == A[P[p].r].name && A[P[q].r].ref_end == A[P[p].r].ref_end) nr++;
            }
            if (nf == 1 && nr == 1) { kp[p] = 1; continue; }
            int best = -1; int64_t bl = 0;
            for (int q = 0; q < np; q++) {
                int inf = A[P[q].f].name == A[P[p].f].name && A[P[q].f].ref_start == A[P[p].f].ref_start;
                int inr = A[P[q].r].name == A[P[p].r].name && A[P[q].r].ref_end == A[P[p].r].ref_end;
                int in = (nf > 1 && nr == 1) ? inf : (nf == 1 && nr > 1) ? inr : (inf || inr);
                if (!in) continue;
                int64_t len = (int64_t)A[P[q].r].ref_end - A[P[q].f].ref_start;
                if (best < 0 || len < bl) { best = q; bl = len; }
            }
            if (best >= 0) kp[best] = 1;
        }
        for (int p = 0; p < np; p++) P[p].alive = kp[p];
        free(kp);
        /* filter_pairs_max_internal_softclip_prop 364-380 */
        for (int p = 0; p < np; p++) {
            if (!P[p].alive) continue;
            const mfk_aln *f = &A[P[p].f], *r = &A[P[p].r];
            if (f->right_clip > 0 && f->ref_end < r->ref_end && (double)f->right_clip / (double)f->qlen > max_internal_softclip_prop) P[p].alive = 0;
            else if (r->left_clip > 0 && r->ref_start > f->ref_start && (double)r->left_clip / (double)r->qlen > max_internal_softclip_prop) P[p].alive = 0;
        }
        /* filter_best_alignments 305-318 */
        int64_t best_as = INT64_MIN;
        for (int p = 0; p < np; p++) if (P[p].alive) { int64_t s = (int64_t)A[P[p].f].as + A[P[p].r].as; if (s > best_as) best_as = s; }
        for (int p = 0; p < np; p++) if (P[p].alive && (int64_t)A[P[p].f].as + A[P[p].r].as != best_as) P[p].alive = 0;
        /* filter_pairs_size 382-390 */
        for (int p = 0; p < np; p++) {
            if (!P[p].alive) continue;
            int64_t len = (int64_t)A[P[p].r].ref_end - A[P[p].f].ref_start;
            if (len < min_size || len > max_size) P[p].alive = 0;
        }
        for (int p = 0; p < np; p++) if (P[p].alive) { C->isq_f[C->n_isq] = P[p].f; C->isq_r[C->n_isq] = P[p].r; C->n_isq++; }
        free(P);
        for (int k = 0; k < 4; k++) { iv_free(&lists[k]); free(maps[k]); }
    }
    free(keep);
    *n_pairs = C->n_isq;
    return 0;
}

int mfko_inferseq_pairs_fetch(void *c, int32_t *forward, int32_t *reverse) {
    octx *C = (octx *)c;
    if (!C) return 1;
    if (C->n_isq) { memcpy(forward, C->isq_f, (size_t)C->n_isq * sizeof(int32_t)); memcpy(reverse, C->isq_r, (size_t)C->n_isq * sizeof(int32_t)); }
    return 0;
}

/* ---- inferseq-database pairing: inferseqdatabase.py:116-322 restated with the reference's lists, sort and greedy loop ---- */
typedef struct { int32_t i, j; int64_t d; int32_t seq; } couple_t;
static int cmp_couple(const void *a, const void *b) {   /* sorted(key=dist, reverse=True) is stable: ties keep the order of creation */
    const couple_t *x = (const couple_t *)a, *y = (const couple_t *)b;
    if (x->d != y->d) return x->d > y->d ? -1 : 1;
    return (x->seq > y->seq) - (x->seq < y->seq);
}
static int db_keep_read(const mfk_aln *x, double min_ident, double max_prop, int64_t L) {   /* prefilter_reads 142-180 */
    double identity = (double)(x->aln_len - x->nm) / (double)x->aln_len;
    if (identity < min_ident) return 0;
    if (!x->is_reverse) {
        if (x->right_clip > 0 && x->ref_end < x->contig_len && (double)x->right_clip / (double)x->qlen > max_prop) return 0;
        else if (x->ref_start > L) return 0;
        else if (x->left_clip > 0 && llabs(0 - ((long long)x->ref_start - x->left_clip)) > L) return 0;
    }
    if (x->is_reverse) {
        if (x->left_clip > 0 && x->ref_start - 1 >= 0 && (double)x->left_clip / (double)x->qlen > max_prop) return 0;
        else if ((int64_t)x->contig_len - x->ref_end > L) return 0;
        else if (x->right_clip > 0 && llabs(0 - ((long long)x->contig_len - ((long long)x->ref_end + x->right_clip))) > L) return 0;
    }
    return 1;
}
int mfko_inferseq_db_pairs(void *c, const mfk_aln *A, int64_t n, double min_perc_identity, double max_internal_softclip_prop,
                           int64_t max_edge_distance, int64_t *n_pairs) {
    octx *C = (octx *)c;
    if (!C || n < 0 || !n_pairs) return 1;
    C->err[0] = 0;
    free(C->isq_f); free(C->isq_r); C->isq_f = C->isq_r = NULL; C->n_isq = 0;
    *n_pairs = 0;
    if (n == 0) return 0;
    const int64_t L = max_edge_distance;
    int32_t n_groups = 0;
    for (int64_t i = 0; i < n; i++) {
        if (A[i].aln_len == 0) { snprintf(C->err, sizeof C->err, "ZeroDivisionError: alignment %lld has query_alignment_length 0 (pysamtools.py:176)", (long long)i); return 7; }
        if (A[i].qlen == 0) { snprintf(C->err, sizeof C->err, "ZeroDivisionError: alignment %lld has an empty query sequence (sctools.py:182-186)", (long long)i); return 7; }
        if (A[i].group + 1 > n_groups) n_groups = A[i].group + 1;
    }
    char *keep = (char *)calloc((size_t)n, 1);
    for (int64_t i = 0; i < n; i++) keep[i] = (char)db_keep_read(&A[i], min_perc_identity, max_internal_softclip_prop, L);
    C->isq_f = (int32_t *)malloc((size_t)(n + 1) * sizeof(int32_t)); C->isq_r = (int32_t *)malloc((size_t)(n + 1) * sizeof(int32_t));
    for (int32_t g = 0; g < n_groups; g++) {
        ivec contigs = {0};   /* in the order the kept reads of the pair id first name them */
        for (int64_t i = 0; i < n; i++) {
            if (A[i].group != g || !keep[i]) continue;
            int seen = 0;
            for (int64_t k = 0; k < contigs.n; k++) if (contigs.v[k] == A[i].contig_id) seen = 1;
            if (!seen) iv_push(&contigs, A[i].contig_id);
        }
        ivec kf = {0}, kr = {0};   /* keep_reads[pair]: (forward-strand read, reverse-strand read) */
        for (int64_t ci = 0; ci < contigs.n; ci++) {
            ivec r1 = {0}, r2 = {0};
            for (int64_t i = 0; i < n; i++) if (A[i].group == g && keep[i] && A[i].contig_id == contigs.v[ci]) iv_push(A[i].mate == 1 ? &r1 : &r2, (int32_t)i);
            int go = r1.n > 0 && r2.n > 0;
            if (go) {   /* reads_mapped_both_ends 211-232 */
                int f1 = 0, f2 = 0, t1 = 0, t2 = 0;
                for (int pass = 0; pass < 2; pass++) {
                    const ivec *l = pass ? &r2 : &r1;
                    for (int64_t k = 0; k < l->n; k++) {
                        const mfk_aln *x = &A[l->v[k]];
                        if (x->is_reverse && (int64_t)x->contig_len - x->ref_end <= L) { if (x->mate == 1) t1++; else t2++; }
                        else if (!x->is_reverse && x->ref_start <= L) { if (x->mate == 1) f1++; else f2++; }
                    }
                }
                go = (f1 > 0 && t2 > 0) || (t1 > 0 && f2 > 0);
            }
            if (go) {   /* match_pairs 269-300 */
                couple_t *cp = (couple_t *)malloc((size_t)(r1.n * r2.n + 1) * sizeof(couple_t));
                int64_t nc = 0;
                for (int64_t i = 0; i < r1.n; i++) {
                    const mfk_aln *x = &A[r1.v[i]];
                    int64_t p1 = x->is_reverse ? x->ref_end : x->ref_start;
                    for (int64_t j = 0; j < r2.n; j++) {
                        const mfk_aln *y = &A[r2.v[j]];
                        int64_t p2 = y->is_reverse ? y->ref_end : x->ref_start;   /* line 284 reads read1.reference_start */
                        cp[nc].i = (int32_t)i; cp[nc].j = (int32_t)j; cp[nc].d = llabs((long long)(p1 - p2)); cp[nc].seq = (int32_t)nc; nc++;
                    }
                }
                qsort(cp, (size_t)nc, sizeof(couple_t), cmp_couple);
                char *u1 = (char *)calloc((size_t)r1.n + 1, 1), *u2 = (char *)calloc((size_t)r2.n + 1, 1);
                for (int64_t k = 0; k < nc; k++) {
                    if (u1[cp[k].i] || u2[cp[k].j]) continue;
                    u1[cp[k].i] = 1; u2[cp[k].j] = 1;
                    const mfk_aln *a = &A[r1.v[cp[k].i]], *b = &A[r2.v[cp[k].j]];
                    /* pair_mapped_both_ends 234-266 */
                    int f1 = 0, t1 = 0, f2 = 0, t2 = 0;
                    if (a->is_reverse && (int64_t)a->contig_len - a->ref_end <= L) t1 = 1; else if (!a->is_reverse && a->ref_start <= L) f1 = 1;
                    if (b->is_reverse && (int64_t)b->contig_len - b->ref_end <= L) t2 = 1; else if (!b->is_reverse && b->ref_start <= L) f2 = 1;
                    if (!((f1 && t2) || (t1 && f2))) continue;
                    if (a->is_reverse && !b->is_reverse) { iv_push(&kf, r2.v[cp[k].j]); iv_push(&kr, r1.v[cp[k].i]); }
                    else if (!a->is_reverse && b->is_reverse) { iv_push(&kf, r1.v[cp[k].i]); iv_push(&kr, r2.v[cp[k].j]); }
                }
                free(u1); free(u2); free(cp);
            }
            free(r1.v); free(r2.v);
        }
        /* keep_best_alignment_score 303-318 */
        int64_t best = 0;
        for (int64_t k = 0; k < kf.n; k++) { int64_t sc = (int64_t)A[kf.v[k]].as + A[kr.v[k]].as; if (sc > best) best = sc; }
        for (int64_t k = 0; k < kf.n; k++)
            if ((int64_t)A[kf.v[k]].as + A[kr.v[k]].as == best) { C->isq_f[C->n_isq] = kf.v[k]; C->isq_r[C->n_isq] = kr.v[k]; C->n_isq++; }
        free(kf.v); free(kr.v); free(contigs.v);
    }
    free(keep);
    *n_pairs = C->n_isq;
    return 0;
}

/* external inverted-repeat results for the next mfko_pairflanks call (see mfk_pairflanks_set_ir) */
int mfko_pairflanks_set_ir(void *c, int64_t n, const int32_t *index_5p, const int32_t *index_3p, const int32_t *ir) {
    octx *C = (octx *)c;
    if (!C || n < 0) return 1;
    free(C->ir_ov); C->ir_ov = NULL; C->n_ir_ov = 0;
    if (n == 0) return 0;
    C->ir_ov = (int32_t *)malloc((size_t)n * 7 * sizeof(int32_t));
    for (int64_t k = 0; k < n; k++) { C->ir_ov[7 * k] = index_5p[k]; C->ir_ov[7 * k + 1] = index_3p[k]; memcpy(C->ir_ov + 7 * k + 2, ir + 5 * k, 5 * sizeof(int32_t)); }
    C->n_ir_ov = n;
    return 0;
}

/* findflanks then pairflanks on its table (see mfk_flankpairs): the oracle simply runs the two restatements in a row */
int mfko_flankpairs(void *c, const mfk_reads *R, const mfk_params *P, const int32_t *contig_rank, const uint8_t *genome,
                    const int64_t *genome_off, int64_t *n_sites, int64_t *n_flanks, int64_t *seq_bytes, int64_t *n_pairs, int64_t *str_bytes) {
    octx *C = (octx *)c;
    int rc = mfko_findflanks(c, R, P, MFK_STAGE_ALL, n_sites, n_flanks, seq_bytes);
    if (rc) return rc;
    return mfko_pairflanks(c, R, P, C->flanks, C->seqs, C->n_flanks, contig_rank, genome, genome_off, n_pairs, str_bytes);
}

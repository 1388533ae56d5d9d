// mfk_consensus.hip -- consensus flank extension (replaces SoftclipConsensus, findflanks.py:862-1104, and the Trie of
// flanktrie.py) without building a trie.  With one "word" (clipped sequence + qualities) per query name:
//
//   trie node   = distinct non-empty prefix; counter(p) = #words with prefix p; qual(p) = sum of their qualities at
//                 that depth; word_count(p) = #words equal to p                       (Trie.add, flanktrie.py:31-59)
//   leaf        = word that is not a proper prefix of another word (first occurrence)  (traverse_seqs 157-170)
//   leaf order  = DFS with children in insertion order = leaves sorted by the sequence t(a[:1]), t(a[:2]), ... where
//                 t(p) = index of the first word with prefix p
//   clusters    = leaf pairs with levenshtein(prefixes of the shorter length) d, 4*d <= m  (== 1 - d/m >= 0.75 in
//                 float64, tests/test_oracle_golden.py) -> connected components; unmatched leaves are singletons
//   node sums   = the trie node of prefix p is named by (t(p), len(p)-1), i.e. by a character position of its first word:
//                 every word adds its quality / a counter / an end-of-word counter at the nodes along its path
//   consensus   = column vote over the sub-trie of the cluster's leaves: at depth d the voters are the distinct nodes of
//                 depth d on the paths to those leaves, each with its quality sum and counter; stop when fewer than mcc
//                 voters; per-base sum of qualities, unique maximum wins, tie -> 'N' (make_consensus_word 306-342)
//   count       = #words that end on a node of that sub-trie                            (make_subtrie 89-118)
//
// All pairwise work (LCP, bit-parallel Myers edit distance) is spread over the whole GPU; words live in an 8-byte
// aligned byte pool so LCPs are 64-bit XOR + count-trailing-zeros.
#include "mfk_host.h"
#include <type_traits>
#include <stdlib.h>

#define MAX_WORD_LEN 512
#define ERR_READ_NUMBER 1u
#define ERR_NO_SEQUENCE 2u
#define ERR_WORD_TOO_LONG 4u

struct CMeta {          // small device struct the host reads back once
    uint32_t err, max_len, max_R, n_big;   // n_big: jobs k_job_small left to the device-wide kernels
    uint32_t n_retry, n_small, pad[2];     // n_retry: jobs that did not fit an LDS rung of k_job_small; n_small: jobs of 1..SJ_MAX_R words
};

// per trie node: quality sum (low 40 bits) and #words through the node (high 24 bits) in ONE 64-bit word, so that a node
// costs one global atomic instead of two (the flush of k_node_sums is bound by the L2's atomic rate); #words ending on it
struct NodeSum { unsigned long long qc; uint32_t e; };   // register view; in memory two arrays (agg_qc, agg_e): 12 bytes per node, 8-byte aligned atomics
#define NS_C_SHIFT 40
#define NS_Q_MASK ((1ull << NS_C_SHIFT) - 1ull)

struct WordSrc {        // where a word's characters come from: pool offsets and clip lengths of its one or two reads
    uint64_t o1, o2;
    uint16_t l1, l2, q1len, q2len;
};

struct Words {
    uint32_t n;                       // number of words (active jobs only)
    WordSrc *src;
    int32_t *first, *second;          // reads
    uint32_t *nextra;                 // reads beyond the first with that name
    uint32_t *job;                    // job = 2*k + side
    uint32_t *len;                    // word length
    uint32_t *plen;                   // padded to a multiple of 16
    uint64_t *off;                    // byte offset into ch / q (multiple of 16)
    uint8_t *ch, *q;                  // characters (4-bit codes, one per byte) and qualities
    uint2 *sig;                       // device-wide path: LCP of a word with each of its job's N_LONGEST longest words (4 x u16), see d_sig_lcp
    uint32_t *npos;                   // per character d of a leaf a: its trie node = position of character d of word t(a[:d+1])
    uint32_t *cover;                  // per word: a leaf (local leaf index) the word is a prefix of
    unsigned long long *agg_qc;       // per character position = trie node: quality sum | #words << 40
    uint32_t *agg_e;                  // #words ending on the node
    uint32_t *share;                  // per leaf slot: longest prefix shared with an earlier leaf of its cluster
    uint8_t *is_leaf;
    uint32_t *leaf_word;              // slot w0+k -> local word index of the k-th leaf (word order)
    uint32_t *leaf_rank;              // DFS rank of the leaf in slot
    uint32_t *label;                  // cluster label (local leaf index of the representative)
    uint8_t *has_match;
    int32_t *cl_count, *cl_len;       // per representative slot
    uint32_t *cl_key;                 // (is_component << 31) | min DFS rank
    // device-side work lists (order arbitrary): the kernels that work per leaf / per cluster run a fixed grid over them
    uint32_t *leaf_list, *rep_list;   // slots of the leaves / of the cluster representatives
    uint32_t *out_list;               // slots of the leaves outside their job's giant component after the pivot pass
    uint32_t *my_list;                // pivot-pass work items of k_leaf_myers: job << 8 | pivot << 4 | partner chunk
    uint32_t *n_list;                 // [1] = representatives, [2] = outsiders; [4] = leaves and [5] = pivot-pass work items (one 64-bit counter)
};

struct Jobs {
    uint32_t n;                       // 2 * n_act slots
    uint32_t *w0, *R, *nl;            // word range, leaf count
    uint32_t *Rbig;                   // k_job_small: 0 = job done there, R = left to the device-wide kernels, R | 2^31 = try again with more LDS
    uint32_t *adj_words;              // adjacency words of the job: R * ceil(R/32)
    uint64_t *adj_off;
    uint32_t *adj;
    uint32_t *best;                   // best representative slot or 0xFFFFFFFF
    uint32_t *giant;                  // label of the largest component found by the pivot pass
    uint32_t *longest;                // N_LONGEST longest words of the job (local indices)
};

// longest common prefix, 32 bytes per step (words start on 16-byte boundaries and the pool has 64 bytes of slack, so
// reading past a word is harmless: the result is clamped to the shorter length)
__device__ __forceinline__ int d_lcp(const uint8_t *a, int la, const uint8_t *b, int lb) {
    const int m = la < lb ? la : lb;
    for (int o = 0; o < m; o += 32) {
        const uint4 a0 = *reinterpret_cast<const uint4 *>(a + o), a1 = *reinterpret_cast<const uint4 *>(a + o + 16);
        const uint4 b0 = *reinterpret_cast<const uint4 *>(b + o), b1 = *reinterpret_cast<const uint4 *>(b + o + 16);
        const uint32_t x[8] = {a0.x ^ b0.x, a0.y ^ b0.y, a0.z ^ b0.z, a0.w ^ b0.w, a1.x ^ b1.x, a1.y ^ b1.y, a1.z ^ b1.z, a1.w ^ b1.w};
        if (x[0] | x[1] | x[2] | x[3] | x[4] | x[5] | x[6] | x[7]) {
            int k = m;
#pragma unroll
            for (int i = 7; i >= 0; i--) if (x[i]) k = o + 4 * i + (__builtin_ctz(x[i]) >> 3);
            return k < m ? k : m;
        }
    }
    return m;
}

// Hamming distance of a[0:m) and b[0:m), 16 bytes per step, given up once it exceeds `limit` (returns limit + 1 or more).
// The edit distance never exceeds it, so a pair with 4 * hamming <= m is a match without running the bit-vector recurrence
// (words of one site-side are copies of one sequence with substitution errors: nearly every pair is settled here).
__device__ __forceinline__ int d_hamming16(const uint8_t *a, const uint8_t *b, int m, int limit) {
    int hd = 0;
    for (int o = 0; o < m && hd <= limit; o += 16) {
        const uint4 x = *reinterpret_cast<const uint4 *>(a + o), y = *reinterpret_cast<const uint4 *>(b + o);
        uint32_t d[4] = {x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w};
        const int left = m - o;   // bytes of this step that count
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t w = d[i];
            w |= w >> 4; w |= w >> 2; w |= w >> 1; w &= 0x01010101u;   // one bit per differing byte
            const int nb = left - 4 * i;
            if (nb < 4) w &= nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u);
            hd += __popc(w);
        }
    }
    return hd;
}

#define SJ_MAX_R 256
// ---- words ------------------------------------------------------------------------------------------------------
// words of a job = the first reads of the distinct query names of its site-side, i.e. the events with info bit 2: their
// number is the site's distinct-name count (k_site_accumulate), so the word ranges come from a scan over the job slots
// and no flag array / scan over all events is needed
__global__ void k_job_meta(ConsensusIO io, Jobs J, CMeta *meta) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= J.n) return;
    uint32_t k = j >> 1, side = j & 1;
    uint32_t st = io.act_st[k];
    uint32_t R = 0;
    if ((st & 4) && (st & (1u << side))) R = io.act_cnt2[j];
    J.R[j] = R; J.nl[j] = 0; J.best[j] = 0xFFFFFFFFu;
    J.adj_words[j] = R * ((R + 31) >> 5);
    if (R) atomicMax(&meta->max_R, R);
    if (R && R <= SJ_MAX_R) atomicAdd(&meta->n_small, 1u);
}

// k_job_meta and the two scans over the job slots (first word, first adjacency word) in ONE launch of one workgroup: up to
// JM_ONE_MAX job slots (a few thousand sites) three launches cost more than the work
#define JM_ONE_MAX 16384
__global__ __launch_bounds__(1024) void k_job_meta_scan(ConsensusIO io, Jobs J, CMeta *meta) {
    __shared__ uint32_t sh[16];
    __shared__ unsigned long long sh64[16];
    uint32_t run32 = 0, maxR = 0, nsmall = 0;
    unsigned long long run64 = 0;
    for (uint32_t base = 0; base < J.n; base += blockDim.x) {   // block-uniform
        const uint32_t j = base + threadIdx.x;
        uint32_t R = 0;
        if (j < J.n) {
            const uint32_t k = j >> 1, side = j & 1, st = io.act_st[k];
            if ((st & 4) && (st & (1u << side))) R = io.act_cnt2[j];
        }
        const uint32_t aw = R * ((R + 31) >> 5);
        uint32_t tot32; unsigned long long tot64;
        const uint32_t o32 = block_excl_scan(R, sh, &tot32);
        const unsigned long long o64 = block_excl_scan64((unsigned long long)aw, sh64, &tot64);
        if (j < J.n) {
            J.R[j] = R; J.nl[j] = 0; J.best[j] = 0xFFFFFFFFu; J.adj_words[j] = aw;
            J.w0[j] = run32 + o32; J.adj_off[j] = run64 + o64;
            if (R > maxR) maxR = R;
            if (R && R <= SJ_MAX_R) nsmall++;
        }
        run32 += tot32; run64 += tot64;
        __syncthreads();
    }
    if (maxR) atomicMax(&meta->max_R, maxR);
    if (nsmall) atomicAdd(&meta->n_small, nsmall);
    if (threadIdx.x == 0) { J.w0[J.n] = run32; J.adj_off[J.n] = run64; }
}

// first / second read of every word (combine_softclip_seq_quals_*: one or two reads per name, findflanks.py:932-1027);
// one block per active site.  Phase 1 numbers the word-starting events of each kept side (block scan over the side's
// events, offset by the job's first word) into wid[]; phase 2 fills the word table and attaches the second reads.
__global__ __launch_bounds__(256) void k_word_reads(DevReads R, DevParams P, ConsensusIO io, const uint32_t *w0, uint32_t *wid, Words W) {
    __shared__ uint32_t sh[8];
    const uint32_t k = blockIdx.x;
    if (k >= io.n_act) return;
    const uint32_t st = io.act_st[k];
    if (!(st & 4)) return;
    const uint32_t s = io.act_site[k], ev_mid = io.site_ev_mid[s];
    const uint32_t ev_b = io.site_ev_begin[s], ev_e = io.site_ev_end[s];
    const uint32_t mid = ev_mid == 0xFFFFFFFFu ? ev_e : ev_mid;   // no 3p event at the site
    for (int side = 0; side < 2; side++) {
        if (!(st & (1u << side))) continue;   // block-uniform
        const uint32_t b = side == 0 ? ev_b : mid, e = side == 0 ? mid : ev_e;
        uint32_t running = w0[2 * k + side];
        for (uint32_t t0 = b; t0 < e; t0 += blockDim.x) {
            const uint32_t ev = t0 + threadIdx.x;
            const uint32_t f = (ev < e && (io.ev_info[ev] & 2)) ? 1u : 0u;
            uint32_t tot;
            const uint32_t off = block_excl_scan(f, sh, &tot);
            if (f) wid[ev] = running + off;
            running += tot;
            __syncthreads();
        }
    }
    __syncthreads();
    for (uint32_t e = ev_b + threadIdx.x; e < ev_e; e += blockDim.x) {
        const int side = e >= mid ? 1 : 0;
        if (!(st & (1u << side))) continue;
        const uint8_t in = io.ev_info[e];
        const uint64_t key = io.ev_keys[e];
        const int64_t r = (int64_t)(key & 0x7fffffffu);
        if (in & 2) {
            const uint32_t w = wid[e];
            W.first[w] = (int32_t)r;
            W.job[w] = 2u * k + (uint32_t)side;
            continue;
        }
        // not the first read of its name here: find the first one (earliest read of the chain with the same event)
        const uint32_t gpos = (uint32_t)(key >> 32);
        const uint16_t contig = R.contig_id[r];
        const uint32_t base = R.contig_base[contig];
        int64_t root = -1, prev = r;
        for (int64_t q = R.link[r]; q >= 0 && q < prev; prev = q, q = R.link[q]) {
            int32_t site;
            if (R.contig_id[q] == contig && d_clip_event(R, P, q, side, &site) && base + (uint32_t)(site + 1) == gpos) root = q;
        }
        if (root < 0) continue;   // cannot happen: info bit 2 is clear exactly when such a read exists
        // its event: binary search by key in this site-side's range
        uint32_t lo = side == 0 ? ev_b : mid;
        uint32_t hi = side == 0 ? mid : ev_e;
        const uint32_t hi0 = hi;
        const uint64_t want = (key & ~(uint64_t)0x7fffffffu) | (uint64_t)root;
        while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (io.ev_keys[m] < want) lo = m + 1; else hi = m; }
        if (lo >= hi0 || io.ev_keys[lo] != want || !(io.ev_info[lo] & 2)) continue;
        const uint32_t w = wid[lo];
        atomicAdd(&W.nextra[w], 1u);
        W.second[w] = (int32_t)r;
    }
}

// one thread per word: length, and where its characters are (a binary search of the pool per read: thousands of
// independent searches in flight instead of one latency chain per wave in k_word_fill)
__global__ __launch_bounds__(256) void k_word_len(DevReads R, Words W, const uint32_t *n_words, CMeta *meta) {
    __shared__ uint32_t s_max;   // longest word of the workgroup: one atomic on meta->max_len per workgroup, not one per wave
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < *n_words) {
        int side = (int)(W.job[w] & 1);
        int64_t r1 = W.first[w], r2 = W.second[w];
        if (W.nextra[w] > 1) atomicOr(&meta->err, ERR_READ_NUMBER);      // findflanks.py:976-978 / 1025-1027
        // the first read's columns together (one round trip): flags, the CIGAR-end length of this side, query length, pool offset
        const bool all_pooled = R.n_pool == R.n;
        uint32_t x_f = R.flags[r1], x_op = side == 0 ? R.last_op_len[r1] : R.first_op_len[r1], x_ql = R.qlen[r1];
        uint64_t x_po = all_pooled ? R.pool_off[r1] : 0;
        asm volatile("" : "+v"(x_f), "+v"(x_op), "+v"(x_ql));
        const uint8_t f1 = (uint8_t)x_f;
        const int k1 = side == 0 ? d_rkind(f1) : d_lkind(f1);
        int l1 = k1 == 1 ? (int)x_op : (k1 == 2 ? 1 : 0), l2 = 0;      // d_rlen / d_llen
        const int64_t p1 = all_pooled ? r1 : d_pool_index(R, r1);
        int64_t p2 = -1;
        WordSrc src;
        src.o1 = all_pooled ? x_po : (p1 >= 0 ? R.pool_off[p1] : 0); src.o2 = 0;
        src.q1len = (uint16_t)x_ql; src.q2len = 0;
        if (r2 >= 0) {
            uint8_t f2 = R.flags[r2];
            l2 = side == 0 ? d_rlen(R, r2, f2) : d_llen(R, r2, f2);
            p2 = d_pool_index(R, r2);
            src.o2 = p2 >= 0 ? R.pool_off[p2] : 0;
            src.q2len = R.qlen[r2];
        }
        int len = l1 > l2 ? l1 : l2;
        if (len > MAX_WORD_LEN) { atomicOr(&meta->err, ERR_WORD_TOO_LONG); len = MAX_WORD_LEN; }
        if (l1 > MAX_WORD_LEN) l1 = MAX_WORD_LEN;
        if (l2 > MAX_WORD_LEN) l2 = MAX_WORD_LEN;
        if (p1 < 0 || (r2 >= 0 && p2 < 0) || l1 > (int)src.q1len || l2 > (int)src.q2len) { atomicOr(&meta->err, ERR_NO_SEQUENCE); l1 = 0; l2 = 0; }
        src.l1 = (uint16_t)l1; src.l2 = (uint16_t)l2;
        W.src[w] = src;
        W.len[w] = (uint32_t)len;
        W.plen[w] = (uint32_t)((len + 15) & ~15);
        atomicMax(&s_max, (uint32_t)len);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_max) atomicMax(&meta->max_len, s_max);
}

// clipped base / quality p of read r in trie orientation: 5p = right clip as is, 3p = left clip reversed (Q14)
__device__ __forceinline__ void d_clip_char(const DevReads &R, uint64_t pool_off, int qlen, int cliplen, int side, int p, uint8_t *c, uint8_t *q) {
    uint64_t k = side == 0 ? pool_off + (uint64_t)(qlen - cliplen + p) : pool_off + (uint64_t)(cliplen - 1 - p);
    *c = d_pool_base(R, k);
    *q = R.pool_quals[k];
}

// mate merge by quality (combine_softclip_seq_quals_*, findflanks.py:932-1027): character p of a word, 0xFF / 0 padding
__device__ __forceinline__ void d_word_char(const DevReads &R, const WordSrc &src, int side, int len, int p, uint8_t *c_out, uint8_t *q_out) {
    uint8_t c = 0xFF, q = 0;
    if (p < len) {
        const int l1 = src.l1, l2 = src.l2;
        uint8_t c1 = 0, qa = 0, c2 = 0, qb = 0;
        if (p < l1) d_clip_char(R, src.o1, src.q1len, l1, side, p, &c1, &qa);
        if (p < l2) d_clip_char(R, src.o2, src.q2len, l2, side, p, &c2, &qb);
        if (p < l1 && p < l2) { if (qa >= qb) { c = c1; q = qa; } else { c = c2; q = qb; } }   // read1 = first in BAM order (Q4)
        else if (p < l1) { c = c1; q = qa; }
        else if (p < l2) { c = c2; q = qb; }
    }
    *c_out = c; *q_out = q;
}

// The 16 bytes at base[s .. s + 16) of an array of which base[0 .. vlen) is known to be readable (s may be negative or run
// past vlen: those bytes of the result are unspecified).  ONE 16-byte load, moved inside the aligned 16-byte blocks that hold
// readable bytes (an aligned block never leaves the allocation its readable byte is in; the load itself need not be
// aligned: gfx950 under HSA runs with unaligned access enabled), then a 128-bit shift puts the window in place.  (A select
// network over two aligned blocks was turned into an indexed array in scratch memory by the compiler: three times the
// kernel's payload in scratch stores.)
__device__ __forceinline__ uint4 d_window16(const uint8_t *base, int64_t s, int64_t vlen) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(base);
    const int64_t lo16 = -(int64_t)(a & 15), hi16 = vlen + (int64_t)((16 - ((a + (uintptr_t)vlen) & 15)) & 15);
    int64_t ld = s < lo16 ? lo16 : s;
    if (ld > hi16 - 16) ld = hi16 - 16;
    const int delta = (int)(s - ld);   // -15 .. 15 whenever the window holds a readable byte
    uint4 L;
    __builtin_memcpy(&L, base + ld, 16);
    unsigned __int128 x = ((unsigned __int128)(((uint64_t)L.w << 32) | L.z) << 64) | (((uint64_t)L.y << 32) | L.x);
    const int n = 8 * (delta < 0 ? -delta : delta);
    x = delta >= 0 ? x >> (n & 127) : x << (n & 127);
    const uint64_t xl = (uint64_t)x, xh = (uint64_t)(x >> 64);
    return make_uint4((uint32_t)xl, (uint32_t)(xl >> 32), (uint32_t)xh, (uint32_t)(xh >> 32));
}
// four packed bytes (two base codes each, the first in the high nibble) -> eight characters, one code per byte
__device__ __forceinline__ void d_unpack8(uint32_t pk, uint32_t *c0, uint32_t *c1) {
    const uint32_t l = pk & 0xFFFFu, h = pk >> 16;
    const uint32_t x = (l | (l << 8)) & 0x00FF00FFu, y = (h | (h << 8)) & 0x00FF00FFu;
    *c0 = ((x >> 4) & 0x000F000Fu) | ((x & 0x000F000Fu) << 8);
    *c1 = ((y >> 4) & 0x000F000Fu) | ((y & 0x000F000Fu) << 8);
}
__device__ __forceinline__ uint32_t d_bswap(uint32_t x) { return __builtin_amdgcn_perm(x, x, 0x00010203u); }

// Characters [p0, p0 + 16) of a word and their qualities, straight from the pool: what one lane of k_word_fill writes, and what
// k_job_small loads into LDS when the fill is fused into it.  The source of a chunk is 16 consecutive quality bytes and 16
// consecutive 4-bit base codes of ONE read (the usual word: one read clipped here; 3p words are read backwards), fetched with one
// clamped 16-byte load each.  A word merged from both mates of a pair (both clipped at this site, findflanks.py:932-1027) goes
// character by character.  Padding behind the word: 0xFF / quality 0.
__device__ __forceinline__ void d_word_chunk(const DevReads &R, const WordSrc &src, int len, int side, int p0, uint32_t *c, uint32_t *q) {
    if (src.l2 == 0 && len > 0) {
        const int l1 = (int)src.l1;   // = len
        // window of the read, in read coordinates: 5p = right clip as is, 3p = left clip backwards (Q14)
        const int64_t s = side == 0 ? (int64_t)src.q1len - l1 + p0 : (int64_t)l1 - 16 - p0;
        const uint8_t *qbase = R.pool_quals + src.o1, *bbase = R.pool_bases + (src.o1 >> 1);   // pool offsets are even
        const uint4 qw = d_window16(qbase, s, (int64_t)src.q1len);
        const int64_t bs = s >> 1;    // floor: s is negative when a 3p word ends inside this chunk
        const uint4 bw = d_window16(bbase, bs, (int64_t)((src.q1len + 1) >> 1));
        uint32_t p0w = bw.x, p1w = bw.y;
        if (s & 1) {   // the window starts at the low nibble of its first byte: move the nibble stream up by one
            p0w = ((bw.x << 4) & 0xF0F0F0F0u) | ((__builtin_amdgcn_alignbyte(bw.y, bw.x, 1) >> 4) & 0x0F0F0F0Fu);
            p1w = ((bw.y << 4) & 0xF0F0F0F0u) | ((__builtin_amdgcn_alignbyte(bw.z, bw.y, 1) >> 4) & 0x0F0F0F0Fu);
        }
        d_unpack8(p0w, &c[0], &c[1]);
        d_unpack8(p1w, &c[2], &c[3]);
        q[0] = qw.x; q[1] = qw.y; q[2] = qw.z; q[3] = qw.w;
        if (side) {
            const uint32_t c0 = d_bswap(c[3]), c1 = d_bswap(c[2]), c2 = d_bswap(c[1]), c3 = d_bswap(c[0]);
            c[0] = c0; c[1] = c1; c[2] = c2; c[3] = c3;
            const uint32_t q0 = d_bswap(q[3]), q1 = d_bswap(q[2]), q2 = d_bswap(q[1]), q3 = d_bswap(q[0]);
            q[0] = q0; q[1] = q1; q[2] = q2; q[3] = q3;
        }
        const int n = len - p0;   // characters of the word in this chunk (>= 1); the rest is padding: 0xFF / quality 0
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int vb = n - 4 * i;
            const uint32_t m = vb >= 4 ? 0xFFFFFFFFu : (vb <= 0 ? 0u : (1u << (8 * vb)) - 1u);
            c[i] = (c[i] & m) | ~m; q[i] &= m;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            c[i] = 0; q[i] = 0;
            for (int k = 0; k < 4; k++) {
                uint8_t cc, qv;
                d_word_char(R, src, side, len, p0 + 4 * i + k, &cc, &qv);
                c[i] |= (uint32_t)cc << (8 * k); q[i] |= (uint32_t)qv << (8 * k);
            }
        }
    }
}

// A lane fills 16 characters (and their qualities) at a time; a wave takes WF_WORDS consecutive words.  The characters of
// consecutive words are neighbours in memory (every word padded to 16), so the wave walks the 16-byte chunks of its words'
// whole region, lane = chunk: every store instruction of the wave writes one contiguous kilobyte (chunks handed out word by
// word made pieces of 64 bytes at odd offsets, and the counters showed four times the bytes going to memory).  The word of
// a chunk is a four-step search over the wave's 17 offsets in LDS.  The source of a chunk is 16 consecutive quality bytes
// and 16 consecutive 4-bit base codes of ONE read (the usual word: one read clipped here; 3p words are read backwards),
// fetched with aligned 16-byte loads.  A word merged from both mates of a pair (both clipped at this site,
// findflanks.py:932-1027) goes character by character.
#define WF_WORDS 16
__global__ __launch_bounds__(256) void k_word_fill(DevReads R, Words W, const uint32_t *rbig) {   // rbig: only the words of jobs with rbig[job] != 0 (those k_job_small left over)
    __shared__ uint32_t s_off[4][WF_WORDS + 1], s_meta[4][WF_WORDS];   // s_meta: length | side << 16
    __shared__ WordSrc s_src[4][WF_WORDS];
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const uint32_t w0 = (blockIdx.x * 4 + (uint32_t)wave) * WF_WORDS;
    const uint32_t nw = w0 < W.n ? (W.n - w0 < WF_WORDS ? W.n - w0 : WF_WORDS) : 0u;
    if ((uint32_t)lane <= nw && nw) s_off[wave][lane] = (uint32_t)W.off[w0 + lane];   // off[] has an entry behind the last word
    if ((uint32_t)lane < nw) {
        const uint32_t jb = W.job[w0 + lane];
        s_src[wave][lane] = W.src[w0 + lane];
        s_meta[wave][lane] = W.len[w0 + lane] | ((jb & 1u) << 16) | ((rbig && rbig[jb] == 0) ? 1u << 17 : 0u);
    }
    __syncthreads();
    if (!nw) return;
    const uint32_t b_end = s_off[wave][nw];
    for (uint32_t b = s_off[wave][0] + 16u * (uint32_t)lane; b < b_end; b += 16u * 64u) {
        uint32_t u = 0;   // the word of this chunk: the last one that starts at or before it (empty words start where the next one does)
#pragma unroll
        for (uint32_t step = WF_WORDS / 2; step; step >>= 1) if (u + step < nw && s_off[wave][u + step] <= b) u += step;
        const WordSrc src = s_src[wave][u];
        const int len = (int)(s_meta[wave][u] & 0xFFFFu), side = (int)((s_meta[wave][u] >> 16) & 1u);
        if (s_meta[wave][u] >> 17) continue;   // its job is done: nobody reads these characters
        const int p0 = (int)(b - s_off[wave][u]);
        uint32_t c[4], q[4];
        d_word_chunk(R, src, len, side, p0, c, q);
        *reinterpret_cast<uint4 *>(W.ch + b) = make_uint4(c[0], c[1], c[2], c[3]);
        *reinterpret_cast<uint4 *>(W.q + b) = make_uint4(q[0], q[1], q[2], q[3]);
    }
}

// ---- leaves -------------------------------------------------------------------------------------------------------
// one block per job: its N_LONGEST longest words (ties: first).  A word without sequencing errors is a prefix of them,
// so testing these first settles most words in k_leaf_flags without the all-pairs scan.
__device__ __forceinline__ uint32_t d_wave_max(uint32_t v);   // (DPP wave maximum, defined with the small-job kernel)
#define N_LONGEST 4
// Signature of a word: p_k = LCP(word, L_k) for the job's N_LONGEST longest words L_k (0xFFFF where the job has fewer words).
// Two words that leave a reference at DIFFERENT positions share exactly the shorter stretch, and two that follow it equally far,
// one of them to its own end, share that stretch: nearly every LCP of a job (all but the pairs with the same sequencing error)
// follows from two 8-byte signatures and the lengths, without a look at the characters.  Returns -1 if it does not.
__device__ __forceinline__ int d_sig_lcp(uint2 sa, int la, uint2 sv, int lv) {
    const uint32_t a[2] = {sa.x, sa.y}, v[2] = {sv.x, sv.y};
#pragma unroll
    for (int k = 0; k < N_LONGEST; k++) {
        const int pa = (int)((a[k >> 1] >> (16 * (k & 1))) & 0xFFFFu), pv = (int)((v[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
        if (pa == 0xFFFF) continue;
        if (pa != pv) return pa < pv ? pa : pv;
        if (pa == la || pa == lv) return pa;
    }
    return -1;
}

// 1024 threads per job: a deep site-side has about a thousand words, i.e. one word per thread in the signature pass
#define JL_THREADS 1024
__global__ __launch_bounds__(JL_THREADS) void k_job_longest(Words W, Jobs J) {
    __shared__ unsigned long long best;
    __shared__ uint32_t s_j[N_LONGEST], s_len[N_LONGEST], s_off[N_LONGEST], s_w[JL_THREADS / 64];
    __shared__ __align__(16) uint8_t s_ref[N_LONGEST][MAX_WORD_LEN + 16];   // the references' characters
    uint32_t job = blockIdx.x;
    if (job >= J.n) return;
    uint32_t w0 = J.w0[job], R = J.R[job];
    // keys (length, first index wins) in 32 bits: 10 bits of length above 22 bits of reversed index; a thread reads the lengths of
    // its words ONCE (up to KL per thread in registers; beyond that from memory again in every round), the block maximum
    // goes through DPP wave maxima and four LDS words: no 64-bit shuffles, no atomics
    constexpr int KL = 4;
    const bool cached = R <= (uint32_t)JL_THREADS * KL && R < (1u << 22);
    uint32_t keys[KL];
#pragma unroll
    for (int u = 0; u < KL; u++) { const uint32_t i = threadIdx.x + (uint32_t)JL_THREADS * u; keys[u] = cached && i < R ? (W.len[w0 + i] << 22) | (0x3FFFFFu - i) : 0u; }
    unsigned long long prev = ~0ull;
    for (int k = 0; k < N_LONGEST; k++) {
        if (cached) {   // block-uniform
            uint32_t mine = 0;
            const uint32_t below = prev == ~0ull ? 0xFFFFFFFFu : (uint32_t)prev;
#pragma unroll
            for (int u = 0; u < KL; u++) if (keys[u] < below && keys[u] > mine) mine = keys[u];
            mine = d_wave_max(mine);
            if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = mine;
            __syncthreads();
            uint32_t m = s_w[0];
            for (int w = 1; w < JL_THREADS / 64; w++) if (s_w[w] > m) m = s_w[w];
            prev = m;
            if (threadIdx.x == 0) {
                const uint32_t j = m ? 0x3FFFFFu - (m & 0x3FFFFFu) : 0xFFFFFFFFu;
                J.longest[job * N_LONGEST + k] = j;
                s_j[k] = j; s_len[k] = m ? m >> 22 : 0u;   // (s_off: below, the four loads together)
            }
            __syncthreads();
            continue;   // (m == 0: nothing left, the remaining references stay empty)
        }
        if (threadIdx.x == 0) best = 0;
        __syncthreads();
        unsigned long long mine = 0;
        for (uint32_t i = threadIdx.x; i < R; i += blockDim.x) {
            unsigned long long key = ((unsigned long long)W.len[w0 + i] << 32) | (unsigned long long)(0xFFFFFFFFu - i);
            if (key < prev && key > mine) mine = key;
        }
        for (int o = 32; o > 0; o >>= 1) {
            unsigned long long other = ((unsigned long long)__shfl_xor((unsigned)(mine >> 32), o, 64) << 32) | (unsigned long long)__shfl_xor((unsigned)mine, o, 64);
            if (other > mine) mine = other;
        }
        if ((threadIdx.x & 63) == 0 && mine) atomicMax(&best, mine);
        __syncthreads();
        prev = best;
        if (threadIdx.x == 0) {
            const uint32_t j = best ? 0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFull) : 0xFFFFFFFFu;
            J.longest[job * N_LONGEST + k] = j;
            s_j[k] = j; s_len[k] = best ? (uint32_t)(best >> 32) : 0u; s_off[k] = best ? (uint32_t)W.off[w0 + j] : 0u;
        }
        __syncthreads();
    }
    if (cached && threadIdx.x < N_LONGEST) s_off[threadIdx.x] = s_j[threadIdx.x] != 0xFFFFFFFFu ? (uint32_t)W.off[w0 + s_j[threadIdx.x]] : 0u;
    __syncthreads();
    // signatures of the job's words (d_sig_lcp).  The references' characters go to LDS once; a word's characters are read once,
    // 16 bytes per step, and compared with all references still in the race (one memory round trip per step, not one per
    // reference and step)
    for (int k = 0; k < N_LONGEST; k++) {   // block-uniform
        if (s_j[k] == 0xFFFFFFFFu) continue;
        const uint4 *src = reinterpret_cast<const uint4 *>(W.ch + s_off[k]);
        for (uint32_t i = threadIdx.x; i < ((s_len[k] + 15u) >> 4); i += blockDim.x) reinterpret_cast<uint4 *>(s_ref[k])[i] = src[i];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < R; i += blockDim.x) {
        const int la = (int)W.len[w0 + i];
        const uint8_t *a = W.ch + W.off[w0 + i];
        uint32_t p[N_LONGEST]; int lim[N_LONGEST]; uint32_t open_ = 0;
#pragma unroll
        for (int k = 0; k < N_LONGEST; k++) {
            lim[k] = la < (int)s_len[k] ? la : (int)s_len[k];
            p[k] = s_j[k] == 0xFFFFFFFFu ? 0xFFFFu : (uint32_t)lim[k];   // (the word itself as a reference: its length)
            if (s_j[k] != 0xFFFFFFFFu && s_j[k] != i && lim[k] > 0) open_ |= 1u << k;
        }
        for (int o0 = 0; o0 < la && open_; o0 += 64) {
            uint4 xs[4];   // 64 bytes of the word per round trip (the pool has 64 bytes of slack behind the last word)
#pragma unroll
            for (int c = 0; c < 4; c++) xs[c] = *reinterpret_cast<const uint4 *>(a + o0 + 16 * c);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int o = o0 + 16 * c;
                const uint4 x = xs[c];
#pragma unroll
                for (int k = 0; k < N_LONGEST; k++) {
                    if (!(open_ & (1u << k))) continue;
                    if (o >= lim[k]) { open_ &= ~(1u << k); continue; }
                    const uint4 y = *reinterpret_cast<const uint4 *>(&s_ref[k][o]);
                    const uint32_t d[4] = {x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w};
                    if (d[0] | d[1] | d[2] | d[3]) {
                        int q = lim[k];
#pragma unroll
                        for (int z = 3; z >= 0; z--) if (d[z]) q = o + 4 * z + (__builtin_ctz(d[z]) >> 3);
                        p[k] = (uint32_t)(q < lim[k] ? q : lim[k]);
                        open_ &= ~(1u << k);
                    }
                }
            }
        }
        W.sig[w0 + i] = make_uint2(p[0] | (p[1] << 16), p[2] | (p[3] << 16));
    }
}

// is some other word an extension of a word (or an earlier copy)?  One wave per LF_WORDS consecutive words.  First
// every word is tested against the N_LONGEST longest words of its job from its signature alone (k_job_longest: p_k >= its
// length means it is a prefix of L_k), all LF_WORDS x N_LONGEST tests at once, one lane each; the words that settles
// (error-free reads) are done.  The others leave every L_k before their end (a sequencing error), and a word that extends
// them leaves every L_k at the same place: the scan of the job's words compares signatures (8 bytes per candidate,
// LF_BATCH x 64 candidates with their lengths in flight together) and compares characters only for the few that agree.
// lf = words per wave (1..LF_WORDS): the host takes fewer when there are too few words to fill the GPU with waves.
#define LF_WORDS 8
#define LF_BATCH 8
__global__ __launch_bounds__(256) void k_leaf_flags(Words W, Jobs J, int lf) {
    const int lane = threadIdx.x & 63;
    const uint32_t wbase = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (uint32_t)lf;
    if (wbase >= W.n) return;
    uint32_t m_w0 = 0, m_R = 0, m_off = 0;
    uint2 m_sig = make_uint2(0u, 0u);
    int m_la = 0;
    bool covered = false;
    {
        const uint32_t w = wbase + (uint32_t)lane / N_LONGEST;
        if (lane < lf * N_LONGEST && w < W.n) {
            const uint32_t job = W.job[w];
            const int k = lane % N_LONGEST;
            m_w0 = J.w0[job]; m_R = J.R[job]; m_off = (uint32_t)W.off[w]; m_la = (int)W.len[w];
            const uint32_t j = J.longest[job * N_LONGEST + k];
            if (j != 0xFFFFFFFFu) {   // (none for a job that k_job_small has done: its signatures were never written)
                m_sig = W.sig[w];
                const uint32_t v = m_w0 + j;
                const int lb = (int)W.len[v];
                const int p = (int)(((k & 2 ? m_sig.y : m_sig.x) >> (16 * (k & 1))) & 0xFFFFu);
                if (v != w && lb >= m_la && p >= m_la && (lb > m_la || v < w)) covered = true;
            }
        }
    }
    const unsigned long long cov = __ballot(covered);
    for (int i = 0; i < lf; i++) {   // wave-uniform
        const uint32_t w = wbase + (uint32_t)i;
        if (w >= W.n) break;
        if ((cov >> (i * N_LONGEST)) & ((1ull << N_LONGEST) - 1)) { if (lane == 0) W.is_leaf[w] = 0; continue; }
        const uint32_t w0 = __shfl(m_w0, i * N_LONGEST, 64), R = __shfl(m_R, i * N_LONGEST, 64);
        if (R == 0) continue;   // word of a job that k_job_small has done
        const int la = __shfl(m_la, i * N_LONGEST, 64);
        if (la <= 0) { if (lane == 0) W.is_leaf[w] = R > 1 ? 0 : 1; continue; }   // an empty word is a prefix of everything (cannot happen: clips have >= 1 base)
        const uint8_t *a = W.ch + __shfl(m_off, i * N_LONGEST, 64);
        const uint32_t swx = __shfl(m_sig.x, i * N_LONGEST, 64), swy = __shfl(m_sig.y, i * N_LONGEST, 64);
        unsigned long long any = 0;
        for (uint32_t c0 = 0; c0 < R; c0 += 64 * LF_BATCH) {   // wave-uniform
            int lens[LF_BATCH]; uint2 sg[LF_BATCH]; bool ok[LF_BATCH];
#pragma unroll
            for (int u = 0; u < LF_BATCH; u++) {
                const uint32_t j = c0 + (uint32_t)u * 64 + lane;
                ok[u] = j < R && w0 + j != w;
                lens[u] = ok[u] ? (int)W.len[w0 + j] : 0;
                sg[u] = ok[u] ? W.sig[w0 + j] : make_uint2(0u, 0u);
            }
            bool c2 = false;
#pragma unroll
            for (int u = 0; u < LF_BATCH; u++) {
                if (ok[u] && lens[u] >= la && sg[u].x == swx && sg[u].y == swy) {
                    const uint32_t v = w0 + c0 + (uint32_t)u * 64 + lane;
                    if (d_lcp(a, la, W.ch + W.off[v], lens[u]) >= la && (lens[u] > la || v < w)) c2 = true;
                }
            }
            any = __ballot(c2);
            if (any) break;
        }
        if (lane == 0) W.is_leaf[w] = any ? 0 : 1;
    }
}

#define N_PIVOTS 4   // pivot leaves of the first edit-distance pass (k_leaf_myers)

// one block per job: ordered list of its leaves
__global__ __launch_bounds__(256) void k_leaf_compact(Words W, Jobs J, uint32_t n_chunks) {
    __shared__ uint32_t sh[8];
    uint32_t job = blockIdx.x;
    if (job >= J.n) return;
    uint32_t w0 = J.w0[job], R = J.R[job];
    if (!R) return;
    uint32_t running = 0;
    for (uint32_t b = 0; b < R; b += blockDim.x) {
        uint32_t i = b + threadIdx.x;
        uint32_t f = (i < R && W.is_leaf[w0 + i]) ? 1u : 0u, tot;
        uint32_t off = block_excl_scan(f, sh, &tot);
        if (f) { W.leaf_word[w0 + running + off] = i; W.label[w0 + running + off] = running + off; }
        running += tot;
    }
    // pivot-pass work items of this job (k_leaf_myers mode 0): (pivot, chunk of 64 partners) pairs that have a partner.
    // A compact list lets that kernel give every item its own wave on its own SIMD; enumerating all (job slot, pivot,
    // chunk) triples leaves three of four waves without work and the busy ones bunched on the same SIMDs.
    // Room in the leaf list and in the item list is reserved with ONE 64-bit atomic per job (leaves low, items high):
    // tens of thousands of jobs on one counter serialise in the L2, so a second counter would double that time.
    __shared__ uint32_t s_base, s_ibase;
    const uint32_t ka = threadIdx.x >> 4, chunk = threadIdx.x & 15;
    const bool valid = W.my_list && threadIdx.x < 64 && ka < N_PIVOTS && chunk < n_chunks && running >= 2 && ka + 1 + 64 * chunk < running;
    const unsigned long long m = __ballot(valid);   // wave 0 holds all candidates
    if (threadIdx.x == 0) {
        J.nl[job] = running;
        const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long *>(W.n_list + 4), (unsigned long long)running | ((unsigned long long)__popcll(m) << 32));
        s_base = (uint32_t)old; s_ibase = (uint32_t)(old >> 32);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < running; i += blockDim.x) W.leaf_list[s_base + i] = w0 + i;
    if (valid) W.my_list[s_ibase + (uint32_t)__popcll(m & ((1ull << threadIdx.x) - 1ull))] = (job << 8) | (ka << 4) | chunk;
}

// one wave per leaf slot: t(a[:d+1]) = first word sharing d+1 characters with the leaf, for every depth d.
// A wave walks a strided sequence of leaves; what it needs to know about a leaf (slot -> job -> word range, leaf word ->
// offset, length) is a chain of dependent loads, so lane k resolves the chain of the wave's k-th next leaf and the
// results are broadcast leaf by leaf: 64 chains in flight instead of one.
__global__ __launch_bounds__(64) void k_leaf_first(Words W, Jobs J, uint16_t *lcpm) {
    __shared__ uint32_t tmin[MAX_WORD_LEN];
    const int lane = threadIdx.x;
    const uint32_t n_leaf = W.n_list[4];
    for (uint64_t it0 = blockIdx.x; it0 < n_leaf; it0 += (uint64_t)gridDim.x * 64) {
    uint32_t m_slot = 0, m_w0 = 0, m_R = 0, m_off = 0, m_la = 0, m_lw = 0;
    unsigned long long m_adj = 0;
    {
        const uint64_t it = it0 + (uint64_t)lane * gridDim.x;
        if (it < n_leaf) {
            m_slot = W.leaf_list[it];
            const uint32_t job = W.job[m_slot];
            m_w0 = J.w0[job]; m_R = J.R[job]; m_adj = J.adj_off[job];
            m_lw = W.leaf_word[m_slot];
            const uint32_t wa = m_w0 + m_lw;
            m_off = (uint32_t)W.off[wa]; m_la = W.len[wa];
        }
    }
    for (int kk = 0; kk < 64 && it0 + (uint64_t)kk * gridDim.x < n_leaf; kk++) {   // wave-uniform
    __syncthreads();
    const uint32_t slot = __shfl(m_slot, kk, 64), w0 = __shfl(m_w0, kk, 64), R = __shfl(m_R, kk, 64);
    const uint32_t off_a = __shfl(m_off, kk, 64);
    const int la = (int)__shfl(m_la, kk, 64);
    const unsigned long long adj_off = __shfl(m_adj, kk, 64);
    const uint32_t k = slot - w0, k_word = __shfl(m_lw, kk, 64);   // leaf number in its job, and the leaf's word
    const uint8_t *a = W.ch + off_a;
    for (int d = lane; d < la; d += 64) tmin[d] = 0xFFFFFFFFu;
    __syncthreads();
    uint16_t *mrow = lcpm + adj_off * 32 + (size_t)k * ((R + 31) >> 5) * 32;   // LCP(leaf, word j), reused by k_leaf_share
    const uint2 sa = W.sig[w0 + k_word];
    for (uint32_t j = lane; j < R; j += 64) {
        uint32_t v = w0 + j;
        const int lv = (int)W.len[v];
        int l = j == k_word ? la : d_sig_lcp(sa, la, W.sig[v], lv);   // nearly always settled without the characters
        if (l < 0) l = d_lcp(a, la, W.ch + W.off[v], lv);
        mrow[j] = (uint16_t)l;
        if (l > 0) atomicMin(&tmin[l - 1], j);
        if (l == lv) atomicMin(&W.cover[v], k);   // word v is a prefix of this leaf; fire-and-forget (a read-before-write check costs a load round trip)
    }
    __syncthreads();
    // suffix minimum over depth: lane owns `per` consecutive depths, then a wave-wide suffix scan of the lane minima
    const int per = (la + 63) >> 6, d0 = lane * per;
    uint32_t mine = 0xFFFFFFFFu;
    for (int d = d0 + per - 1; d >= d0; d--) if (d < la) { uint32_t v = tmin[d]; if (v < mine) mine = v; }
    uint32_t suf = mine;   // inclusive suffix minimum over lanes >= lane
    for (int o = 1; o < 64; o <<= 1) { uint32_t other = __shfl_down(suf, o, 64); if (lane + o < 64 && other < suf) suf = other; }
    uint32_t after = __shfl_down(suf, 1, 64);   // minimum over the lanes behind
    if (lane == 63) after = 0xFFFFFFFFu;
    uint32_t *np = W.npos + off_a;
    uint32_t m = after;
    for (int d = d0 + per - 1; d >= d0; d--) if (d < la) {
        uint32_t v = tmin[d]; if (v < m) m = v;
        np[d] = (uint32_t)(W.off[w0 + m] + (uint64_t)d);   // node of a[:d+1] = character d of the first word with that prefix
    }
    }
    }
}

// ---- leaf pairs: edit distance (Myers / Hyyro bit-vector, global alignment) + DFS order ------------------------------
// The match vectors Eq(c) of the pattern come from a table in LDS (one 64-bit limb per code and limb, limb-major so the
// 16 codes of a limb lie in 16 different banks): one LDS read per limb and step instead of rebuilding Eq from the four
// bit planes.  The reads of MY_BATCH steps are issued together.
#define MY_BATCH 4
template <int NL>
__device__ __forceinline__ int d_myers(const uint64_t *peq, const uint8_t *text, int m) {
    // pattern = leaf a (peq[limb * 16 + code]), text = b[0:m); returns D(a[0:m), b[0:m))
    uint64_t Pv[NL], Mv[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) { Pv[i] = ~0ull; Mv[i] = 0ull; }
    int score = m;
    const int tl = (m - 1) >> 6;
    const uint64_t tb = 1ull << ((m - 1) & 63);
    uint64_t next = *reinterpret_cast<const uint64_t *>(text);
    for (int o = 0; o < m; o += 8) {
        const uint64_t chunk = next;
        if (o + 8 < m) next = *reinterpret_cast<const uint64_t *>(text + o + 8);   // in flight while this chunk is processed
#pragma unroll
        for (int h = 0; h < 8; h += MY_BATCH) {
            if (o + h >= m) break;
            uint64_t Eqs[MY_BATCH][NL];
#pragma unroll
            for (int s = 0; s < MY_BATCH; s++) {
                const uint32_t c = (uint32_t)(chunk >> (8 * (h + s))) & 0xFF;
#pragma unroll
                for (int i = 0; i < NL; i++) Eqs[s][i] = c > 15 ? 0ull : peq[i * 16 + (c & 15)];
            }
            const int lim = m - o - h < MY_BATCH ? m - o - h : MY_BATCH;
#pragma unroll
            for (int s = 0; s < MY_BATCH; s++) {
                if (s >= lim) break;
                uint64_t carry_add = 0, carry_ph = 1, carry_mh = 0;   // carry_ph = 1: D[0][j] = j (global alignment)
#pragma unroll
                for (int i = 0; i < NL; i++) {
                    const uint64_t Eq = Eqs[s][i];
                    uint64_t pv = Pv[i], mv = Mv[i];
                    uint64_t Xv = Eq | mv;
                    uint64_t X = Eq & pv;
                    uint64_t sum = X + pv;
                    uint64_t c1 = sum < X ? 1ull : 0ull;
                    uint64_t sum2 = sum + carry_add;
                    uint64_t c2 = sum2 < sum ? 1ull : 0ull;
                    carry_add = c1 | c2;
                    uint64_t Xh = (sum2 ^ pv) | Eq;
                    uint64_t Ph = mv | ~(Xh | pv);
                    uint64_t Mh = pv & Xh;
                    if (i == tl) { if (Ph & tb) score++; else if (Mh & tb) score--; }
                    uint64_t ph_out = Ph >> 63, mh_out = Mh >> 63;
                    Ph = (Ph << 1) | carry_ph;
                    Mh = (Mh << 1) | carry_mh;
                    carry_ph = ph_out; carry_mh = mh_out;
                    Pv[i] = Mh | ~(Xv | Ph);
                    Mv[i] = Ph & Xv;
                }
            }
        }
    }
    return score;
}


// DFS order of the leaves (traverse_seqs, flanktrie.py:157-170): at the first differing depth the branch whose first word
// came earlier is visited first.  One block per leaf slot a, threads over the later leaves b; rank = #leaves before.
__global__ __launch_bounds__(256) void k_leaf_rank(Words W, Jobs J, const uint16_t *lcpm) {
    __shared__ uint32_t s_rank;
    const uint32_t n_leaf = W.n_list[4];
    for (uint32_t it = blockIdx.x; it < n_leaf; it += gridDim.x) {
        __syncthreads();
        if (threadIdx.x == 0) s_rank = 0;
        __syncthreads();
        const uint32_t slot = W.leaf_list[it];
        const uint32_t job = W.job[slot], w0 = J.w0[job], nl = J.nl[job], R = J.R[job];
        const uint32_t ka = slot - w0;
        const uint32_t wa = w0 + W.leaf_word[slot];
        const int la = (int)W.len[wa];
        const uint32_t *ta = W.npos + W.off[wa];
        const uint16_t *mrow = lcpm + J.adj_off[job] * 32 + (size_t)ka * ((R + 31) >> 5) * 32;   // LCP(leaf a, word j) from k_leaf_first
        uint32_t before = 0;   // leaves visited before a: every leaf counts its own predecessors, nobody writes to another leaf
        for (uint32_t kb = threadIdx.x; kb < nl; kb += blockDim.x) {
            if (kb == ka) continue;
            const uint32_t jb = W.leaf_word[w0 + kb], wb = w0 + jb;
            const int lb = (int)W.len[wb];
            const int m = la < lb ? la : lb;
            const int k = (int)mrow[jb];
            if (k < m && W.npos[W.off[wb] + k] < ta[k]) before++;   // same depth: positions order like first-word indices
        }
        for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
        if ((threadIdx.x & 63) == 0 && before) atomicAdd(&s_rank, before);
        __syncthreads();
        if (threadIdx.x == 0) W.leaf_rank[slot] = s_rank;
    }
}

// k_leaf_rank for jobs with few leaves (a 100-200x site has a few dozen): one wave per leaf, no LDS, no barrier, and the
// dependent loads that describe a leaf are resolved for 64 leaves at a time (lane k: the wave's k-th next leaf).
__global__ __launch_bounds__(256) void k_leaf_rank_wave(Words W, Jobs J, const uint16_t *lcpm) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    const uint32_t n_leaf = W.n_list[4];
    for (uint64_t it0 = wave; it0 < n_leaf; it0 += (uint64_t)n_waves * 64) {
        uint32_t m_slot = 0, m_w0 = 0, m_nl = 0, m_off = 0, m_la = 0, m_rw = 0;
        unsigned long long m_adj = 0;
        {
            const uint64_t it = it0 + (uint64_t)lane * n_waves;
            if (it < n_leaf) {
                m_slot = W.leaf_list[it];
                const uint32_t job = W.job[m_slot];
                m_w0 = J.w0[job]; m_nl = J.nl[job]; m_rw = (J.R[job] + 31) >> 5; m_adj = J.adj_off[job];
                const uint32_t wa = m_w0 + W.leaf_word[m_slot];
                m_off = (uint32_t)W.off[wa]; m_la = W.len[wa];
            }
        }
        for (int kk = 0; kk < 64 && it0 + (uint64_t)kk * n_waves < n_leaf; kk++) {   // wave-uniform
            const uint32_t slot = __shfl(m_slot, kk, 64), w0 = __shfl(m_w0, kk, 64), nl = __shfl(m_nl, kk, 64);
            const uint32_t off_a = __shfl(m_off, kk, 64);
            const int la = (int)__shfl(m_la, kk, 64);
            const uint32_t ka = slot - w0;
            const uint32_t *ta = W.npos + off_a;
            const uint16_t *mrow = lcpm + __shfl(m_adj, kk, 64) * 32 + (size_t)ka * __shfl(m_rw, kk, 64) * 32;   // LCP(leaf, word j) from k_leaf_first
            uint32_t before = 0;
            for (uint32_t kb = lane; kb < nl; kb += 64) {
                if (kb == ka) continue;
                const uint32_t jb = W.leaf_word[w0 + kb], wb = w0 + jb;
                const int lb = (int)W.len[wb];
                const uint64_t off_b = W.off[wb];
                const int m = la < lb ? la : lb;
                const int k = (int)mrow[jb];
                if (k < m && W.npos[off_b + k] < ta[k]) before++;
            }
            for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
            if (lane == 0) W.leaf_rank[slot] = before;
        }
    }
}

// Match-vector table of a leaf, built by the whole wave into its part of LDS: lane l looks at character 64*i + l, a
// ballot per code bit gives limb i of the four bit planes (wave-uniform), lane c < 16 combines them into Eq(c) of that limb.
__device__ __forceinline__ void d_peq_wave(const uint8_t *a, int la, int need, int lane, uint64_t *peq) {
    for (int i = 0; i < need; i++) {   // need is wave-uniform
        uint32_t c = 0;
        const int pos = i * 64 + lane;
        if (pos < la) c = a[pos];
        uint64_t eq = ~0ull;
#pragma unroll
        for (int p = 0; p < 4; p++) { const uint64_t pl = __ballot((c >> p) & 1); eq &= ((lane >> p) & 1) ? pl : ~pl; }
        // positions beyond the leaf carry code 0 in the planes: they sit above bit m-1 of every test, where they cannot
        // reach the score bit (carries only travel upwards)
        if (lane < 16) peq[i * 16 + lane] = eq;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

__device__ __forceinline__ int d_myers_any(const uint64_t *peq, int need, const uint8_t *text, int m) {
    switch (need) {
        case 1: return d_myers<1>(peq, text, m);
        case 2: return d_myers<2>(peq, text, m);
        case 3: return d_myers<3>(peq, text, m);
        case 4: return d_myers<4>(peq, text, m);
        default: return d_myers<8>(peq, text, m);
    }
}

// Edit-distance tests between leaves (get_sequence_clusters, findflanks.py:1041-1053).  Only connectivity and "has any
// match" are needed downstream, so pairs inside an already known component are never tested:
//   mode 0: the first N_PIVOTS leaves of the job against every later leaf;
//   mode 1: every leaf outside the giant component (W.out_list, written by k_components) against every leaf of a
//           different component.
// Every pair across two components of the mode-0 graph is tested, so the final components are those of the full graph.
// The work item is (pattern leaf a, chunk of partners) and belongs to ONE WAVE: lane l of chunk c takes the partners
// c*64 + l, + 64*n_chunks, ...  A site at 200x has a few dozen leaves and one chunk, a 1000x site hundreds of leaves in
// up to 16 chunks: either way the GPU is filled with waves that each run one bit-vector recurrence, and nothing waits
// at a barrier.
__global__ __launch_bounds__(256) void k_leaf_myers(Words W, Jobs J, int mode, uint32_t n_chunks, unsigned long long *work) {
    __shared__ uint64_t s_peq[4][8 * 16];   // per wave: Eq[limb][code] of its current pattern leaf
    const int lane = threadIdx.x & 63;
    uint32_t limb_steps = 0;   // columns x limbs of the bit-vector recurrences this lane ran (bench.py: work counter of the VALU roofline)
    uint64_t *peq = s_peq[threadIdx.x >> 6];
    const uint32_t wpb = blockDim.x >> 6;   // waves per block: 1 in the pivot pass (every item on its own SIMD), 4 otherwise
    const uint32_t wave = blockIdx.x * wpb + (threadIdx.x >> 6), n_waves = gridDim.x * wpb;
    const bool listed = mode == 0 && W.my_list != nullptr;
    const uint32_t n_items = listed ? W.n_list[5] : (mode == 0 ? J.n * N_PIVOTS : W.n_list[2]) * n_chunks;
    for (uint32_t it = wave; it < n_items; it += n_waves) {   // wave-uniform
        uint32_t job, ka, slot, chunk;
        if (listed) { const uint32_t item = W.my_list[it]; job = item >> 8; ka = (item >> 4) & 15; chunk = item & 15; slot = J.w0[job] + ka; }
        else if (mode == 0) { chunk = it / (J.n * N_PIVOTS); const uint32_t r = it % (J.n * N_PIVOTS); job = r / N_PIVOTS; ka = r % N_PIVOTS; slot = J.w0[job] + ka; }
        else { chunk = it % n_chunks; slot = W.out_list[it / n_chunks]; job = W.job[slot]; ka = slot - J.w0[job]; }
        const uint32_t w0 = J.w0[job], R = J.R[job], nl = J.nl[job];
        if (ka >= nl || nl < 2) continue;
        const uint32_t first = (mode == 0 ? ka + 1 : 0) + chunk * 64;
        if (first >= nl) continue;
        uint32_t my_label = 0, giant = 0;
        if (mode == 1) { my_label = W.label[slot]; giant = J.giant[job]; }
        const uint32_t wa = w0 + W.leaf_word[slot];
        const uint8_t *a = W.ch + W.off[wa];
        const int la = (int)W.len[wa];
        const int need = (la + 63) >> 6;
        if (need > 8) continue;   // cannot happen: words are capped at MAX_WORD_LEN
        d_peq_wave(a, la, need, lane, peq);
        const uint32_t rw = (R + 31) >> 5;
        uint32_t *adj = J.adj + J.adj_off[job];
        for (uint32_t kb = first + lane; kb < nl; kb += 64 * n_chunks) {
            if (mode == 1) {
                if (kb == ka) continue;
                const uint32_t lb_label = W.label[w0 + kb];
                if (lb_label == my_label) continue;
                if (lb_label != giant && kb < ka) continue;       // two outsiders: the lower slot's wave takes the pair
            }
            const uint32_t wb = w0 + W.leaf_word[w0 + kb];
            const int lb = (int)W.len[wb];
            const int m = la < lb ? la : lb;
            const uint8_t *tb = W.ch + W.off[wb];
            int d = d_hamming16(a, tb, m, m >> 2);
            if (4 * d > m) { d = d_myers_any(peq, need, tb, m); limb_steps += (uint32_t)(m * (need > 4 ? 8 : need)); }
            if (4 * d <= m) {   // similarity = 1 - d/m >= 0.75 (findflanks.py:1044-1048, quirk Q11)
                atomicOr(&adj[(size_t)ka * rw + (kb >> 5)], 1u << (kb & 31));
                atomicOr(&adj[(size_t)kb * rw + (ka >> 5)], 1u << (ka & 31));
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) limb_steps += __shfl_xor(limb_steps, o, 64);
    if (lane == 0 && limb_steps && work) atomicAdd(work, (unsigned long long)limb_steps);
}

// one block per job: connected components by label propagation over the adjacency bit matrix
__global__ __launch_bounds__(256) void k_components(Words W, Jobs J, int final_pass) {
    __shared__ int changed;
    uint32_t job = blockIdx.x;
    if (job >= J.n) return;
    uint32_t w0 = J.w0[job], R = J.R[job], nl = J.nl[job];
    if (nl == 0) return;
    const uint32_t rw = (R + 31) >> 5;
    const uint32_t *adj = J.adj + J.adj_off[job];
    uint32_t *label = W.label + w0;
    for (uint32_t a = threadIdx.x; a < nl; a += blockDim.x) {
        bool any = false;
        for (uint32_t x = 0; x < rw; x++) if (adj[(size_t)a * rw + x]) { any = true; break; }
        W.has_match[w0 + a] = any ? 1 : 0;
    }
    // Labels are written by atomics (L2) and read around the L1 (agent-scope relaxed loads), so every wave of the block
    // sees the current value after the barrier.
#define LBL(i) __hip_atomic_load(&label[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
    // Every edge is handled by its higher-numbered end only (the adjacency is symmetric): that leaf pulls the smaller
    // label and pushes its own to the neighbour when that one's is larger.  The pivots' rows (hundreds of bits at a
    // deep site: a pivot matches nearly every leaf) are therefore never walked; their few lower neighbours do the work.
    for (int iter = 0; iter < (int)nl + 2; iter++) {
        __syncthreads();
        if (threadIdx.x == 0) changed = 0;
        __syncthreads();
        for (uint32_t a = threadIdx.x; a < nl; a += blockDim.x) {
            const uint32_t l0 = LBL(a);
            uint32_t l = l0;
            const uint32_t xa = a >> 5, low = (1u << (a & 31)) - 1u;
            for (uint32_t x = 0; x <= xa; x++) {
                uint32_t bits = adj[(size_t)a * rw + x];
                if (x == xa) bits &= low;
                while (bits) { int b = __builtin_ctz(bits); bits &= bits - 1; uint32_t lb = LBL(x * 32 + b); if (lb < l) l = lb; }
            }
            uint32_t ll = LBL(l);   // pointer jumping
            if (ll < l) l = ll;
            if (l < l0) { atomicMin(&label[a], l); changed = 1; }
            for (uint32_t x = 0; x <= xa; x++) {
                uint32_t bits = adj[(size_t)a * rw + x];
                if (x == xa) bits &= low;
                while (bits) { int b = __builtin_ctz(bits); bits &= bits - 1; if (LBL(x * 32 + b) > l) { atomicMin(&label[x * 32 + b], l); changed = 1; } }
            }
        }
        __syncthreads();
        if (!changed) break;
    }
#undef LBL
    __syncthreads();
    // the largest component among those of the first N_PIVOTS leaves (k_leaf_myers mode 1 skips its members)
    __shared__ uint32_t pc[N_PIVOTS];
    if (threadIdx.x < N_PIVOTS) pc[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t np = nl < N_PIVOTS ? nl : N_PIVOTS;
    for (uint32_t a = threadIdx.x; a < nl; a += blockDim.x) {
        uint32_t l = label[a];
        for (uint32_t p = 0; p < np; p++) if (label[p] == l) { atomicAdd(&pc[p], 1u); break; }
    }
    __syncthreads();
    __shared__ uint32_t s_giant;
    if (threadIdx.x == 0) {
        uint32_t best = 0;
        for (uint32_t p = 1; p < np; p++) if (pc[p] > pc[best]) best = p;
        s_giant = label[best];
        J.giant[job] = s_giant;
    }
    if (!final_pass) {   // list of the leaves outside the giant component for k_leaf_myers mode 1: one reservation per job
        __shared__ uint32_t s_nout, s_obase;
        if (threadIdx.x == 0) s_nout = 0;
        __syncthreads();
        const uint32_t giant = s_giant;
        uint32_t mine = 0;
        for (uint32_t a = threadIdx.x; a < nl; a += blockDim.x) if (label[a] != giant) mine++;
        uint32_t at = mine ? atomicAdd(&s_nout, mine) : 0;
        __syncthreads();
        if (threadIdx.x == 0 && s_nout) s_obase = atomicAdd(&W.n_list[2], s_nout);
        __syncthreads();
        at += s_obase;
        for (uint32_t a = threadIdx.x; a < nl; a += blockDim.x) if (label[a] != giant) W.out_list[at++] = w0 + a;
    }
    if (final_pass) {   // list of the cluster representatives for k_cluster_vote: one global reservation per job
        __shared__ uint32_t s_nrep, s_rbase;
        __syncthreads();
        if (threadIdx.x == 0) s_nrep = 0;
        __syncthreads();
        uint32_t mine = 0;
        for (uint32_t a = threadIdx.x; a < nl; a += blockDim.x) if (label[a] == a) mine++;
        uint32_t at = mine ? atomicAdd(&s_nrep, mine) : 0;
        __syncthreads();
        if (threadIdx.x == 0 && s_nrep) s_rbase = atomicAdd(&W.n_list[1], s_nrep);
        __syncthreads();
        at += s_rbase;
        for (uint32_t a = threadIdx.x; a < nl; a += blockDim.x) if (label[a] == a) W.rep_list[at++] = w0 + a;
    }
}

// ---- trie node sums ---------------------------------------------------------------------------------------------------
// Every word adds its qualities / a counter / an end-of-word counter to the nodes along its path (Trie.add,
// flanktrie.py:31-59); the covering leaf's npos row names them.  At a 1000x site ~900 words share the same trunk nodes,
// so the sums are first combined in an LDS table (per depth NS_SLOTS nodes, keyed by position) per block of NS_WORDS
// words and flushed with one global atomic per node; a depth with more distinct nodes than slots goes to HBM directly.
#define NS_WORDS 128
#define NS_SLOTS 8
__global__ __launch_bounds__(1024) void k_node_sums(Words W, Jobs J, uint32_t max_len) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t n = max_len * NS_SLOTS;
    uint32_t *tk = reinterpret_cast<uint32_t *>(smem), *tq = tk + n, *tc = tq + n, *te = tc + n;   // [slot][depth]
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) { tk[i] = 0xFFFFFFFFu; tq[i] = 0; tc[i] = 0; te[i] = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t w_end = (blockIdx.x + 1) * NS_WORDS < W.n ? (blockIdx.x + 1) * NS_WORDS : W.n;
    // a wave takes NS_WORDS / waves consecutive words.  Where a word's nodes are is five dependent loads away (job ->
    // word range, covering leaf -> its word -> its npos row): lane i resolves that chain for the wave's i-th word, so
    // the chains of all its words are in flight together, and the results are broadcast word by word.
    const uint32_t wpw = NS_WORDS / (blockDim.x >> 6);
    const uint32_t wbase = blockIdx.x * NS_WORDS + (threadIdx.x >> 6) * wpw;
    uint32_t my_np = 0, my_q = 0;
    int my_len = 0;
    if ((uint32_t)lane < wpw && wbase + lane < w_end) {
        const uint32_t w = wbase + lane;
        const uint32_t job = W.job[w], w0 = J.w0[job];
        const uint32_t k = W.cover[w];
        if (k != 0xFFFFFFFFu) {   // always: every word is a prefix of a leaf
            const uint32_t wl = w0 + W.leaf_word[w0 + k];
            my_np = (uint32_t)W.off[wl]; my_q = (uint32_t)W.off[w]; my_len = (int)W.len[w];   // offsets < 2^32 (checked by the host)
        }
    }
    for (uint32_t i = 0; i < wpw; i++) {
        const int len = __shfl(my_len, (int)i, 64);
        if (!len) continue;
        const uint32_t *np = W.npos + __shfl(my_np, (int)i, 64);
        const uint8_t *q = W.q + __shfl(my_q, (int)i, 64);
        for (int d = lane; d < len; d += 64) {
            const uint32_t pos = np[d], qv = q[d];
            int sl = 0;
            for (; sl < NS_SLOTS; sl++) { uint32_t old = atomicCAS(&tk[sl * max_len + d], 0xFFFFFFFFu, pos); if (old == 0xFFFFFFFFu || old == pos) break; }
            if (sl < NS_SLOTS) {
                const uint32_t i2 = sl * max_len + d;
                atomicAdd(&tq[i2], qv); atomicAdd(&tc[i2], 1u);
                if (d == len - 1) atomicAdd(&te[i2], 1u);
            } else {
                atomicAdd(&W.agg_qc[pos], (unsigned long long)qv | (1ull << NS_C_SHIFT));
                if (d == len - 1) atomicAdd(&W.agg_e[pos], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t pos = tk[i];
        if (pos == 0xFFFFFFFFu) continue;
        atomicAdd(&W.agg_qc[pos], (unsigned long long)tq[i] | ((unsigned long long)tc[i] << NS_C_SHIFT));
        if (te[i]) atomicAdd(&W.agg_e[pos], te[i]);
    }
}

// ---- clusters -------------------------------------------------------------------------------------------------------
// one wave per leaf: share(l) = longest prefix shared with a leaf of the same cluster that comes before it.  The nodes
// of its path above that depth are counted by that earlier leaf.
__global__ __launch_bounds__(256) void k_leaf_share(Words W, Jobs J, const uint16_t *lcpm) {
    const int lane = threadIdx.x & 63;
    const uint32_t n_leaf = W.n_list[4];
    for (uint32_t it = blockIdx.x * 4 + (threadIdx.x >> 6); it < n_leaf; it += gridDim.x * 4) {
        const uint32_t slot = W.leaf_list[it];
        const uint32_t job = W.job[slot], w0 = J.w0[job], R = J.R[job], kl = slot - w0;
        const uint32_t ka = W.label[slot], lw = W.leaf_word[slot];
        const size_t rstride = (size_t)((R + 31) >> 5) * 32;
        const uint16_t *lcp0 = lcpm + J.adj_off[job] * 32;
        int share = 0;
        for (uint32_t kp = lane; kp < kl; kp += 64)
            if (W.label[w0 + kp] == ka) { int c = (int)lcp0[(size_t)kp * rstride + lw]; if (c > share) share = c; }
        for (int o = 32; o > 0; o >>= 1) { int other = __shfl_xor(share, o, 64); if (other > share) share = other; }
        if (lane == 0) W.share[slot] = (uint32_t)share;
    }
}

// one block per cluster (representative leaf slot): column vote over the sub-trie of its leaves (make_consensus_word
// flanktrie.py:306-342), merged count (make_subtrie 89-118) and the cluster's place in the reference's list order.
// A member leaf adds the nodes of its path from depth share(l) on.  dynamic LDS: hq[max_len*16] (u32) | hc[max_len*16]
// The walk of one member is a chain of dependent loads (word offset -> npos -> node sums), so the members of a tile of
// leaves are first compacted into an LDS list (offset, length, share resolved by one thread per leaf, all at once), then
// dealt to the waves CV_BATCH at a time: a wave has the loads of CV_BATCH members in flight before the first LDS add.
#define CV_THREADS 512
#define CV_BATCH 4
__global__ __launch_bounds__(CV_THREADS) void k_cluster_vote(DevParams P, Words W, Jobs J, uint32_t max_len, char *cl_chars, uint32_t cl_stride) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *hq = reinterpret_cast<uint32_t *>(smem);
    uint32_t *hc = hq + (size_t)max_len * 16;
    __shared__ int s_maxd, s_stop;
    __shared__ uint32_t s_minrank, s_words, s_nmem[2];
    __shared__ uint32_t s_moff[CV_THREADS], s_mls[CV_THREADS];   // members of the tile: word offset, len | share << 16
    const int lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const uint32_t n_rep = W.n_list[1];
    for (uint32_t it = blockIdx.x; it < n_rep; it += gridDim.x) {
    __syncthreads();
    const uint32_t slot = W.rep_list[it];
    const uint32_t job = W.job[slot], w0 = J.w0[job], nl = J.nl[job], ka = slot - w0;
    for (uint32_t d = threadIdx.x; d < max_len * 16; d += blockDim.x) { hq[d] = 0; hc[d] = 0; }
    if (threadIdx.x == 0) { s_maxd = 0; s_minrank = 0xFFFFFFFFu; s_words = 0; s_nmem[0] = 0; s_nmem[1] = 0; }
    __syncthreads();
    uint32_t words = 0;
    int maxlen = 0; uint32_t minrank = 0xFFFFFFFFu;
    uint32_t tile = 0;
    for (uint32_t k0 = 0; k0 < nl; k0 += blockDim.x, tile ^= 1u) {
        const uint32_t kl = k0 + threadIdx.x;
        const bool member = kl < nl && W.label[w0 + kl] == ka;
        uint32_t m_off = 0, m_ls = 0;
        if (member) {
            const uint32_t wl = w0 + W.leaf_word[w0 + kl], len = W.len[wl], share = W.share[w0 + kl];
            m_off = (uint32_t)W.off[wl];   // offsets < 2^32 (checked by the host)
            m_ls = len | (share << 16);
            if ((int)len > maxlen) maxlen = (int)len;
            const uint32_t rk = W.leaf_rank[w0 + kl]; if (rk < minrank) minrank = rk;
        }
        const unsigned long long mb = __ballot(member);
        uint32_t base = 0;
        if (lane == 0 && mb) base = atomicAdd(&s_nmem[tile], (uint32_t)__popcll(mb));
        base = __shfl(base, 0, 64);
        if (member) { const uint32_t at = base + (uint32_t)__popcll(mb & ((1ull << lane) - 1ull)); s_moff[at] = m_off; s_mls[at] = m_ls; }
        __syncthreads();
        const uint32_t nm = s_nmem[tile];
        if (threadIdx.x == 0) s_nmem[tile ^ 1u] = 0;   // next tile's counter: last read before the barrier that ended the previous tile
        for (uint32_t i = wave * CV_BATCH; i < nm; i += nwaves * CV_BATCH) {
            uint32_t o[CV_BATCH], np[CV_BATCH], code[CV_BATCH];
            int len[CV_BATCH], d[CV_BATCH];
            bool ok[CV_BATCH];
            NodeSum ns[CV_BATCH];
#pragma unroll
            for (int u = 0; u < CV_BATCH; u++) {
                const bool in = i + u < nm;
                o[u] = in ? s_moff[i + u] : 0u;
                const uint32_t ls = in ? s_mls[i + u] : 0u;
                len[u] = (int)(ls & 0xFFFFu); d[u] = (int)(ls >> 16) + lane;
                ok[u] = d[u] < len[u];
                np[u] = 0; code[u] = 0;
                if (ok[u]) { np[u] = W.npos[(size_t)o[u] + d[u]]; code[u] = W.ch[(size_t)o[u] + d[u]] & 15; }
            }
#pragma unroll
            for (int u = 0; u < CV_BATCH; u++) { ns[u].qc = 0; ns[u].e = 0; if (ok[u]) { ns[u].qc = W.agg_qc[np[u]]; ns[u].e = W.agg_e[np[u]]; } }
#pragma unroll
            for (int u = 0; u < CV_BATCH; u++) if (ok[u]) {
                atomicAdd(&hq[d[u] * 16 + code[u]], (uint32_t)(ns[u].qc & NS_Q_MASK));
                atomicAdd(&hc[d[u] * 16 + code[u]], (uint32_t)(ns[u].qc >> NS_C_SHIFT));
                words += ns[u].e;                            // words ending here: prefixes of a leaf of the cluster
            }
#pragma unroll
            for (int u = 0; u < CV_BATCH; u++)               // a path longer than 64 below its shared depth
                for (int dd = d[u] + 64; dd < len[u]; dd += 64) {
                    const uint32_t np2 = W.npos[(size_t)o[u] + dd];
                    NodeSum n2; n2.qc = W.agg_qc[np2]; n2.e = W.agg_e[np2];
                    const uint32_t c2 = W.ch[(size_t)o[u] + dd] & 15;
                    atomicAdd(&hq[dd * 16 + c2], (uint32_t)(n2.qc & NS_Q_MASK));
                    atomicAdd(&hc[dd * 16 + c2], (uint32_t)(n2.qc >> NS_C_SHIFT));
                    words += n2.e;
                }
        }
        __syncthreads();
    }
    if (maxlen) atomicMax(&s_maxd, maxlen);
    if (minrank != 0xFFFFFFFFu) atomicMin(&s_minrank, minrank);
    if (words) atomicAdd(&s_words, words);
    __syncthreads();
    const int maxd = s_maxd;
    if (threadIdx.x == 0) s_stop = maxd;
    __syncthreads();
    // stop at the first depth with fewer than mcc voters
    for (int d = threadIdx.x; d < maxd; d += blockDim.x) {
        uint32_t tot = 0;
        for (int c = 0; c < 16; c++) tot += hc[d * 16 + c];
        if ((int64_t)tot < (int64_t)P.mcc) atomicMin(&s_stop, d);
    }
    __syncthreads();
    const int stop = s_stop;
    if (threadIdx.x == 0) {
        W.cl_count[slot] = (int32_t)s_words;
        W.cl_len[slot] = stop;
        W.cl_key[slot] = ((W.has_match[slot] ? 1u : 0u) << 31) | (s_minrank & 0x7fffffffu);
    }
    char *out = cl_chars + (size_t)slot * cl_stride;
    const int side = (int)(job & 1);
    for (int d = threadIdx.x; d < stop; d += blockDim.x) {
        int best = -1, nbest = 0;
        for (int c = 0; c < 16; c++) {
            if (!hc[d * 16 + c]) continue;
            if (best < 0 || hq[d * 16 + c] > hq[d * 16 + best]) { best = c; nbest = 1; }
            else if (hq[d * 16 + c] == hq[d * 16 + best]) nbest++;
        }
        char chv = nbest > 1 ? 'N' : d_code_char(best < 0 ? 15 : best);   // get_best_qual_chars flanktrie.py:344-353
        out[side == 0 ? d : stop - 1 - d] = chv;                     // 3p consensus is reversed back (findflanks.py:894-895)
    }
    }
}

// one block per job: filters + best consensus (findflanks.py:241-346): max by (count, len), the last in list order
// wins ties; list order = singletons in leaf (DFS) order, then components by their first leaf (quirk Q12 build rule),
// which is the order of cl_key.  Then the winner's characters are copied to the job's output row.
__global__ __launch_bounds__(256) void k_job_best(DevParams P, Words W, Jobs J, ConsensusIO io, const char *cl_chars, uint32_t cl_stride) {
    __shared__ uint32_t s_npl, s_npb;
    __shared__ unsigned long long s_hi;   // (count << 32) | len of the best
    __shared__ uint32_t s_key, s_slot;
    const uint32_t job = blockIdx.x;
    if (job >= J.n) return;
    if (J.R[job] == 0) return;   // inactive, or done by k_job_small: its outputs stand
    const uint32_t w0 = J.w0[job], nl = J.nl[job];
    if (threadIdx.x == 0) { s_npl = 0; s_npb = 0; s_hi = 0; s_key = 0; s_slot = 0xFFFFFFFFu; }
    __syncthreads();
    uint32_t npl = 0, npb = 0;
    for (uint32_t k = threadIdx.x; k < nl; k += blockDim.x) {
        const uint32_t slot = w0 + k;
        if (W.label[slot] != k) continue;
        const int32_t cnt = W.cl_count[slot], len = W.cl_len[slot];
        if (len < P.minlen) continue;
        npl++;
        if (cnt < P.mincount) continue;
        npb++;
        atomicMax(&s_hi, ((unsigned long long)(uint32_t)cnt << 32) | (unsigned long long)(uint32_t)len);
    }
    if (npl) atomicAdd(&s_npl, npl);
    if (npb) atomicAdd(&s_npb, npb);
    __syncthreads();
    const unsigned long long hi = s_hi;
    const bool any = s_npb > 0;
    if (any)
        for (uint32_t k = threadIdx.x; k < nl; k += blockDim.x) {
            const uint32_t slot = w0 + k;
            if (W.label[slot] != k) continue;
            const int32_t cnt = W.cl_count[slot], len = W.cl_len[slot];
            if (len < P.minlen || cnt < P.mincount) continue;
            if ((((unsigned long long)(uint32_t)cnt << 32) | (unsigned long long)(uint32_t)len) == hi) atomicMax(&s_key, W.cl_key[slot] + 1u);   // +1: key 0 is a valid key
        }
    __syncthreads();
    if (any)
        for (uint32_t k = threadIdx.x; k < nl; k += blockDim.x) {
            const uint32_t slot = w0 + k;
            if (W.label[slot] != k) continue;
            const int32_t cnt = W.cl_count[slot], len = W.cl_len[slot];
            if (len < P.minlen || cnt < P.mincount) continue;
            if ((((unsigned long long)(uint32_t)cnt << 32) | (unsigned long long)(uint32_t)len) == hi && W.cl_key[slot] + 1u == s_key) s_slot = slot;
        }
    __syncthreads();
    const uint32_t best = s_slot;
    if (threadIdx.x == 0) {
        io.n_pass_len[job] = s_npl; io.n_pass_both[job] = s_npb;
        J.best[job] = best;
        if (best != 0xFFFFFFFFu) { io.cons_count[job] = (int32_t)(hi >> 32); io.cons_len[job] = (int32_t)(hi & 0xFFFFFFFFull); }
    }
    if (best == 0xFFFFFFFFu) return;
    const int len = (int)(hi & 0xFFFFFFFFull);
    const char *src = cl_chars + (size_t)best * cl_stride;
    char *dst = io.cons_chars + (size_t)job * io.cons_stride;
    for (int d = threadIdx.x; d < len; d += blockDim.x) dst[d] = src[d];
}

// ---- small jobs: the whole consensus of one site-side in ONE workgroup, state in LDS -------------------------------------
// A 100-200x site-side has ~50-200 words and a few dozen leaves: ~10 KB of characters.  The device-wide kernels above pay,
// per job and per kernel, a chain of dependent global loads (job -> word range -> offsets -> characters); with 20 000 such
// jobs (50 Mb at 200x) that chain IS the run time, and with 40 jobs (an isolate) the dozen launches are.  Here one
// workgroup copies the job's characters into LDS once and runs every step on them:
//   leaves            word v is no leaf iff another word extends it (or equals it and comes first): the job's longest
//                     word settles the error-free words, the rest scan the job
//   M[a][v]           longest common prefix of leaf a and word v (one byte each: words here are <= 255 characters)
//   DFS order         T[a][b] = first word that shares more than LCP(a, b) characters with a; b is visited before a iff
//                     T[b][a] < T[a][b] (children in insertion order, flanktrie.py:157-170)
//   clusters          Myers bit-vector distances in rounds of 4 pattern leaves (match vectors in LDS): the first 4 leaves
//                     against all later ones, then every leaf outside the largest component against the leaves of other
//                     components (every cross-component pair is tested, so the components are those of the full graph);
//                     label propagation on the adjacency bits
//   vote              per cluster: leaf a contributes the nodes of its path below share(a) = deepest node it has in common
//                     with an earlier member.  The node of a at depth d sums the words v with M[a][v] > d, so every such
//                     (a, v) adds q_v[d] and a count at (d, character) for d in [share(a), M[a][v]): a job's characters
//                     are each added once for the trunk and a few more times for the branches.  Sum of qualities (<= 256 x
//                     255) and number of words (<= 256) of a (depth, character) share one 32-bit LDS word.
//   count             words v with M[a][v] == len(v) for some member a (make_subtrie, flanktrie.py:89-118)
// and writes the job's result exactly as k_job_best does.  A job that does not fit (more than SJ_MAX_R words, a word longer
// than SJ_MAX_LEN, tables beyond the LDS budget of the launch) is left to a second launch with a larger budget or to the
// device-wide kernels: J.Rbig says which.
#define SJ_THREADS 256
#define SJ_MAX_LEN 255

__device__ __forceinline__ int d_lcp16(const uint8_t *a, int la, const uint8_t *b, int lb) {
    const int m = la < lb ? la : lb;
    for (int o = 0; o < m; o += 16) {
        const uint4 x = *reinterpret_cast<const uint4 *>(a + o), y = *reinterpret_cast<const uint4 *>(b + o);
        const uint32_t d[4] = {x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w};
        if (d[0] | d[1] | d[2] | d[3]) {
            int k = m;
#pragma unroll
            for (int i = 3; i >= 0; i--) if (d[i]) k = o + 4 * i + (__builtin_ctz(d[i]) >> 3);
            return k < m ? k : m;
        }
    }
    return m;
}

// block-wide maximum / sum over SJ_THREADS threads (4 waves); sh = 4 words of LDS
__device__ __forceinline__ uint32_t sj_block_max(uint32_t v, uint32_t *sh) {
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(v, o, 64); if (t > v) v = t; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t r = sh[0];
    for (int w = 1; w < SJ_THREADS / 64; w++) if (sh[w] > r) r = sh[w];
    return r;
}
__device__ __forceinline__ uint32_t sj_block_sum(uint32_t v, uint32_t *sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t r = 0;
    for (int w = 0; w < SJ_THREADS / 64; w++) r += sh[w];
    return r;
}

// words here are at most SJ_MAX_LEN = 255 characters: four limbs at most (the 8-limb variant would cost the kernel half its waves)
__device__ __forceinline__ int d_myers_le4(const uint64_t *peq, int need, const uint8_t *text, int m) {
    switch (need) {
        case 1: return d_myers<1>(peq, text, m);
        case 2: return d_myers<2>(peq, text, m);
        case 3: return d_myers<3>(peq, text, m);
        default: return d_myers<4>(peq, text, m);
    }
}

// maximum over the 64 lanes, every lane gets it: row-wise by DPP lane permutations (no LDS round trips as in __shfl_xor), then
// lane 15 of a row to the next row (row_bcast:15, rows 1 and 3), lane 31 to the upper half (row_bcast:31); lane 63 holds the result
__device__ __forceinline__ uint32_t d_wave_max(uint32_t v) {
#define SJ_DPP(x, ctrl, rows) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), ctrl, rows, 0xF, false)
    uint32_t t;
    t = SJ_DPP(v, 0xB1, 0xF); v = t > v ? t : v;     // quad_perm [1,0,3,2]
    t = SJ_DPP(v, 0x4E, 0xF); v = t > v ? t : v;     // quad_perm [2,3,0,1]
    t = SJ_DPP(v, 0x141, 0xF); v = t > v ? t : v;    // row_half_mirror
    t = SJ_DPP(v, 0x140, 0xF); v = t > v ? t : v;    // row_mirror
    t = SJ_DPP(v, 0x142, 0xA); v = t > v ? t : v;    // row_bcast:15
    t = SJ_DPP(v, 0x143, 0xC); v = t > v ? t : v;    // row_bcast:31
#undef SJ_DPP
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

#define SJ_A16(x) (((x) + 15u) & ~15u)
// diagnostic build only (make SJ_STAMP=1): shader cycles of thread 0 between the phases of k_job_small, summed over the jobs
#ifdef SJ_STAMP
__device__ unsigned long long g_sj_cyc[16], g_sj_need[40];   // g_sj_need: jobs by dynamic LDS need, 4 KB bins
// kept in registers and added to the totals once, at the end of the kernel: an atomic per stamp would make the next barrier
// wait for it (thousands of workgroups on one address) and charge that to the phase behind the stamp
#define SJ_T(i) do { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); sj_acc[i] += t_ - sj_last; sj_last = t_; } } while (0)
extern "C" int mfk_debug_sj(unsigned long long *out) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sj_cyc), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out + 16, HIP_SYMBOL(g_sj_need), sizeof(unsigned long long) * 40) != hipSuccess) return 1;
    unsigned long long z[40] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_sj_need), z, sizeof z) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_sj_cyc), z, sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1;
}
#else
#define SJ_T(i) do { } while (0)
#endif
#ifndef SJ_WAVES
#define SJ_WAVES 4
#endif
__global__ __launch_bounds__(SJ_THREADS) __attribute__((amdgpu_waves_per_eu(SJ_WAVES, SJ_WAVES))) void k_job_small(DevReads RD, DevParams P, Words W, Jobs J, ConsensusIO io, CMeta *meta, uint32_t lds_bytes, int pass, int fused) {
    extern __shared__ __align__(16) unsigned char smem[];
    // static LDS is part of every workgroup's footprint (four workgroups per CU: 40 KB each): indices and lengths are bytes
    // (at most SJ_MAX_R = 256 words of at most SJ_MAX_LEN = 255 characters)
    __shared__ uint16_t s_loff[SJ_MAX_R];                               // offset of a word's characters in units of 16 bytes
    __shared__ __align__(16) uint8_t s_leafw[SJ_MAX_R + 16];            // word of a leaf (read four at a time)
    __shared__ uint8_t s_len[SJ_MAX_R], s_label[SJ_MAX_R], s_rank[SJ_MAX_R], s_list[SJ_MAX_R];
    __shared__ uint8_t s_hasm[SJ_MAX_R];
    __shared__ __align__(16) uint32_t s_sig[SJ_MAX_R], s_usig[SJ_MAX_R], s_wr[SJ_MAX_R + 8];   // signatures (leaves); reach | character offset << 8 of a word (vote)
    __shared__ uint32_t s_red[8], s_u[8];
    __shared__ int s_changed, s_other;   // s_other: bit c = a base code c other than A, C, G, T takes part in the vote
    const uint32_t job = blockIdx.x;
    if (job >= J.n) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t R = J.R[job];
    if (pass == 0) {
        if (R == 0 || R > SJ_MAX_R) {
            if (tid == 0) { J.Rbig[job] = R; if (R) atomicAdd(&meta->n_big, 1u); }
            return;
        }
    } else if (!(J.Rbig[job] >> 31)) return;
    // every exit below is taken by the whole workgroup
#define SJ_FAIL(permanent)                                                                                          \
    do {                                                                                                            \
        if (tid == 0) {                                                                                             \
            if ((permanent) || pass == 2) { J.Rbig[job] = R; atomicAdd(&meta->n_big, 1u); }                         \
            else { J.Rbig[job] = R | 0x80000000u; atomicAdd(&meta->n_retry, 1u); }                                  \
        }                                                                                                           \
        return;                                                                                                     \
    } while (0)
#ifdef SJ_STAMP
    unsigned long long sj_last = __builtin_amdgcn_s_memtime();
    unsigned long long sj_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const uint32_t w0 = J.w0[job];
    const uint64_t goff0 = W.off[w0];
    const uint32_t C = (uint32_t)(W.off[w0 + R] - goff0);     // characters of the job, every word padded to 16
    int mylen = 0;
    uint32_t myloff = 0;
    WordSrc mysrc;
    mysrc.o1 = mysrc.o2 = 0; mysrc.l1 = mysrc.l2 = mysrc.q1len = mysrc.q2len = 0;
    if ((uint32_t)tid < R) {
        mylen = (int)W.len[w0 + tid];
        myloff = (uint32_t)((W.off[w0 + tid] - goff0) >> 4);
        if (fused) mysrc = W.src[w0 + tid];   // (with the two loads above in one round trip)
        s_len[tid] = (uint8_t)mylen;   // a longer word makes the job fail just below
        s_loff[tid] = (uint16_t)myloff;
    }
    const uint32_t lkey = sj_block_max(((uint32_t)mylen << 16) | (0xFFFFu - (uint32_t)tid), s_red);   // longest word, the first of them
    const uint32_t maxlen = lkey >> 16;
    if (maxlen > SJ_MAX_LEN) SJ_FAIL(true);
    if (C + 4096u > lds_bytes) SJ_FAIL(false);
    uint8_t *ch = smem;   // characters of the job's words (the qualities are read once, by the vote, straight from memory)
    // fused fill (the usual case: k_word_fill has not run): the characters come straight from the sequence pool, 16 per thread
    // and step as k_word_fill makes them; W.ch / W.q are written only for jobs that end up with the device-wide kernels.
    // The qualities go to W.q as k_word_fill writes them (the vote reads them once, from there or staged in LDS); the
    // characters never leave the chip.
    const int side_j = (int)(job & 1u);
    if (fused) {
        // the words' sources go through LDS (behind the characters: the tables that live there are laid out later), so that a
        // thread can take any chunk: all chunks of the job are in flight together, two dependent round trips in all
        WordSrc *ssrc = reinterpret_cast<WordSrc *>(smem + SJ_A16(C));
        if (SJ_A16(C) + R * (uint32_t)sizeof(WordSrc) > lds_bytes) SJ_FAIL(false);
        if ((uint32_t)tid < R) ssrc[tid] = mysrc;
        __syncthreads();
        const uint32_t nch = C >> 4;
#pragma unroll 4
        for (uint32_t i = tid; i < nch; i += SJ_THREADS) {
            uint32_t u = 0;   // the word of chunk i: the last one that starts at or before it
#pragma unroll
            for (uint32_t step = SJ_MAX_R / 2; step; step >>= 1) if (u + step < R && (uint32_t)s_loff[u + step] <= i) u += step;
            const WordSrc sw = ssrc[u];
            uint32_t c4[4], q4[4];
            d_word_chunk(RD, sw, (int)s_len[u], side_j, (int)((i - (uint32_t)s_loff[u]) << 4), c4, q4);
            reinterpret_cast<uint4 *>(ch)[i] = make_uint4(c4[0], c4[1], c4[2], c4[3]);
            *reinterpret_cast<uint4 *>(W.q + goff0 + ((uint64_t)i << 4)) = make_uint4(q4[0], q4[1], q4[2], q4[3]);
        }
    } else {
        const uint4 *src = reinterpret_cast<const uint4 *>(W.ch + goff0);
        for (uint32_t i = tid; i < (C >> 4); i += SJ_THREADS) reinterpret_cast<uint4 *>(ch)[i] = src[i];
    }
    __syncthreads();
    SJ_T(0);   // load
    // ---- leaves ---------------------------------------------------------------------------------------------------
    // Up to four reference words L_0 .. L_3: L_0 is the job's (first) longest word, L_k the longest word not yet settled.
    // p_k(v) = LCP(v, L_k).  A word that is a prefix of a reference is no leaf; a reference is a leaf (nothing longer is
    // left that could extend it).  A word v still unsettled leaves every reference at p_k(v) < len(v), and so does every
    // word that extends v: candidates must have the same four p_k ("signature"), and only those are compared in full.
    // The signatures also give most of M below without looking at a character.
    const uint8_t *mych = ch + ((uint32_t)s_loff[tid < (int)R ? tid : 0] << 4);
    bool unsettled = (uint32_t)tid < R, leaf = false;
    uint32_t sig = 0;
    for (int k = 0; k < 4; k++) {
        const uint32_t key = sj_block_max(unsettled ? (((uint32_t)mylen << 16) | (0xFFFFu - (uint32_t)tid)) : 0u, s_red);
        if (!key) break;   // block-uniform: every word is settled
        const uint32_t Lk = 0xFFFFu - (key & 0xFFFFu);
        const int lenL = (int)(key >> 16);
        if ((uint32_t)tid < R) {
            const int p = (uint32_t)tid == Lk ? mylen : d_lcp16(mych, mylen, ch + ((uint32_t)s_loff[Lk] << 4), lenL);
            sig |= (uint32_t)p << (8 * k);
            if ((uint32_t)tid == Lk) { leaf = true; unsettled = false; }
            else if (unsettled && p == mylen) unsettled = false;   // a prefix of L_k (or a later copy of it)
        }
    }
    if ((uint32_t)tid < R) s_sig[tid] = sig;
    uint32_t nU;
    {
        const uint32_t k = block_excl_scan(unsettled ? 1u : 0u, s_red, &nU);
        if (unsettled) { s_list[k] = (uint8_t)tid; s_usig[k] = sig; }
    }
    __syncthreads();
    if (unsettled) {
        bool cov = false;
        for (uint32_t i0 = 0; i0 < nU && !cov; i0 += 4) {
            uint32_t sg[4];
#pragma unroll
            for (int u = 0; u < 4; u++) sg[u] = i0 + u < nU ? s_usig[i0 + u] : ~sig;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (sg[u] != sig || cov) continue;
                const uint32_t w = s_list[i0 + u];
                if (w == (uint32_t)tid) continue;
                const int lb = (int)s_len[w];
                if (lb < mylen) continue;
                if (d_lcp16(mych, mylen, ch + ((uint32_t)s_loff[w] << 4), lb) >= mylen && (lb > mylen || w < (uint32_t)tid)) cov = true;
            }
        }
        leaf = !cov;
    }
    uint32_t NL;
    {
        const uint32_t k = block_excl_scan(leaf ? 1u : 0u, s_red, &NL);   // (its first barrier also ends the reads of s_list)
        if (leaf) { s_leafw[k] = (uint8_t)tid; s_label[k] = (uint8_t)k; }
    }
    // ---- LDS layout --------------------------------------------------------------------------------------------------
    const uint32_t NLW = (NL + 31) >> 5, needmax = (maxlen + 63) >> 6;
    uint32_t o = SJ_A16(C);
    uint32_t *adj = reinterpret_cast<uint32_t *>(smem + o); o += SJ_A16(NL * NLW * 4);
    uint8_t *bestc = smem + o; o += SJ_A16(maxlen + 1);
    const uint32_t RS = (R + 15u) & ~15u;
    uint8_t *M = smem + o; o += NL * RS;   // M[a][v] = longest common prefix of leaf a and word v
    // one region for three uses that follow each other: the match vectors of the edit-distance rounds (four pattern leaves), the
    // DFS table when clusters tie (T[a][b], NL x NL bytes, and a row of first words per wave), 3 KB of partial sums of the vote
    const uint32_t X = o;
    uint64_t *peq = reinterpret_cast<uint64_t *>(smem + X);
    {
        const uint32_t x1 = SJ_A16(NL * NL) + 4 * maxlen * 4, x2 = 3072 + 64, x3 = 4 * needmax * 128;
        o += x1 > x2 ? (x1 > x3 ? x1 : x3) : (x2 > x3 ? x2 : x3);
    }
#ifdef SJ_STAMP
    if (tid == 0) atomicAdd(&g_sj_need[(o >> 12) < 39 ? (o >> 12) : 39], 1ull);
#endif
    if (o > lds_bytes) SJ_FAIL(false);
    for (uint32_t i = tid; i < NL * NLW; i += SJ_THREADS) adj[i] = 0;
    __syncthreads();
    SJ_T(1);   // leaves
    // LCP of two words of the job.  Two words that leave a reference at different positions share exactly the shorter of the
    // two stretches, so the signatures settle it without a look at the characters unless all four positions agree.
    auto wlcp = [&](uint32_t sa, uint32_t wa, uint32_t sv, uint32_t v) -> uint32_t {
        const uint32_t x = sa ^ sv;
        if (x) {
            const int k8 = (__builtin_ctz(x) >> 3) << 3;   // first reference they leave at different positions
            const uint32_t pa = (sa >> k8) & 0xFFu, pv = (sv >> k8) & 0xFFu;
            return pa < pv ? pa : pv;
        }
        if (wa == v) return (uint32_t)s_len[v];   // a leaf and its own word: every lane meets that once, each at another step
        return (uint32_t)d_lcp16(ch + ((uint32_t)s_loff[wa] << 4), (int)s_len[wa], ch + ((uint32_t)s_loff[v] << 4), (int)s_len[v]);
    };
    // ---- DFS order (only when two clusters tie, see below) ---------------------------------------------------------------
    // T[a][b] = first word that shares more than LCP(a, b) characters with leaf a (that word made the trie node where the
    // branch of a starts below the node where a and b part); b is visited before a iff T[b][a] < T[a][b] (children in
    // insertion order).  One wave per leaf: lanes over the words put their index at depth LCP - 1 (atomicMin), a suffix
    // minimum over the depths gives the first word for every prefix of the leaf.
    auto dfs_ranks = [&]() {
        uint8_t *T = smem + X;
        uint32_t *tmin = reinterpret_cast<uint32_t *>(smem + X + SJ_A16(NL * NL)) + (uint32_t)wave * maxlen;
        for (uint32_t a = wave; a < NL; a += SJ_THREADS / 64) {
            const uint32_t wa = s_leafw[a], sa = s_sig[wa];
            const int la = (int)s_len[wa];
            for (int d = lane; d < la; d += 64) tmin[d] = 0xFFFFFFFFu;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            for (uint32_t v = lane; v < R; v += 64) { const uint32_t l = wlcp(sa, wa, s_sig[v], v); if (l > 0) atomicMin(&tmin[l - 1], v); }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            const int per = (la + 63) >> 6, d0 = lane * per;
            uint32_t mine = 0xFFFFFFFFu;
            for (int d = d0 + per - 1; d >= d0; d--) if (d < la) { const uint32_t v = tmin[d]; if (v < mine) mine = v; }
            uint32_t suf = mine;
            for (int s2 = 1; s2 < 64; s2 <<= 1) { const uint32_t other = __shfl_down(suf, s2, 64); if (lane + s2 < 64 && other < suf) suf = other; }
            uint32_t m = __shfl_down(suf, 1, 64);
            if (lane == 63) m = 0xFFFFFFFFu;
            for (int d = d0 + per - 1; d >= d0; d--) if (d < la) { const uint32_t v = tmin[d]; if (v < m) m = v; tmin[d] = m; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            for (uint32_t b = lane; b < NL; b += 64) {
                const uint32_t wb = s_leafw[b];
                T[a * NL + b] = b == a ? (uint8_t)0 : (uint8_t)tmin[wlcp(sa, wa, s_sig[wb], wb)];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        }
        __syncthreads();
        if ((uint32_t)tid < NL) {
            uint32_t rk = 0;
            for (uint32_t b = 0; b < NL; b++) if (b != (uint32_t)tid && T[b * NL + tid] < T[(uint32_t)tid * NL + b]) rk++;
            s_rank[tid] = (uint8_t)rk;
        }
        __syncthreads();
    };
    // ---- clusters ------------------------------------------------------------------------------------------------------
    // one round: up to 4 pattern leaves (plist[p0 ..] or p0 .. when plist is null) against the partners `mode` names
    uint32_t giant = 0, limb_steps = 0;
    auto round = [&](const uint8_t *plist, uint32_t p0, uint32_t np, int mode) {
        __syncthreads();   // the previous round's readers are done with the match vectors
        if ((uint32_t)wave < np) {
            const uint32_t a = plist ? plist[p0 + wave] : p0 + wave, wa = s_leafw[a];
            const int la = (int)s_len[wa];
            d_peq_wave(ch + ((uint32_t)s_loff[wa] << 4), la, (la + 63) >> 6, lane, peq + (uint32_t)wave * needmax * 16);
        }
        __syncthreads();
        for (uint32_t it = tid; it < np * NL; it += SJ_THREADS) {
            const uint32_t pi = it / NL, b = it - pi * NL, a = plist ? plist[p0 + pi] : p0 + pi;
            if (mode == 0) { if (b <= a) continue; }
            else {
                if (b == a) continue;
                const uint32_t lbl = s_label[b];
                if (lbl == s_label[a]) continue;
                if (lbl != giant && b < a) continue;   // two leaves outside the giant component: the lower one's round takes the pair
            }
            const uint32_t wa = s_leafw[a], wb = s_leafw[b];
            const int la = (int)s_len[wa], lb = (int)s_len[wb], m = la < lb ? la : lb;
            const uint8_t *tb = ch + ((uint32_t)s_loff[wb] << 4);
            int d = d_hamming16(ch + ((uint32_t)s_loff[wa] << 4), tb, m, m >> 2);
            if (4 * d > m) { d = d_myers_le4(peq + pi * needmax * 16, (m + 63) >> 6, tb, m); limb_steps += (uint32_t)(m * ((m + 63) >> 6)); }
            if (4 * d <= m) {   // similarity = 1 - d/m >= 0.75 (findflanks.py:1044-1048, quirk Q11)
                atomicOr(&adj[a * NLW + (b >> 5)], 1u << (b & 31));
                atomicOr(&adj[b * NLW + (a >> 5)], 1u << (a & 31));
            }
        }
    };
    auto components = [&]() {
        for (;;) {
            __syncthreads();
            if (tid == 0) s_changed = 0;
            __syncthreads();
            if ((uint32_t)tid < NL) {
                const uint32_t l0 = s_label[tid];
                uint32_t l = l0;
                for (uint32_t x = 0; x < NLW; x++) {
                    uint32_t bits = adj[(uint32_t)tid * NLW + x];
                    while (bits) { const int b = __builtin_ctz(bits); bits &= bits - 1; const uint32_t lb = s_label[x * 32 + b]; if (lb < l) l = lb; }
                }
                const uint32_t ll = s_label[l];
                if (ll < l) l = ll;
                if (l < l0) { s_label[tid] = (uint8_t)l; s_changed = 1; }
            }
            __syncthreads();
            if (!s_changed) break;
        }
    };
    if (NL >= 2) {
        round(nullptr, 0, NL < N_PIVOTS ? NL : N_PIVOTS, 0);
        __syncthreads();
        SJ_T(4);   // pivot round
        components();
        SJ_T(5);   // components
        // the largest component among those of the first leaves
        const uint32_t npv = NL < N_PIVOTS ? NL : N_PIVOTS;
        if (tid < N_PIVOTS) s_u[tid] = 0;
        __syncthreads();
        if ((uint32_t)tid < NL) {
            const uint32_t l = s_label[tid];
            for (uint32_t p = 0; p < npv; p++) if (s_label[p] == l) { atomicAdd(&s_u[p], 1u); break; }
        }
        __syncthreads();
        {
            uint32_t best = 0;
            for (uint32_t p = 1; p < npv; p++) if (s_u[p] > s_u[best]) best = p;
            giant = s_label[best];
        }
        uint32_t n_out;
        {
            const bool out = (uint32_t)tid < NL && s_label[tid] != giant;
            const uint32_t k = block_excl_scan(out ? 1u : 0u, s_red, &n_out);
            if (out) s_list[k] = (uint8_t)tid;
        }
        if (n_out) {
            for (uint32_t p0 = 0; p0 < n_out; p0 += 4) round(s_list, p0, n_out - p0 < 4 ? n_out - p0 : 4, 1);
            __syncthreads();
            SJ_T(6);   // rounds of the leaves outside the giant component
            components();
            SJ_T(7);
        }
    }
    __syncthreads();
    if (io.work) {   // one atomic per job (and none for a job without an edit distance), not one per wave
        const uint32_t all = sj_block_sum(limb_steps, s_red);
        if (tid == 0 && all) atomicAdd(io.work, (unsigned long long)all);
    }
    if ((uint32_t)tid < NL) {
        bool any = false;
        for (uint32_t x = 0; x < NLW; x++) if (adj[(uint32_t)tid * NLW + x]) { any = true; break; }
        s_hasm[tid] = any ? 1 : 0;
    }
    // ---- M[a][v] = LCP(leaf a, word v): one thread per word, four leaves per step (their signatures settle nearly every pair) ----
    if ((uint32_t)tid < NL) s_usig[tid] = s_sig[s_leafw[tid]];
    __syncthreads();
    SJ_T(11);
    if ((uint32_t)tid < R)
        for (uint32_t a0 = 0; a0 < NL; a0 += 4) {
            const uint4 sg4 = *reinterpret_cast<const uint4 *>(&s_usig[a0]);
            const uint32_t lw4 = *reinterpret_cast<const uint32_t *>(&s_leafw[a0]);
            const uint32_t sg[4] = {sg4.x, sg4.y, sg4.z, sg4.w}, lw[4] = {lw4 & 0xFFu, (lw4 >> 8) & 0xFFu, (lw4 >> 16) & 0xFFu, lw4 >> 24};
            uint32_t l[4];
            bool slow = false;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t x = sg[u] ^ sig;
                const int k8 = (__builtin_ctz(x | 0x80000000u) >> 3) << 3;   // first reference the two leave at different positions
                const uint32_t pa = (sg[u] >> k8) & 0xFFu, pv = (sig >> k8) & 0xFFu;
                l[u] = pa < pv ? pa : pv;
                if (!x && a0 + u < NL) { l[u] = (uint32_t)mylen; slow = slow || lw[u] != (uint32_t)tid; }   // same four positions: the leaf's own word, or (rarely) a look at the characters
            }
            if (slow) {
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (sg[u] == sig && a0 + u < NL && lw[u] != (uint32_t)tid)
                        l[u] = (uint32_t)d_lcp16(ch + ((uint32_t)s_loff[lw[u]] << 4), (int)s_len[lw[u]], mych, mylen);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) if (a0 + u < NL) M[(a0 + u) * RS + (uint32_t)tid] = (uint8_t)l[u];
        }
    SJ_T(6);
    // the cluster representatives, in leaf order
    uint32_t nrep;
    {
        const bool rep = (uint32_t)tid < NL && s_label[tid] == (uint32_t)tid;
        const uint32_t k = block_excl_scan(rep ? 1u : 0u, s_red, &nrep);   // (its barriers also publish M)
        if (rep) s_list[k] = (uint8_t)tid;
    }
    __syncthreads();
    SJ_T(2);   // LCP matrix
    // ---- every cluster's merged count and consensus length, without a vote -------------------------------------------------
    // The sub-trie of a cluster holds the node of v[:d+1] iff some member leaf shares more than d characters with word v:
    // reach(v) = max over the members a of M[a][v].  Word v votes at every depth d < reach(v) and is counted iff it ends
    // inside (reach(v) == len(v), make_subtrie flanktrie.py:89-118).  The voters of depth d are the words with reach > d, a
    // number that only falls with d, so the consensus stops (fewer than mcc voters, flanktrie.py:325-326) at the mcc-th largest
    // reach: count and length of EVERY cluster follow from the reach values alone, and only the cluster that wins
    // (findflanks.py:241-346) needs its characters voted.  One wave per cluster, lanes over the words (up to four each);
    // the members of a cluster are the set bits of a ballot over the leaves' labels, which every wave holds in registers.
    uint32_t lab[4], lenv[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t v = (uint32_t)lane + 64u * i;
        lab[i] = v < NL ? (uint32_t)s_label[v] : 0xFFFFFFFFu;
        lenv[i] = v < R ? (uint32_t)s_len[v] : 0xFFFFFFFFu;     // no reach equals it: not counted
    }
    const uint32_t nI = (R + 63u) >> 6;   // 64-word groups in use
    // reach of the wave's words (lane + 64 i) for the cluster with label r
    auto reach4 = [&](uint32_t r, uint32_t *rc, uint32_t *mlen, int only) {   // only >= 0: just that group of 64 words
        rc[0] = rc[1] = rc[2] = rc[3] = 0;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            if (64u * g >= NL) break;   // wave-uniform
            unsigned long long mask = __ballot(lab[g] == r);
            while (mask) {   // wave-uniform; two members per step, their rows read together
                const uint32_t a1 = 64u * g + (uint32_t)__builtin_ctzll(mask);
                mask &= mask - 1ull;
                uint32_t a2 = a1;
                if (mask) { a2 = 64u * g + (uint32_t)__builtin_ctzll(mask); mask &= mask - 1ull; }
                const uint8_t *m1 = M + a1 * RS + lane, *m2 = M + a2 * RS + lane;
                uint32_t x1[4], x2[4];
#pragma unroll
                for (int i = 0; i < 4; i++) { const bool in = (uint32_t)i < nI && (only < 0 || only == i); x1[i] = in ? (uint32_t)m1[in ? 64 * i : 0] : 0u; x2[i] = in ? (uint32_t)m2[in ? 64 * i : 0] : 0u; }
#pragma unroll
                for (int i = 0; i < 4; i++) { const uint32_t x = x1[i] > x2[i] ? x1[i] : x2[i]; if (x > rc[i]) rc[i] = x; }
                if (mlen) { const uint32_t l1 = s_len[s_leafw[a1]], l2 = s_len[s_leafw[a2]]; const uint32_t lm = l1 > l2 ? l1 : l2; if (lm > *mlen) *mlen = lm; }
            }
        }
    };
    for (uint32_t ir = wave; ir < nrep; ir += SJ_THREADS / 64) {
        const uint32_t r = s_list[ir];
        uint32_t rc[4], mlen = 0;
        reach4(r, rc, P.mcc <= 0 ? &mlen : nullptr, -1);
        uint32_t words = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) words += (uint32_t)__popcll(__ballot(rc[i] == lenv[i]));
        uint32_t stop;
        if (P.mcc <= 0) stop = mlen;
        else if ((uint32_t)P.mcc > R) stop = 0;
        else {   // the mcc-th largest reach: take the largest away mcc - 1 times (entries taken away or beyond R count as 0, the smallest value)
            uint32_t t4[4] = {rc[0], (uint32_t)lane + 64u < R ? rc[1] : 0u, (uint32_t)lane + 128u < R ? rc[2] : 0u, (uint32_t)lane + 192u < R ? rc[3] : 0u};
            if ((uint32_t)lane >= R) t4[0] = 0;
            for (int k = 1; k < P.mcc; k++) {
                uint32_t m = t4[0] > t4[1] ? t4[0] : t4[1]; { const uint32_t m2 = t4[2] > t4[3] ? t4[2] : t4[3]; if (m2 > m) m = m2; }
                const uint32_t wm = d_wave_max(m);
                if (wm == 0) break;
                const unsigned long long holders = __ballot(m == wm);
                if (lane == __builtin_ctzll(holders)) { if (t4[0] == wm) t4[0] = 0; else if (t4[1] == wm) t4[1] = 0; else if (t4[2] == wm) t4[2] = 0; else t4[3] = 0; }
            }
            uint32_t m = t4[0] > t4[1] ? t4[0] : t4[1]; { const uint32_t m2 = t4[2] > t4[3] ? t4[2] : t4[3]; if (m2 > m) m = m2; }
            stop = d_wave_max(m);
        }
        if (lane == 0) s_wr[r] = (words << 16) | stop;
    }
    __syncthreads();
    SJ_T(3);   // summaries of all clusters
    // ---- filters and the best cluster (max by (count, length), ties: the last in list order = the largest key) ----------------
    uint32_t npl, npb, win = 0xFFFFFFFFu, win_words = 0, win_stop = 0;
    {
        const bool rep = (uint32_t)tid < NL && s_label[tid] == (uint32_t)tid;
        const uint32_t sm = rep ? s_wr[tid] : 0u, words = sm >> 16, stop = sm & 0xFFFFu;
        const bool pl = rep && (int32_t)stop >= P.minlen, pb = pl && (int32_t)words >= P.mincount;
        const uint32_t cnt = sj_block_sum((pl ? 1u : 0u) | (pb ? 1u << 12 : 0u), s_red);
        npl = cnt & 0xFFFu; npb = cnt >> 12;
        if (npb) {   // block-uniform
            const uint32_t pack = pb ? ((words << 9) | (stop << 1) | (s_hasm[tid] ? 1u : 0u)) + 1u : 0u;
            const uint32_t top = sj_block_max(pack, s_red);
            const uint32_t ntop = sj_block_sum(pack == top ? 1u : 0u, s_red);
            if (ntop > 1) {
                // several clusters with the winning (count, length) and the same kind (singleton / component): the reference
                // keeps the last of them in its list order, i.e. the one whose first leaf comes last in the DFS (quirk Q12)
                __syncthreads();
                if ((uint32_t)tid < NL) s_wr[tid] = 0xFFFFFFFFu;
#ifdef SJ_STAMP
                if (tid == 0) atomicAdd(&g_sj_cyc[15], 1ull);
#endif
                dfs_ranks();
                if ((uint32_t)tid < NL) atomicMin(&s_wr[s_label[tid]], (uint32_t)s_rank[tid]);
                __syncthreads();
                win = sj_block_max(pack == top ? ((s_wr[tid] << 8) | (uint32_t)tid) + 1u : 0u, s_red);
                win = (win - 1u) & 0xFFu;
            } else win = sj_block_max(pack == top ? (uint32_t)tid + 1u : 0u, s_red) - 1u;
            win_words = (top - 1u) >> 9; win_stop = ((top - 1u) >> 1) & 0xFFu;
        }
    }
    SJ_T(9);   // the choice (with the DFS order when clusters tie)
    // ---- vote of the winning cluster --------------------------------------------------------------------------------------
    // Word v votes with q_v[d] under v[d] at every depth d < min(reach(v), length).  A thread adds up ONE depth over one part
    // of the words (a consensus of up to 64 characters: four parts; up to 128: two), eight words per step: their reach and
    // offsets come together from LDS, then the eight characters (LDS) and the eight qualities (memory: a wave reads 64
    // consecutive bytes of a word) are all fetched whether they count or not.  A, C, G, T go into two 64-bit registers of four
    // 16-bit fields (sum of qualities <= 256 x 255, words <= 256); the threads of the other parts hand their registers over
    // through LDS.  Any other base code (N, IUPAC) is rare: the depths that met one add those codes up in a pass of their own.
    const bool have = win != 0xFFFFFFFFu;
    const unsigned long long best_hi = ((unsigned long long)win_words << 32) | (unsigned long long)win_stop;
    if (have && win_stop) {   // block-uniform
        const uint32_t maxd = win_stop;
        unsigned long long *part = reinterpret_cast<unsigned long long *>(smem + X);
        __syncthreads();   // the DFS table (if it was needed) and the summaries are done with
        {
            uint32_t rc[4];
            reach4(win, rc, nullptr, wave);   // every wave the group of words it publishes
            const uint32_t mine = wave == 0 ? rc[0] : (wave == 1 ? rc[1] : (wave == 2 ? rc[2] : rc[3]));
            if ((uint32_t)tid < R) s_wr[tid] = mine | ((uint32_t)s_loff[tid] << 8);
        }
        if ((uint32_t)tid < 8) s_wr[R + tid] = 0;   // the eight-word steps may look past the last word
        if (tid == 0) s_other = 0;
        const uint32_t dsh = maxd <= 64 ? 6u : (maxd <= 128 ? 7u : 8u), D = 1u << dsh, nparts = SJ_THREADS >> dsh;
        const uint32_t d = (uint32_t)tid & (D - 1u), pi = (uint32_t)tid >> dsh;
        const uint32_t per = ((R + nparts - 1u) / nparts + 7u) & ~7u, vb = pi * per, ve = vb + per < R ? vb + per : R;
        // the qualities of the job's words: into LDS behind the tables when there is room (one round of 16-byte loads for the
        // whole workgroup; the loop below then waits for LDS only), else every step fetches its eight bytes from memory
        const uint8_t *gq = W.q + goff0;
        const bool q_lds = o + C <= lds_bytes;   // block-uniform
        if (q_lds) {
            uint4 *ql = reinterpret_cast<uint4 *>(smem + o);
            const uint4 *qsrc = reinterpret_cast<const uint4 *>(gq);
            for (uint32_t i = tid; i < (C >> 4); i += SJ_THREADS) ql[i] = qsrc[i];
        }
        const uint32_t qo = o;
        __syncthreads();
        SJ_T(10);   // reach of the winner
        unsigned long long qacc = 0, cacc = 0;
        uint32_t rare = 0;   // bit c: base code c (not A, C, G, T) met at this depth
        auto sums = [&](auto in_lds) {
            for (uint32_t v0 = vb; v0 < ve; v0 += 8) {
                const uint4 w0 = *reinterpret_cast<const uint4 *>(&s_wr[v0]), w1 = *reinterpret_cast<const uint4 *>(&s_wr[v0 + 4]);
                const uint32_t wr[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                uint32_t c[8], q[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    uint32_t co = ((wr[u] >> 8) << 4) + d;
                    co = co < C ? co : 0u;
                    c[u] = ch[co];
                    if constexpr (decltype(in_lds)::value) q[u] = smem[qo + co]; else q[u] = gq[co];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const bool act = (wr[u] & 0xFFu) > d && v0 + u < ve;
                    const uint32_t cc = (c[u] & 15u) | (act ? 0u : 16u);   // 16 = nothing to add
                    const bool pw2 = cc < 16u && cc && !(cc & (cc - 1));   // A C G T
                    const int shf = (__builtin_ctz(cc | 16u) & 3) << 4;
                    qacc += (unsigned long long)(pw2 ? q[u] : 0u) << shf;
                    cacc += (unsigned long long)(pw2 ? 1u : 0u) << shf;
                    rare |= (cc < 16u && !pw2) ? 1u << cc : 0u;
                }
            }
        };
        if (d < maxd) { if (q_lds) sums(std::true_type{}); else sums(std::false_type{}); }
        if (rare) atomicOr(&s_other, (int)rare);
        if (pi) { part[2 * ((pi - 1u) * D + d)] = qacc; part[2 * ((pi - 1u) * D + d) + 1] = cacc; }
        __syncthreads();
        const uint32_t other = (uint32_t)s_other;
        if (!pi && d < maxd) {
            for (uint32_t k = 1; k < nparts; k++) { qacc += part[2 * ((k - 1u) * D + d)]; cacc += part[2 * ((k - 1u) * D + d) + 1]; }
            int best = -1, nbest = 0;
            uint32_t bq = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t xq = (uint32_t)(qacc >> (16 * k)) & 0xFFFFu, xc = (uint32_t)(cacc >> (16 * k)) & 0xFFFFu;
                if (!xc) continue;
                if (best < 0 || xq > bq) { best = 1 << k; nbest = 1; bq = xq; }
                else if (xq == bq) nbest++;
            }
            for (uint32_t m = other; m; m &= m - 1u) {   // block-uniform list of the rare codes met anywhere in the job; nearly always empty
                const uint32_t cx = (uint32_t)__builtin_ctz(m);
                uint32_t xq = 0, xc = 0;
                for (uint32_t v = 0; v < R; v++) {
                    const uint32_t w = s_wr[v];
                    if ((w & 0xFFu) <= d) continue;
                    const uint32_t co = ((w >> 8) << 4) + d;
                    if (((uint32_t)ch[co] & 15u) == cx) { xq += q_lds ? (uint32_t)smem[qo + co] : (uint32_t)gq[co]; xc++; }
                }
                if (!xc) continue;
                if (best < 0 || xq > bq) { best = (int)cx; nbest = 1; bq = xq; }
                else if (xq == bq) nbest++;
            }
            bestc[d] = nbest > 1 ? 'N' : d_code_char(best < 0 ? 15 : best);   // get_best_qual_chars flanktrie.py:344-353
        }
        __syncthreads();
    }
    SJ_T(8);   // vote over all clusters
#ifdef SJ_STAMP
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 12; i++) if (sj_acc[i]) atomicAdd(&g_sj_cyc[i], sj_acc[i]);
    }
    if (tid == 0) { atomicAdd(&g_sj_cyc[12], 1ull); atomicAdd(&g_sj_cyc[13], (unsigned long long)NL); atomicAdd(&g_sj_cyc[14], (unsigned long long)npl | ((unsigned long long)R << 32)); }
#endif
    if (tid == 0) {
        io.n_pass_len[job] = npl; io.n_pass_both[job] = npb;
        if (have) { io.cons_count[job] = (int32_t)(best_hi >> 32); io.cons_len[job] = (int32_t)(best_hi & 0xFFFFFFFFull); }
        J.Rbig[job] = 0;
    }
    if (have) {
        const int len = (int)(best_hi & 0xFFFFFFFFull), side = (int)(job & 1);
        char *dst = io.cons_chars + (size_t)job * io.cons_stride;
        for (int d = tid; d < len; d += SJ_THREADS) dst[side == 0 ? d : len - 1 - d] = (char)bestc[d];   // 3p consensus is reversed back (findflanks.py:894-895)
    }
#undef SJ_FAIL
}

// A launch whose static + dynamic LDS passes the default 64 KB per workgroup needs the kernel's limit raised first; what the
// device cannot give (160 KB per CU on gfx950) is an error of the call with a message, not a failed launch.
static int ensure_lds(mfk_ctx *ctx, const void *func, size_t dynamic_bytes, const char *name) {
    // the static part of a kernel never changes: asked once per kernel (two kernels use this), not on every call
    static const void *known[4] = {nullptr, nullptr, nullptr, nullptr};
    static size_t known_static[4] = {0, 0, 0, 0};
    hipFuncAttributes fa;
    int slot = -1;
    for (int i = 0; i < 4; i++) if (known[i] == func) { slot = i; break; }
    if (slot >= 0) fa.sharedSizeBytes = known_static[slot];
    else {
        HIPCHK(hipFuncGetAttributes(&fa, func));
        for (int i = 0; i < 4; i++) if (!known[i]) { known_static[i] = fa.sharedSizeBytes; known[i] = func; break; }
    }
    const size_t need = fa.sharedSizeBytes + dynamic_bytes;
    if (need <= 64 * 1024) return 0;
    int optin = 0;
    if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, ctx->device) != hipSuccess || optin <= 0) optin = 160 * 1024;
    if (need > (size_t)optin || hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynamic_bytes) != hipSuccess) {
        (void)hipGetLastError();
        snprintf(ctx->err, sizeof ctx->err, "%s needs %zu bytes of LDS per workgroup (%zu static + %zu for the longest soft clip): more than this device allows", name, need, (size_t)fa.sharedSizeBytes, dynamic_bytes);
        return 6;
    }
    return 0;
}

// =====================================================================================================================
int run_consensus(mfk_ctx *ctx, const DevReads &R, const DevParams &P, ConsensusIO &io) {
    hipStream_t st = ctx->stream;
    const uint32_t E = io.n_events, n_act = io.n_act;
    io.cons_chars = nullptr; io.cons_stride = 0;
    if (!E || !n_act) return 0;
    // words of the active sites: job sizes -> word ranges -> ids.  The word tables are sized by the event count (an upper
    // bound), so the host reads the sizes it needs (words, adjacency words, characters, longest word) back in ONE batch.
    uint32_t *wid = aalloc<uint32_t>(ctx, (size_t)E + 1);   // word of a word-starting event (active sites only)
    CMeta *meta = nullptr;
    Words W; memset(&W, 0, sizeof W);
    Jobs J; memset(&J, 0, sizeof J);
    J.n = 2 * n_act;
    J.w0 = aalloc<uint32_t>(ctx, (size_t)J.n + 1); J.R = aalloc<uint32_t>(ctx, J.n); J.nl = aalloc<uint32_t>(ctx, J.n);
    J.adj_words = aalloc<uint32_t>(ctx, J.n); J.adj_off = aalloc<uint64_t>(ctx, (size_t)J.n + 1); J.best = aalloc<uint32_t>(ctx, J.n); J.giant = aalloc<uint32_t>(ctx, J.n); J.longest = aalloc<uint32_t>(ctx, 4 * (size_t)J.n);
    J.Rbig = aalloc<uint32_t>(ctx, J.n);
    W.first = aalloc<int32_t>(ctx, E); W.job = aalloc<uint32_t>(ctx, E); W.len = aalloc<uint32_t>(ctx, E); W.src = aalloc<WordSrc>(ctx, E);
    W.off = aalloc<uint64_t>(ctx, (size_t)E + 1); W.is_leaf = aalloc<uint8_t>(ctx, E); W.leaf_word = aalloc<uint32_t>(ctx, E);
    W.cl_count = aalloc<int32_t>(ctx, E); W.cl_len = aalloc<int32_t>(ctx, E); W.cl_key = aalloc<uint32_t>(ctx, E);
    W.leaf_list = aalloc<uint32_t>(ctx, E); W.rep_list = aalloc<uint32_t>(ctx, E); W.share = aalloc<uint32_t>(ctx, E); W.out_list = aalloc<uint32_t>(ctx, E);
    {
        FillGroup z(ctx, 0), f(ctx, 0xFF);
        z.want(&meta, 1); z.want(&W.nextra, E); z.want(&W.plen, E); z.want(&W.leaf_rank, E);
        z.want(&W.has_match, E); z.want(&W.n_list, 8);
        f.want(&W.second, E); f.want(&W.label, E); f.want(&W.cover, E);
        if (!wid || !J.w0 || !J.R || !J.nl || !J.adj_words || !J.adj_off || !J.best || !J.giant || !J.longest || !J.Rbig || !W.first || !W.job || !W.len || !W.src || !W.off ||
            !W.is_leaf || !W.leaf_word || !W.cl_count || !W.cl_len || !W.cl_key || !W.leaf_list || !W.rep_list || !W.share || !W.out_list || fill_commit2(z, f)) return 1;
    }
    if (J.n <= JM_ONE_MAX) {
        PROF_LAUNCH(k_job_meta_scan, dim3(1), dim3(1024), 0, st, io, J, meta);
        KCHK();
    } else {
        PROF_LAUNCH(k_job_meta, dim3(nblocks(J.n, 256)), dim3(256), 0, st, io, J, meta);
        KCHK();
        if (int rc = dev_excl_scan32(ctx, J.R, J.w0, J.n)) return rc;            // first word of every job; J.w0[J.n] = number of words
        if (int rc = dev_excl_scan64(ctx, J.adj_words, J.adj_off, J.n)) return rc;
    }
    PROF_LAUNCH(k_word_reads, dim3(n_act), dim3(256), 0, st, R, P, io, (const uint32_t *)J.w0, wid, W);
    KCHK();
    const uint32_t *n_words_p = J.w0 + J.n;
    PROF_LAUNCH(k_word_len, dim3(nblocks(E, 256)), dim3(256), 0, st, R, W, n_words_p, meta);
    KCHK();
    if (int rc = dev_excl_scan64(ctx, W.plen, W.off, E)) return rc;   // plen is 0 beyond the last word: off[E] = all characters
    uint32_t Wn = 0; uint64_t adj_total = 0;
    CMeta hm; uint64_t total_chars = 0;
    { ReadBatch rb(ctx); rb.add(n_words_p, &Wn, 4); rb.add(J.adj_off + J.n, &adj_total, 8); rb.add(meta, &hm, sizeof hm); rb.add(W.off + E, &total_chars, 8); if (int rc = rb.flush()) return rc; }
    if (Wn == 0) return 0;
    W.n = Wn;
    const unsigned ww = nblocks(Wn, 4 * WF_WORDS);   // four waves of WF_WORDS words
    if (hm.err & ERR_READ_NUMBER) { snprintf(ctx->err, sizeof ctx->err, "READ NUMBER ERROR - ZERO OR MORE THAN TWO READS ASSOCIATED WITH QUERY NAME"); return 3; }
    if (hm.err & ERR_NO_SEQUENCE) { snprintf(ctx->err, sizeof ctx->err, "a clipped read at a surviving site has no sequence in the pool (pack with pool='clipped')"); return 2; }
    if (hm.err & ERR_WORD_TOO_LONG) { snprintf(ctx->err, sizeof ctx->err, "soft clip longer than %d bases: not supported by this build", MAX_WORD_LEN); return 6; }
    W.ch = aalloc<uint8_t>(ctx, (size_t)total_chars + 64); W.q = aalloc<uint8_t>(ctx, (size_t)total_chars + 64);
    if (total_chars >= 0xFFFFFFF0ull) { snprintf(ctx->err, sizeof ctx->err, "more than 2^32 clipped bases at the surviving sites of one call"); return 1; }
    if (!W.ch || !W.q) return 1;
    // The characters of a job that k_job_small takes are made inside it, straight from the pool (no 2 bytes per clipped base
    // written here and read back there); k_word_fill runs first only when every job goes to the device-wide kernels, and
    // afterwards for the jobs k_job_small leaves over.
    const char *ns_env0 = getenv("MFK_NO_SMALL_JOBS"), *nf_env = getenv("MFK_NO_FUSED_FILL");
    const bool no_small0 = ns_env0 && *ns_env0 && *ns_env0 != '0';
    const bool fused = !no_small0 && hm.n_small && !(nf_env && *nf_env && *nf_env != '0');
    if (!fused) {
        PROF_LAUNCH(k_word_fill, dim3(ww), dim3(256), 0, st, R, W, (const uint32_t *)nullptr);
        KCHK();
    }
    const uint32_t max_len = hm.max_len ? hm.max_len : 1;
    io.cons_stride = (max_len + 7) & ~7u;
    io.cons_chars = aalloc<char>(ctx, (size_t)J.n * io.cons_stride);
    if (!io.cons_chars) return 1;
    io.err_dev = &meta->err;   // ERR_NO_SEQUENCE can still be raised by k_word_fill: the caller reads it with the results
    // ---- small jobs: one workgroup per job, everything in LDS (k_job_small).  What it cannot take (more than SJ_MAX_R words,
    // a word beyond SJ_MAX_LEN, tables beyond 128 KB of LDS) is counted in meta->n_big and goes through the device-wide kernels.
    {
        const char *ns_env = getenv("MFK_NO_SMALL_JOBS");   // test hook: everything through the device-wide kernels
        const bool no_small = ns_env && *ns_env && *ns_env != '0';
        if (no_small || !hm.n_small) {   // (no job of 1..SJ_MAX_R words, e.g. a deep isolate: no launch, no read-back)
            hm.n_big = 1;
            J.Rbig = J.R;   // every job goes to the device-wide kernels
        } else {
            // first pass: four workgroups per CU (160 KB of LDS, 128 VGPRs per lane: four waves per SIMD): 40 KB each, static part
            // included; nearly every site-side of a 200x sample needs 16-28 KB.  Second pass, for the few that need more: 128 KB.
            const uint32_t lds1 = 128 * 1024;
            if (!ctx->sj_attr) {
                hipFuncAttributes fa;
                HIPCHK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_job_small)));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_job_small), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
                ctx->sj_static = (uint32_t)fa.sharedSizeBytes;
                ctx->sj_attr = true;
            }
            // Three rungs: SJ_WAVES workgroups per CU (nearly every site-side of a 200x sample fits), then four per CU for the
            // few that need up to 40 KB, then one per CU with 128 KB.  A job that fails on a rung for want of LDS is marked in
            // J.Rbig (bit 31) and counted in n_retry; the next rung runs only those.
            const uint32_t per_wg = (160u * 1024u / SJ_WAVES) & ~255u;
            const uint32_t rung[3] = {ctx->sj_static + 8192 < per_wg ? ((per_wg - ctx->sj_static) & ~255u) : 32 * 1024,
                                      ctx->sj_static + 8192 < 40960 ? ((40960 - ctx->sj_static) & ~255u) : 32 * 1024, lds1};
            // When even a job of max_R words of max_len characters that are ALL leaves fits the last rung, every job is done after
            // the last pass whatever the first one leaves over: the passes are queued back to back (a later pass returns at once
            // where nothing is left to retry) and the read-back between them, which only decides whether to go on, is not needed.
            // A shallow isolate (a hundred words per site-side) is that case; a 200x sample is not (its bound is above 128 KB although
            // its jobs need 16-28 KB), there the outcome of a pass is read back as before.
            bool sure = false;
            if (hm.max_R <= SJ_MAX_R && hm.max_len <= SJ_MAX_LEN) {
                const size_t Rm = hm.max_R, Lm = hm.max_len ? hm.max_len : 1, a16 = 15;
                const size_t chars = (Rm * ((Lm + a16) & ~a16) + a16) & ~a16, adjb = (Rm * ((Rm + 31) / 32) * 4 + a16) & ~a16, bestb = (Lm + 1 + a16) & ~a16;
                const size_t Mb = Rm * ((Rm + a16) & ~a16);
                size_t x1 = ((Rm * Rm + a16) & ~a16) + 16 * Lm, x2 = 3072 + 64, x3 = 4 * ((Lm + 63) / 64) * 128;
                const size_t X = x1 > x2 ? (x1 > x3 ? x1 : x3) : (x2 > x3 ? x2 : x3);
                sure = chars + 4096 <= lds1 && chars + adjb + bestb + Mb + X + Rm * sizeof(WordSrc) + 256 <= lds1;
            }
            for (int pass = 0; pass < 3; pass++) {
                if (pass == 1 && rung[1] <= rung[0]) continue;   // the first rung already is four per CU
                if (pass > 0 && !sure) {
                    if (!hm.n_retry) break;
                    HIPCHK(hipMemsetAsync(&meta->n_retry, 0, 4, st));
                }
                PROF_LAUNCH(k_job_small, dim3(J.n), dim3(SJ_THREADS), rung[pass], st, R, P, W, J, io, meta, rung[pass], pass, fused ? 1 : 0);
                KCHK();
                if (!sure) { if (int rc = read_back(ctx, meta, &hm, sizeof hm)) return rc; }
            }
            if (sure) { hm.n_big = 0; io.sure_small = true; }   // (checked with the results: meta->n_big comes back beside the error word)
        }
    }
    if (!hm.n_big) return 0;
    // ---- the device-wide kernels, for the jobs left: they see R = 0 for every job that is done ------------------------
    if (fused) {   // their words' characters and qualities
        PROF_LAUNCH(k_word_fill, dim3(ww), dim3(256), 0, st, R, W, (const uint32_t *)J.Rbig);
        KCHK();
    }
    J.R = J.Rbig;
    {
        FillGroup z(ctx, 0);
        z.want(&J.adj, (size_t)adj_total); z.want(&W.agg_qc, (size_t)total_chars + 8); z.want(&W.agg_e, (size_t)total_chars + 8);
        // first touched by k_leaf_compact and later: the fill runs on the third stream next to the two kernels before it
        HIPCHK(hipEventRecord(ctx->ev_fork3, st));
        HIPCHK(hipStreamWaitEvent(ctx->stream3, ctx->ev_fork3, 0));   // the arena block may be one an earlier kernel of this call has used
        ctx->fill_stream = ctx->stream3;
        const int fill_rc = z.commit();
        ctx->fill_stream = nullptr;
        if (fill_rc) return 1;
        HIPCHK(hipEventRecord(ctx->ev_fill, ctx->stream3));
    }
    W.npos = aalloc<uint32_t>(ctx, (size_t)total_chars + 8);
    if (!W.npos) return 1;
    W.sig = aalloc<uint2>(ctx, Wn);
    if (!W.sig) return 1;
    PROF_LAUNCH(k_job_longest, dim3(J.n), dim3(JL_THREADS), 0, st, W, J);
    KCHK();
    const int lf = Wn >= (1u << 18) ? LF_WORDS : (Wn >= (1u << 17) ? 4 : (Wn >= (1u << 16) ? 2 : 1));
    PROF_LAUNCH(k_leaf_flags, dim3(nblocks(Wn, 4 * (size_t)lf)), dim3(256), 0, st, W, J, lf);
    KCHK();
    // partner chunks per pattern leaf: enough 64-lane chunks for the largest job, at most 16 (larger jobs stride)
    uint32_t n_chunks = (hm.max_R + 63) / 64;
    if (n_chunks < 1) n_chunks = 1;
    if (n_chunks > 16) n_chunks = 16;
    const uint64_t my_items = (uint64_t)J.n * N_PIVOTS * n_chunks;
    if (J.n < (1u << 24)) { W.my_list = aalloc<uint32_t>(ctx, (size_t)my_items); if (!W.my_list) return 1; }   // else: k_leaf_myers enumerates the triples
    HIPCHK(hipStreamWaitEvent(st, ctx->ev_fill, 0));
    PROF_LAUNCH(k_leaf_compact, dim3(J.n), dim3(256), 0, st, W, J, n_chunks);
    KCHK();
    uint16_t *lcpm = aalloc<uint16_t>(ctx, (size_t)adj_total * 32);
    if (!lcpm) return 1;
    // per-leaf / per-cluster kernels: fixed grids over the device-side lists (the host never learns the list sizes)
    const unsigned g_leaf1 = Wn < 8192 ? Wn : 8192, g_leaf4 = Wn < 2048 ? Wn : 2048, g_rep = Wn < 1024 ? Wn : 1024;
    // Two chains that do not depend on each other follow: (A) trie nodes of the leaves -> node sums -> DFS ranks, and
    // (B) edit distances of the pivots -> components -> the remaining cross-component pairs -> components.  With a few
    // dozen jobs (a deep isolate) neither fills the GPU, so (B) runs on the side stream while (A) runs here; with
    // thousands of jobs both fill it and the overlap costs nothing.  (Timing events of a launch go to the stream of that launch.)
    hipStream_t sb = ctx->stream2;
    HIPCHK(hipEventRecord(ctx->ev_fork, st));
    HIPCHK(hipStreamWaitEvent(sb, ctx->ev_fork, 0));
    // chain (B)
    if (W.my_list) {   // one wave per block: consecutive work items land on different CUs
        const unsigned g_my0 = (unsigned)(my_items < 8192 ? (my_items ? my_items : 1) : 8192);
        PROF_LAUNCH(k_leaf_myers, dim3(g_my0), dim3(64), 0, sb, W, J, 0, n_chunks, io.work);
    } else {
        const unsigned g_my0 = (unsigned)(my_items / 4 + 1 < 8192 ? my_items / 4 + 1 : 8192);
        PROF_LAUNCH(k_leaf_myers, dim3(g_my0), dim3(256), 0, sb, W, J, 0, n_chunks, io.work);
    }
    KCHK();
    PROF_LAUNCH(k_components, dim3(J.n), dim3(256), 0, sb, W, J, 0);
    KCHK();
    PROF_LAUNCH(k_leaf_myers, dim3(g_leaf4), dim3(256), 0, sb, W, J, 1, n_chunks, io.work);
    KCHK();
    PROF_LAUNCH(k_components, dim3(J.n), dim3(256), 0, sb, W, J, 1);
    KCHK();
    // chain (A)
    PROF_LAUNCH(k_leaf_first, dim3(g_leaf1), dim3(64), 0, st, W, J, lcpm);
    KCHK();
    // the DFS ranks need only what k_leaf_first wrote (LCP rows, trie nodes), not the node sums: third stream
    hipStream_t sc = ctx->stream3;
    HIPCHK(hipEventRecord(ctx->ev_fork3, st));
    HIPCHK(hipStreamWaitEvent(sc, ctx->ev_fork3, 0));
    if (hm.max_R <= 256) { PROF_LAUNCH(k_leaf_rank_wave, dim3(g_leaf4), dim3(256), 0, sc, W, J, (const uint16_t *)lcpm); }
    else { PROF_LAUNCH(k_leaf_rank, dim3(g_leaf4), dim3(256), 0, sc, W, J, (const uint16_t *)lcpm); }
    KCHK();
    HIPCHK(hipEventRecord(ctx->ev_join3, sc));
    {
        const size_t ns_lds = (size_t)max_len * NS_SLOTS * 16;   // 64 KB with MAX_WORD_LEN 512
        if (int rc = ensure_lds(ctx, reinterpret_cast<const void *>(&k_node_sums), ns_lds, "k_node_sums")) return rc;
        PROF_LAUNCH(k_node_sums, dim3(nblocks(Wn, NS_WORDS)), dim3(1024), ns_lds, st, W, J, max_len);
        KCHK();
    }
    HIPCHK(hipEventRecord(ctx->ev_join, sb));
    HIPCHK(hipStreamWaitEvent(st, ctx->ev_join, 0));
    HIPCHK(hipStreamWaitEvent(st, ctx->ev_join3, 0));
    size_t lds = (size_t)max_len * 16 * 8;
    if (int rc = ensure_lds(ctx, reinterpret_cast<const void *>(&k_cluster_vote), lds, "k_cluster_vote")) return rc;   // 64 KB + 4 KB static at 512 bases
    char *cl_chars = aalloc<char>(ctx, (size_t)Wn * io.cons_stride);
    if (!cl_chars) return 1;
    PROF_LAUNCH(k_leaf_share, dim3(g_leaf4), dim3(256), 0, st, W, J, (const uint16_t *)lcpm);
    KCHK();
    PROF_LAUNCH(k_cluster_vote, dim3(g_rep), dim3(CV_THREADS), lds, st, P, W, J, max_len, cl_chars, io.cons_stride);
    KCHK();
    PROF_LAUNCH(k_job_best, dim3(J.n), dim3(256), 0, st, P, W, J, io, (const char *)cl_chars, io.cons_stride);
    KCHK();
    return 0;
}

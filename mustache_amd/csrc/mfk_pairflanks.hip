// mfk_pairflanks.hip -- the pairflanks pipeline of libmfk.so (replaces FlankPairer.run_pair_flanks,
// pairflanks.py:96-122).  Kernels, in call order:
//
//   k_pair_window<count/fill>  pair_all_nearby_flanks (pairflanks.py:138-169): for each 5p flank the 3p flanks of the same
//                              contig with pos5-maxdr-1 <= pos3 < pos5, in file order (binary-searched window when the
//                              table is sorted, full scan otherwise)
//   k_inverted_repeat          check_pairs_for_ir (172-219): one wave per candidate; local alignment of the combined
//                              sequence against its reverse complement (einverted's model, match 3 / mismatch -4 / gap 12 /
//                              threshold 15) with the score matrix in LDS, processed span by span (parity unpinned: the
//                              reference shells out to EMBOSS einverted; DESIGN.md "IR definition")
//   radix sort + k_assign      assign_pairs (222-250): stable order by (IR_length desc, diffcount asc, difflength asc),
//                              then the greedy exclusion resolved in parallel rounds (a pair is kept when it is the
//                              best undecided pair of both its flanks)
//   k_spanning                 count_insertion_spanning_reads (370-391): one workgroup per kept pair
//   k_junction_filter          filter_junction_spanning (421-431), fp64
//   k_direct_repeats           get_direct_repeats (277-367): reference slice + quality-weighted pileup consensus over the
//                              50-bp window in LDS
#include "mfk_host.h"

#include <algorithm>
#include <stdlib.h>

#define IR_MAXLEN 100      // 40 + 20 + 40 (truncated_flank_length, pairflanks.py:78, 183-186)
#define IR_MAXREP 8
#define IR_CM 168          // padded row of column codes: a lane reaches index 127 + 35 + 1
#define IR_S (128 * IR_MAXLEN + 128 + 64)   // score matrix + room for the rows 64..127 of lanes beyond the matrix
#define PW_HOST_MAX 256    // up to this many flanks the candidate pairs are listed on the host (mfk_pairflanks)

struct FlankDev {
    uint32_t n;
    const int32_t *contig, *pos, *orient, *ccount, *total, *seq_len;
    const int64_t *seq_off;
    const char *seqs;
};

// ---- pair_all_nearby_flanks ------------------------------------------------------------------------------------------
template <bool FILL>
__global__ void k_pair_window(FlankDev F, int maxdr, int sorted, uint32_t *count, const uint32_t *offset, uint32_t *c5, uint32_t *c3) {
    uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= F.n) return;
    uint32_t c = 0;
    if (F.orient[a] == 5) {
        const int32_t contig = F.contig[a], pos5 = F.pos[a];
        const int64_t min_pos = (int64_t)pos5 - maxdr - 1;
        uint32_t lo = 0, hi = F.n;
        if (sorted) {   // rows ordered by (contig, pos): only [first row >= (contig, min_pos), first row >= (contig, pos5)) can match
            uint32_t l = 0, h = F.n;
            while (l < h) { uint32_t m = (l + h) >> 1; if (F.contig[m] < contig || (F.contig[m] == contig && (int64_t)F.pos[m] < min_pos)) l = m + 1; else h = m; }
            lo = l; h = F.n;
            while (l < h) { uint32_t m = (l + h) >> 1; if (F.contig[m] < contig || (F.contig[m] == contig && F.pos[m] < pos5)) l = m + 1; else h = m; }
            hi = l;
        }
        uint32_t o = FILL ? offset[a] : 0;
        for (uint32_t b = lo; b < hi; b++) {
            if (F.orient[b] != 3 || F.contig[b] != contig || (int64_t)F.pos[b] < min_pos || F.pos[b] >= pos5) continue;
            if (FILL) { c5[o + c] = a; c3[o + c] = b; }
            c++;
        }
    }
    if (!FILL) count[a] = c;
}

// mfk_flankpairs hand-over: the flank rows k_rows_fill left on the device -> the columns the kernels here read.  The table is
// sized by an upper bound (two rows per active site); rows behind the last one sort behind every contig and are neither 5p nor
// 3p, so no search lands on them and no candidate holds one.
__global__ void k_flank_cols(const mfk_flank *rows, const uint64_t *rows_total, uint32_t cap, int32_t *tab, int64_t *soff) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const uint32_t nf = (uint32_t)(*rows_total >> 24);
    if (i < nf) {
        const mfk_flank f = rows[i];
        tab[i] = f.contig_id; tab[cap + i] = f.pos; tab[2 * (size_t)cap + i] = f.orient; tab[3 * (size_t)cap + i] = f.consensus_softclip_count;
        tab[4 * (size_t)cap + i] = f.total_count; tab[5 * (size_t)cap + i] = f.seq_len; soff[i] = f.seq_off;
    } else {
        tab[i] = 0x7FFFFFFF; tab[cap + i] = 0x7FFFFFFF; tab[2 * (size_t)cap + i] = 0; tab[3 * (size_t)cap + i] = 0;
        tab[4 * (size_t)cap + i] = 0; tab[5 * (size_t)cap + i] = 0; soff[i] = 0;
    }
}

// ---- inverted repeats ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int d_ir_code(char ch) {
    switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

// one wave per sequence.  s: LDS copy of the sequence codes; S: LDS L x L score matrix, one byte per cell (a score is at most
// 3 x 50 pairs of a 100-base sequence and never negative), so that eleven workgroups share a CU instead of six.
// out[6]: has_ir, arm1 start/end, arm2 start/end (1-based inclusive), number of repeats reported
// early: stop as soon as no ACCEPTED repeat can follow (pairflanks needs has_IR and the arms, not the number of repeats): a
// repeat is accepted only if its outer cell (arm1 start, arm2 end) lies in the corner i <= 14, j >= L - 16 (pairflanks.py:253-266),
// a repeat is reported only with a score >= 15 in that cell, and masking more positions can only lower a score (the recurrence
// is monotone).  So once every corner cell of a pass is below the threshold, this pass and all later ones report nothing that
// the acceptance loop would take, and has / arms are final.  Pairs without a terminal repeat then cost one pass instead of up to
// eight (chance repeats of five pairs elsewhere in 100 bases are common).  out[5] is then the number found so far.
__device__ void d_ir_detect(const uint8_t *code, int L, uint8_t *S, uint8_t *mask, int32_t *rep /* 4*IR_MAXREP, LDS */, int32_t *out, bool early = false) {
    const int MATCH = 3, MISMATCH = -4, GAP = 12, THRESHOLD = 15;
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < L; i += 64) mask[i] = 0;
    __syncthreads();
    int nrep = 0;
    // Anti-diagonal wavefront held in registers: lane owns rows iA = lane and iB = lane + 64 (L <= 100) and keeps its
    // cells of the last two diagonals packed in one register (scores fit 16 bits); the cells of row i+1 come from the
    // next lane through a DPP wave shift, row 64 (lane 0's second slot) reaches lane 63 through a scalar broadcast.  S is
    // still written to LDS, but only the traceback reads it; the column characters are fetched one diagonal ahead, so a
    // diagonal is a short chain of VALU instructions instead of three LDS round trips and a barrier.
    const int iA = lane, iB = lane + 64;
    uint8_t *cm = mask + IR_MAXLEN + 4;   // code | masked << 3, rebuilt every pass (the caller's mask array has room for IR_CM entries)
    // Row A's part of a diagonal is branch-free: cm is padded with "masked N" up to the largest index a lane can reach, so a
    // cell beyond the matrix scores 0 like any masked cell, its store lands in the unused lower triangle of S (row A never
    // leaves the array), and the best cell of a row is kept as (score, diagonal) instead of a packed key per cell.
    for (int i = L + lane; i < IR_CM; i += 64) cm[i] = 12;
    while (nrep < IR_MAXREP) {
        for (int i = lane; i < L; i += 64) { S[i * L + i] = 0; cm[i] = (uint8_t)(code[i] | (mask[i] ? 8 : 0)); }
        __syncthreads();
        const int cA = cm[iA], cB = cm[iB];
        const int wantA = cA < 4 ? 3 - cA : 16, wantB = cB < 4 ? 3 - cB : 16;   // the code that pairs with the row's base (none for N)
        // The column character of a row moves one position per diagonal: lane i's next one is lane i + 1's current one, so it
        // travels with the scores in the same DPP shift (bits 8..11 of the packed pair; a score is below 256) instead of
        // coming from LDS every diagonal (the load's latency, not the arithmetic, set the pace of a diagonal).
        int cjA = cm[iA + 1], cjB = cm[iB + 1];
        unsigned pA = 0, pB = 0;               // S(i, i+d-1) | S(i, i+d-2) << 16
        int bestvA = 0, bestdA = 0, bestvB = 0, bestdB = 0;   // per row: highest score and the first diagonal that reached it (= smallest j)
        const int nB = L - 64;  // row B has cells only on diagonals d < L - 64
        uint8_t *sA = S + iA * L + iA;         // S(iA, iA + d) = sA[d]
        uint8_t *sB = S + iB * L + iB;         // rows beyond the matrix write behind it: S has IR_S entries
        for (int d = 1; d < L; d++) {
            const unsigned xA = pA | ((unsigned)cjA << 8), xB = pB | ((unsigned)cjB << 8);
            unsigned gA = (unsigned)__builtin_amdgcn_update_dpp(0, (int)xA, 0x130 /* wave_shl:1: lane i <- lane i+1 */, 0xF, 0xF, false);
            const unsigned fromB = (unsigned)__builtin_amdgcn_readfirstlane((int)xB);
            if (lane == 63) gA = fromB;
            const unsigned nA = gA & 0x00FF00FFu;
            const int ncjA = (int)((gA >> 8) & 15u);         // next diagonal's column character
            const int m = cjA == wantA ? MATCH : MISMATCH;   // a masked column (bit 3) never equals wantA
            int vA = (int)(nA >> 16) + m;                    // S(i+1, j-1) is 0 on the first two diagonals by construction
            const int up = (int)(nA & 0xFFFF) - GAP, left = (int)(pA & 0xFFFF) - GAP;
            vA = max(max(vA, up), max(left, 0));
            if ((cA | cjA) & 8) vA = 0;
            sA[d] = (uint8_t)vA;
            if (vA > bestvA) { bestvA = vA; bestdA = d; }
            pA = (pA << 16) | (unsigned)vA;
            cjA = ncjA;
            if (d < nB) {   // wave-uniform: rows 64.. have cells only on the first L - 64 diagonals; same branch-free step
                unsigned gB = (unsigned)__builtin_amdgcn_update_dpp(0, (int)xB, 0x130, 0xF, 0xF, false);
                if (lane == 63) gB = 12u << 8;               // behind the last row: masked N
                const unsigned nBv = gB & 0x00FF00FFu;
                const int ncjB = (int)((gB >> 8) & 15u);
                const int mb = cjB == wantB ? MATCH : MISMATCH;
                int vB = (int)(nBv >> 16) + mb;
                const int upb = (int)(nBv & 0xFFFF) - GAP, leftb = (int)(pB & 0xFFFF) - GAP;
                vB = max(max(vB, upb), max(leftb, 0));
                if ((cB | cjB) & 8) vB = 0;
                sB[d] = (uint8_t)vB;
                if (vB > bestvB) { bestvB = vB; bestdB = d; }
                pB = (pB << 16) | (unsigned)vB;
                cjB = ncjB;
            } else cjB = 12;   // columns behind the sequence: masked N (row A's lane 63 takes its next character from here)
        }
        unsigned bestkey = 0;   // score << 16 | (255 - j) << 8 | i : highest score, then smallest j, then largest i
        if (bestvB > 0) bestkey = ((unsigned)bestvB << 16) | ((unsigned)(255 - (iB + bestdB)) << 8) | (unsigned)iB;
        if (bestvA > 0) {
            const unsigned keyA = ((unsigned)bestvA << 16) | ((unsigned)(255 - (iA + bestdA)) << 8) | (unsigned)iA;
            if (keyA > bestkey) bestkey = keyA;
        }
        __syncthreads();
        for (int o = 32; o > 0; o >>= 1) { unsigned other = __shfl_xor(bestkey, o, 64); if (other > bestkey) bestkey = other; }
        int best = (int)(bestkey >> 16);
        if (best < THRESHOLD) break;
        if (early) {
            int cmax = 0;
            for (int t = lane; t < 15 * 16; t += 64) {
                const int i = t >> 4, j = L - 16 + (t & 15);
                if (i < L && j > i) { const int v = S[i * L + j]; cmax = v > cmax ? v : cmax; }
            }
            for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(cmax, o, 64); cmax = other > cmax ? other : cmax; }
            if (cmax < THRESHOLD) break;
        }
        if (lane == 0) {
            int bi = (int)(bestkey & 255), bj = 255 - (int)((bestkey >> 8) & 255);
            int i = bi, j = bj, ii = bi, jj = bj;
            while (i < j && S[i * L + j] > 0) {
                int diag = (j - i >= 2) ? S[(i + 1) * L + (j - 1)] : 0;
                int m = (code[i] < 4 && code[j] < 4 && code[i] + code[j] == 3) ? MATCH : MISMATCH;
                ii = i; jj = j;
                if (S[i * L + j] == diag + m) { i++; j--; }
                else if (S[i * L + j] == S[i * L + (j - 1)] - GAP) j--;
                else i++;
            }
            rep[4 * nrep] = bi; rep[4 * nrep + 1] = ii; rep[4 * nrep + 2] = jj; rep[4 * nrep + 3] = bj;
            for (int k = bi; k <= ii; k++) mask[k] = 1;
            for (int k = jj; k <= bj; k++) mask[k] = 1;
        }
        nrep++;
        __syncthreads();
    }
    if (lane == 0) {
        // report order: outer end ascending, then arm1 start descending (scan order of the maxima)
        for (int a = 1; a < nrep; a++) {
            int32_t t0 = rep[4 * a], t1 = rep[4 * a + 1], t2 = rep[4 * a + 2], t3 = rep[4 * a + 3];
            int b = a - 1;
            while (b >= 0 && (rep[4 * b + 3] > t3 || (rep[4 * b + 3] == t3 && rep[4 * b] < t0))) {
                rep[4 * b + 4] = rep[4 * b]; rep[4 * b + 5] = rep[4 * b + 1]; rep[4 * b + 6] = rep[4 * b + 2]; rep[4 * b + 7] = rep[4 * b + 3];
                b--;
            }
            rep[4 * b + 4] = t0; rep[4 * b + 5] = t1; rep[4 * b + 6] = t2; rep[4 * b + 7] = t3;
        }
        // the reference's acceptance loop (pairflanks.py:196-205, pair_has_ir 253-266): the last accepted repeat wins
        int has = 0, k1s = 0, k1e = 0, k2s = 0, k2e = 0;
        for (int k = 0; k < nrep; k++) {
            int ir1_pos_5p = rep[4 * k] + 1, ir2_pos_3p = rep[4 * k + 3] + 1;
            if (ir1_pos_5p <= 15 && ir2_pos_3p >= L - 15) { has = 1; k1s = rep[4 * k] + 1; k1e = rep[4 * k + 1] + 1; k2s = rep[4 * k + 2] + 1; k2e = rep[4 * k + 3] + 1; }
        }
        out[0] = has; out[1] = k1s; out[2] = k1e; out[3] = k2s; out[4] = k2e; out[5] = nrep;
    }
    __syncthreads();
}

// generic entry: sequences given explicitly (mfk_ir_detect)
__global__ __launch_bounds__(64) void k_ir_seqs(const char *seqs, const int64_t *off, uint32_t n, int32_t *out) {
    __shared__ uint8_t S[IR_S];
    __shared__ uint8_t code[IR_MAXLEN + 4], mask[IR_MAXLEN + 4 + IR_CM];   // mask | cm (d_ir_detect)
    __shared__ int32_t rep[4 * IR_MAXREP + 8];
    uint32_t k = blockIdx.x;
    if (k >= n) return;
    int L = (int)(off[k + 1] - off[k]);
    if (L > IR_MAXLEN) L = IR_MAXLEN;
    for (int i = threadIdx.x; i < L; i += 64) code[i] = (uint8_t)d_ir_code(seqs[off[k] + i]);
    __syncthreads();
    d_ir_detect(code, L, S, mask, rep, out + 6 * (size_t)k);
}

// candidate pairs: combined = seq5[:40] + 'N'*20 + seq3[-40:]  (pairflanks.py:183-186, truncate_sequence 268-275)
__device__ __forceinline__ uint64_t d_assign_key(const FlankDev &F, uint32_t a, uint32_t b, const int32_t *ir) {
    int irlen = ir[1] ? ir[4] - ir[3] + 1 : 0;      // IR_length = len(IR_5p) = arm2 (pairflanks.py:212-213, 225)
    int64_t dc = (int64_t)F.ccount[a] - (int64_t)F.ccount[b]; if (dc < 0) dc = -dc;   // diffcount (226)
    int dl = F.seq_len[a] - F.seq_len[b]; if (dl < 0) dl = -dl;             // difflength (227)
    if (dl > 65535) dl = 65535;
    return ((uint64_t)(127 - irlen) << 48) | ((uint64_t)(dc & 0xFFFFFFFFll) << 16) | (uint64_t)dl;
}

// also writes the candidate's sort key for assign_pairs (stable order by IR_length desc, diffcount asc, difflength asc)
__global__ __launch_bounds__(64) void k_inverted_repeat(FlankDev F, const uint32_t *c5, const uint32_t *c3, uint32_t n, int32_t *out, uint64_t *keys, uint32_t *vals) {
    __shared__ uint8_t S[IR_S];
    __shared__ uint8_t code[IR_MAXLEN + 4], mask[IR_MAXLEN + 4 + IR_CM];   // mask | cm (d_ir_detect)
    __shared__ int32_t rep[4 * IR_MAXREP + 8];
    uint32_t k = blockIdx.x;
    if (k >= n) return;
    uint32_t a = c5[k], b = c3[k];
    int l5 = F.seq_len[a], l3 = F.seq_len[b];
    int n5 = l5 > 40 ? 40 : l5, n3 = l3 > 40 ? 40 : l3;
    int L = n5 + 20 + n3;
    const char *s5 = F.seqs + F.seq_off[a], *s3 = F.seqs + F.seq_off[b] + (l3 - n3);
    for (int i = threadIdx.x; i < L; i += 64) {
        char ch = i < n5 ? s5[i] : (i < n5 + 20 ? 'N' : s3[i - n5 - 20]);
        code[i] = (uint8_t)d_ir_code(ch);
    }
    __syncthreads();
    d_ir_detect(code, L, S, mask, rep, out + 6 * (size_t)k, true);
    if (threadIdx.x == 0) { keys[k] = d_assign_key(F, a, b, out + 6 * (size_t)k); vals[k] = k; }
}

// external inverted-repeat table (mfk_pairflanks_set_ir, e.g. from EMBOSS einverted): rows i5 i3 has a1s a1e a2s a2e
// sorted by (i5, i3); a candidate without a row raises the error word
__global__ void k_ir_lookup(FlankDev F, const uint32_t *c5, const uint32_t *c3, uint32_t n, const int32_t *tab, uint32_t n_tab, int32_t *out, uint64_t *keys, uint32_t *vals, uint32_t *err) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int32_t a = (int32_t)c5[k], b = (int32_t)c3[k];
    uint32_t lo = 0, hi = n_tab;
    while (lo < hi) { uint32_t m = (lo + hi) >> 1; const int32_t *r = tab + 7 * (size_t)m; if (r[0] < a || (r[0] == a && r[1] < b)) lo = m + 1; else hi = m; }
    int32_t *o = out + 6 * (size_t)k;
    if (lo < n_tab && tab[7 * (size_t)lo] == a && tab[7 * (size_t)lo + 1] == b) { for (int z = 0; z < 5; z++) o[z] = tab[7 * (size_t)lo + 2 + z]; o[5] = -1; }
    else { for (int z = 0; z < 6; z++) o[z] = 0; atomicOr(err, 1u); }
    keys[k] = d_assign_key(F, (uint32_t)a, (uint32_t)b, o); vals[k] = k;
}

// ---- assign_pairs ------------------------------------------------------------------------------------------------------
// greedy exclusion (pairflanks.py:233-245) in parallel rounds, one workgroup.  order[p] = candidate at sorted position p.
__global__ __launch_bounds__(1024) void k_assign(const uint32_t *order, const uint32_t *c5, const uint32_t *c3, uint32_t n, uint32_t nF,
                                                 uint32_t *best5, uint32_t *best3, uint8_t *used5, uint8_t *used3, uint8_t *state /* by sorted position */, uint32_t *kflag) {
    __shared__ int undecided;
    for (uint32_t i = threadIdx.x; i < nF; i += blockDim.x) { best5[i] = 0xFFFFFFFFu; best3[i] = 0xFFFFFFFFu; used5[i] = 0; used3[i] = 0; }
    for (uint32_t p = threadIdx.x; p < n; p += blockDim.x) state[p] = 0;
    __syncthreads();
    for (uint32_t round = 0; round <= n; round++) {
        if (threadIdx.x == 0) undecided = 0;
        __syncthreads();
        for (uint32_t p = threadIdx.x; p < n; p += blockDim.x) {
            if (state[p]) continue;
            uint32_t c = order[p], a = c5[c], b = c3[c];
            if (used5[a] || used3[b]) { state[p] = 2; continue; }
            atomicMin(&best5[a], p); atomicMin(&best3[b], p);
            undecided = 1;
        }
        __syncthreads();
        if (!undecided) break;
        for (uint32_t p = threadIdx.x; p < n; p += blockDim.x) {
            if (state[p]) continue;
            uint32_t c = order[p], a = c5[c], b = c3[c];
            if (best5[a] == p && best3[b] == p) { state[p] = 1; used5[a] = 1; used3[b] = 1; }
        }
        __syncthreads();
        for (uint32_t p = threadIdx.x; p < n; p += blockDim.x) {
            if (state[p]) continue;
            uint32_t c = order[p];
            best5[c5[c]] = 0xFFFFFFFFu; best3[c3[c]] = 0xFFFFFFFFu;
        }
        __syncthreads();
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < n; p += blockDim.x) kflag[p] = state[p] == 1 ? 1u : 0u;
}

// assign_pairs without a device-wide sort and without one workgroup walking every candidate (thousands of candidates, flank
// table sorted by (contig, position)).  The greedy exclusion (pairflanks.py:233-245) couples two candidates only through a
// shared flank, and such candidates are neighbours in the candidate list: the candidates of a 5p flank are one stretch of it (3p
// rows ascending), the windows of successive 5p flanks are intervals whose two ends only move forward, so two 5p flanks that
// share a 3p flank share it with every 5p flank between them.  A stretch of the list therefore starts a new COMPONENT exactly
// where the 5p flank changes and its first 3p row lies behind the previous candidate's 3p row.  Inside a component the
// outcome depends only on the order of ITS candidates in the stable sort by (IR_length desc, diffcount asc, difflength asc) =
// order by (key, candidate index).  One thread per component: pick the best candidate not yet decided, keep it if both its
// flanks are free, until none is left (components are a handful of candidates; quadratic in their size).
// state[c]: 1 kept, 2 excluded; kflag[c] = kept.
__global__ void k_assign_seg(const uint64_t *keys, const uint32_t *c5, const uint32_t *c3, uint32_t n, uint8_t *used5, uint8_t *used3, uint8_t *state, uint32_t *kflag) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (p && !(c5[p] != c5[p - 1] && c3[p - 1] < c3[p])) return;   // not the first candidate of a component
    uint32_t e = p + 1;
    while (e < n && !(c5[e] != c5[e - 1] && c3[e - 1] < c3[e])) e++;
    for (uint32_t q = p; q < e; q++) state[q] = 0;
    for (uint32_t left = e - p; left; left--) {
        uint32_t best = p; uint64_t bk = ~0ull; bool have = false;
        for (uint32_t q = p; q < e; q++) {
            if (state[q]) continue;
            const uint64_t k = keys[q];
            if (!have || k < bk) { have = true; bk = k; best = q; }   // ties: the earlier candidate (stable sort)
        }
        const uint32_t a = c5[best], b = c3[best];
        if (used5[a] || used3[b]) state[best] = 2;
        else { state[best] = 1; used5[a] = 1; used3[b] = 1; }
    }
    for (uint32_t q = p; q < e; q++) kflag[q] = state[q] == 1 ? 1u : 0u;
}

// its kept pairs -> keys for sort_values(['contig','pos_5p','pos_3p']) (candidate order instead of k_assign's sorted order)
__global__ void k_kept_keys_seg(FlankDev F, const uint32_t *c5, const uint32_t *c3, const uint32_t *kflag, const uint32_t *scan, uint32_t n,
                                const int32_t *contig_rank, uint64_t *keys, uint32_t *vals) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n || !kflag[c]) return;
    const uint32_t a = c5[c], b = c3[c], o = scan[c];
    uint32_t diff = (uint32_t)(F.pos[a] - F.pos[b]);
    if (diff > 0xFFFF) diff = 0xFFFF;
    keys[o] = ((uint64_t)(uint32_t)contig_rank[F.contig[a]] << 48) | ((uint64_t)(uint32_t)(F.pos[a] + 1) << 16) | (uint64_t)(0xFFFFu - diff);
    vals[o] = c;
}

// kept pairs in sorted_pairs order -> keys for sort_values(['contig','pos_5p','pos_3p']) (pairflanks.py:248)
__global__ void k_kept_keys(FlankDev F, const uint32_t *order, const uint32_t *c5, const uint32_t *c3, const uint8_t *state, const uint32_t *scan, uint32_t n,
                            const int32_t *contig_rank, uint64_t *keys, uint32_t *vals) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n || state[p] != 1) return;
    uint32_t c = order[p], a = c5[c], b = c3[c], o = scan[p];
    uint32_t diff = (uint32_t)(F.pos[a] - F.pos[b]);
    if (diff > 0xFFFF) diff = 0xFFFF;
    keys[o] = ((uint64_t)(uint32_t)contig_rank[F.contig[a]] << 48) | ((uint64_t)(uint32_t)(F.pos[a] + 1) << 16) | (uint64_t)(0xFFFFu - diff);
    vals[o] = c;
}

// assign_pairs for up to PA_MAX candidates in ONE launch of one workgroup (a deep isolate has a few dozen candidates, and
// five launches -- two device-wide sorts among them -- cost more than the work): stable sort by the assign key (rank =
// number of candidates that sort before), the greedy exclusion rounds of k_assign, then the kept pairs ranked by
// (contig, pos_5p, pos_3p).  Same results as the sort / k_assign / scan / k_kept_keys / sort chain below.
#define PA_MAX 1024
#define PA_SMALL 256   // the host takes this path up to here: the rank sorts are quadratic, 1000 candidates are faster through rocPRIM
__global__ __launch_bounds__(PA_MAX) void k_assign_small(FlankDev F, const uint64_t *keys, const uint32_t *c5, const uint32_t *c3, uint32_t n, uint32_t nF,
                                                         const int32_t *contig_rank, uint32_t *best5, uint32_t *best3, uint8_t *used5, uint8_t *used3,
                                                         uint32_t *kept, uint32_t *n_kept) {
    __shared__ uint64_t sk[PA_MAX];
    __shared__ uint32_t order[PA_MAX], kc[PA_MAX];
    __shared__ uint8_t state[PA_MAX];
    __shared__ uint32_t sh[16];
    __shared__ int undecided;
    const uint32_t t = threadIdx.x;
    const uint64_t key = t < n ? keys[t] : ~0ull;
    sk[t] = key;
    state[t] = 0;
    for (uint32_t i = t; i < nF; i += blockDim.x) { best5[i] = 0xFFFFFFFFu; best3[i] = 0xFFFFFFFFu; used5[i] = 0; used3[i] = 0; }
    __syncthreads();
    if (t < n) {
        uint32_t rank = 0;
        for (uint32_t q = 0; q < n; q++) { const uint64_t kq = sk[q]; rank += (kq < key || (kq == key && q < t)) ? 1u : 0u; }
        order[rank] = t;
    }
    __syncthreads();
    // greedy exclusion (pairflanks.py:233-245): p = sorted position
    const uint32_t p = t;
    uint32_t a = 0, b = 0, c = 0;
    if (p < n) { c = order[p]; a = c5[c]; b = c3[c]; }
    for (uint32_t round = 0; round <= n; round++) {
        if (t == 0) undecided = 0;
        __syncthreads();
        if (p < n && !state[p]) {
            if (used5[a] || used3[b]) state[p] = 2;
            else { atomicMin(&best5[a], p); atomicMin(&best3[b], p); undecided = 1; }
        }
        __syncthreads();
        if (!undecided) break;
        if (p < n && !state[p] && best5[a] == p && best3[b] == p) { state[p] = 1; used5[a] = 1; used3[b] = 1; }
        __syncthreads();
        if (p < n && !state[p]) { best5[a] = 0xFFFFFFFFu; best3[b] = 0xFFFFFFFFu; }
        __syncthreads();
    }
    __syncthreads();
    // kept pairs in sorted_pairs order, then ranked by sort_values(['contig','pos_5p','pos_3p']) (pairflanks.py:248)
    const uint32_t flag = (p < n && state[p] == 1) ? 1u : 0u;
    uint32_t nk;
    const uint32_t o = block_excl_scan(flag, sh, &nk);
    __syncthreads();
    if (flag) {
        uint32_t diff = (uint32_t)(F.pos[a] - F.pos[b]);
        if (diff > 0xFFFF) diff = 0xFFFF;
        sk[o] = ((uint64_t)(uint32_t)contig_rank[F.contig[a]] << 48) | ((uint64_t)(uint32_t)(F.pos[a] + 1) << 16) | (uint64_t)(0xFFFFu - diff);
        kc[o] = c;
    }
    __syncthreads();
    if (t < nk) {
        const uint64_t k2 = sk[t];
        uint32_t rank = 0;
        for (uint32_t q = 0; q < nk; q++) { const uint64_t kq = sk[q]; rank += (kq < k2 || (kq == k2 && q < t)) ? 1u : 0u; }
        kept[rank] = kc[t];
    }
    if (t == 0) *n_kept = nk;
}

// k_pair_finish + scan + k_pair_final for up to PA_MAX kept pairs, one workgroup
__global__ __launch_bounds__(PA_MAX) void k_pair_finish_small(DevParams P, FlankDev F, const uint32_t *kept, const uint32_t *c5, const uint32_t *c3, const uint32_t *n_kept, const int32_t *ir,
                                                              const uint32_t *span_all, uint32_t *fin, uint32_t *fin_span, uint32_t *n_fin, uint32_t *pstats) {
    __shared__ uint32_t sh[16];
    const uint32_t k = threadIdx.x, nk = *n_kept;
    uint32_t flag = 0, c = 0, cnt = 0;
    if (k < nk) {
        c = kept[k]; cnt = span_all[c];
        const double total = (double)((int64_t)F.total[c5[c]] + (int64_t)F.total[c3[c]]) / 2;
        const double prop = (double)cnt / total;
        flag = prop < P.maxjsp ? 1u : 0u;   // filter_junction_spanning pairflanks.py:421-431 (float64, strict <, quirk P9)
        if (ir[6 * c]) atomicAdd(&pstats[2], 1u);
        if (cnt > 0) atomicAdd(&pstats[3], 1u);
    }
    uint32_t nf;
    const uint32_t o = block_excl_scan(flag, sh, &nf);
    if (flag) { fin[o] = c; fin_span[o] = cnt; }
    if (threadIdx.x == 0) *n_fin = nf;
}

// ---- count_insertion_spanning_reads -------------------------------------------------------------------------------------
__device__ __forceinline__ bool d_spans(const DevReads &R, const DevParams &P, int64_t r, uint8_t f, int32_t pos3, int32_t pos5) {
    // pairflanks.py:381-383, insertion_spanning_length = 10 (quirk P7)
    return R.ref_start[r] < pos3 - 10 && R.ref_end[r] > pos5 + 10 + 1 &&
           !(d_large_insertion_at(R, P, r, f, pos3) || d_large_insertion_at(R, P, r, f, pos5));
}

// One workgroup per CANDIDATE pair (spanning[c]): the count does not depend on the assignment, so it runs on the side
// stream while the inverted-repeat alignment and assign_pairs run on the main one; k_pair_finish picks the kept pairs'
// counts and applies filter_junction_spanning.
__global__ __launch_bounds__(1024) void k_spanning(DevReads R, DevParams P, FlankDev F, const uint32_t *c5, const uint32_t *c3, uint32_t n_cand, uint32_t *spanning) {
    __shared__ uint32_t cnt;
    __shared__ int64_t range[2];
    const uint32_t c = blockIdx.x;
    if (c >= n_cand) return;
    const int32_t contig = F.contig[c5[c]], pos5 = F.pos[c5[c]], pos3 = F.pos[c3[c]];
    const int32_t clen = R.contig_len[contig];
    int32_t start = pos3 - 1, end = pos5 + 2;
    if (start < 0) start = 0;
    if (end > clen) end = clen;
    if (threadIdx.x == 0) cnt = 0;
    if (threadIdx.x < 128) d_fetch_range_2waves(R, contig, start, end, range);   // first two waves: one bound each
    __syncthreads();
    uint32_t mine = 0;
    for (int64_t r = range[0] + threadIdx.x; r < range[1]; r += blockDim.x) {
        // the columns every test looks at, loaded together and pinned before the first branch (one round trip instead of a chain
        // of three: positions, then flags, then mapping quality); the two coordinates settle nearly every read of an insertion
        // site (its reads end AT the junction, they do not span it), the rest is looked at only for the few that are left
        int32_t rs = R.ref_start[r], re = R.ref_end[r];
        uint32_t f32 = R.flags[r], mq = R.mapq[r];
        int32_t lk = R.link[r];
        asm volatile("" : "+v"(rs), "+v"(re), "+v"(f32), "+v"(mq), "+v"(lk));
        if ((re <= rs ? rs + 1 : re) <= start) continue;                 // d_overlaps
        if (!(rs < pos3 - 10 && re > pos5 + 10 + 1)) continue;           // pairflanks.py:381-383 (quirk P7)
        const uint8_t f = (uint8_t)f32;
        if ((int)mq < P.minq) continue;                                   // d_passes: mapping quality, then the inner length of a
        if (d_lkind(f) && d_rkind(f) && (int)R.qlen[r] - d_rlen(R, r, f) - d_llen(R, r, f) < P.minial) continue;   // read clipped at both ends
        if (d_large_insertion_at(R, P, r, f, pos3) || d_large_insertion_at(R, P, r, f, pos5)) continue;
        bool dup = false;
        int64_t prev = r;
        for (int64_t p = lk; p >= 0 && p < prev; prev = p, p = R.link[p]) {
            if ((int32_t)R.contig_id[p] != contig) continue;
            if (!(R.ref_start[p] < end && d_overlaps(R, p, start))) continue;
            uint8_t fp = R.flags[p];
            if (d_passes(R, P, p, fp) && d_spans(R, P, p, fp, pos3, pos5)) { dup = true; break; }
        }
        if (!dup) mine++;
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) spanning[c] = cnt;
}

// kept pair k (candidate kept[k]): its spanning count + filter_junction_spanning pairflanks.py:421-431 (float64, strict <,
// quirk P9) + the logged counters
__global__ void k_pair_finish(DevParams P, FlankDev F, const uint32_t *kept, const uint32_t *c5, const uint32_t *c3, const uint32_t *n_kept, const int32_t *ir,
                              const uint32_t *span_all, uint32_t *spanning, uint32_t *jflag, uint32_t *pstats) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= *n_kept) return;
    const uint32_t c = kept[k], cnt = span_all[c];
    spanning[k] = cnt;
    double total = (double)((int64_t)F.total[c5[c]] + (int64_t)F.total[c3[c]]) / 2;
    double prop = (double)cnt / total;
    jflag[k] = prop < P.maxjsp ? 1u : 0u;
    if (ir[6 * c]) atomicAdd(&pstats[2], 1u);
    if (cnt > 0) atomicAdd(&pstats[3], 1u);
}

// final rows: the kept pairs that pass the junction filter, in kept order
__global__ void k_pair_final(const uint32_t *kept, const uint32_t *n_kept, const uint32_t *jflag, const uint32_t *jscan, const uint32_t *spanning, uint32_t *fin, uint32_t *fin_span) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= *n_kept || !jflag[k]) return;
    fin[jscan[k]] = kept[k];
    fin_span[jscan[k]] = spanning[k];
}


// ---- get_direct_repeats ---------------------------------------------------------------------------------------------------
#define DR_WIN 50
#define DR_SEGS 1536
#define DR_SEGS_SMALL 384   // 256-thread workgroups
#define DR_FLIGHT 4   // eight in flight were no faster at C2 and slower with a thousand pairs
// per candidate pair c: dr_out[c*stride ...] = reference repeat, dr_out[c*stride + half ...] = reads consensus.
// One block per CANDIDATE pair (like k_spanning, on the side stream next to the inverted-repeat alignment); the host
// takes the rows of the final pairs.
// MODE 0: all of it in one workgroup per pair.  A deep isolate has a few dozen pairs with more than a thousand reads over
// each window, i.e. a few dozen workgroups each walking a long chain: then the reads of a pair are split over gridDim.y
// workgroups (MODE 1: pile-up of a share of the reads, flushed into per-pair accumulators in HBM by atomics) and a
// second launch (MODE 2) reads the accumulators and writes the repeats.
#define DR_ACC (DR_WIN * 17)
template <int MODE, int SEGS>
__global__ __launch_bounds__(SEGS == DR_SEGS ? 1024 : 256) void k_direct_repeats(DevReads R, FlankDev F, uint32_t n_cand, const uint32_t *c5, const uint32_t *c3,
                                                        const uint8_t *genome, const int64_t *goff, char *dr_out, uint32_t stride, int32_t *dr_meta /* 3 per pair: ref_len, reads_len, err */,
                                                        uint32_t *g_sum, uint32_t *g_first, uint32_t *g_err) {
    __shared__ uint32_t sum[DR_WIN][17];
    __shared__ uint32_t firstr[DR_WIN][17];
    __shared__ uint8_t refch[DR_WIN];
    __shared__ int64_t range[2], prange[2];
    __shared__ int err;
    // SEGS: room for the segments of one chunk of blockDim.x reads (a read has one aligned block, a few have two or three);
    // with thousands of pairs the workgroups are 256 threads wide and the short list lets eight of them share a CU
    __shared__ uint64_t seg_q[SEGS];
    __shared__ int32_t seg_lo[SEGS], seg_hi[SEGS];
    __shared__ uint32_t seg_r[SEGS], nseg;
    if (blockIdx.x >= n_cand) return;
    const uint32_t c = blockIdx.x, k = c;      // output row = candidate
    const int32_t contig = F.contig[c5[c]], end = F.pos[c5[c]], start = F.pos[c3[c]];
    const int64_t glen = goff[contig + 1] - goff[contig];
    const uint8_t *g = genome + goff[contig];
    // round((end + start) / 2): Python 3 rounds x.5 to the even neighbour (pairflanks.py:294); sum >= -1
    int64_t s2 = (int64_t)end + start, center;
    if (s2 % 2 == 0) center = s2 / 2; else { int64_t lo = (s2 - 1) / 2; center = (lo % 2 == 0) ? lo : lo + 1; }
    int64_t es = center - DR_WIN / 2, ee = center + DR_WIN / 2, add_start_n = 0;
    if (es < 0) { add_start_n = -es; es = 0; }
    if (ee > glen) ee = glen;
    int W = (int)(ee - es); if (W < 0) W = 0;
    for (int i = threadIdx.x; i < DR_WIN * 17; i += blockDim.x) { (&sum[0][0])[i] = 0; (&firstr[0][0])[i] = 0xFFFFFFFFu; }
    if (threadIdx.x < W) {
        // initialize_target_region_reads pairflanks.py:347-353 with the mis-indexed left padding of 307 (quirk P10)
        int64_t idx = threadIdx.x;
        refch[threadIdx.x] = idx < add_start_n ? (uint8_t)'N' : ((es + idx - add_start_n) < glen ? g[es + idx - add_start_n] : (uint8_t)'N');
    }
    // cooperative searches, two at a time (first two waves, one bound each): the read range of the window, then the
    // pooled reads of that read range (the per-read lookups below search only this slice of pool_read)
    if (threadIdx.x == 0) err = 0;
    if (MODE != 2 && threadIdx.x < 128) d_fetch_range_2waves(R, contig, W > 0 ? (int32_t)es : 0, W > 0 ? (int32_t)ee : 0, range);
    __syncthreads();
    if (MODE != 2 && threadIdx.x < 128) {
        const int w = threadIdx.x >> 6;
        // (every read pooled: pool_read is the identity, no search)
        const int64_t pv = R.n_pool == R.n ? range[w] : d_wave_lower_bound<uint32_t, int64_t>(R.pool_read, 0, R.n_pool, range[w]);
        if ((threadIdx.x & 63) == 0) prange[w] = pv;
    }
    __syncthreads();
    // this workgroup's share of the read range (MODE 1: contiguous shares, one per blockIdx.y)
    int64_t r_begin = 0, r_end = 0;
    if (MODE != 2) {
        r_begin = range[0]; r_end = range[1];
        if (MODE == 1) {
            const int64_t share = (r_end - r_begin + gridDim.y - 1) / gridDim.y;
            r_begin += (int64_t)blockIdx.y * share;
            if (r_begin + share < r_end) r_end = r_begin + share;
        }
    }
    // reads of bam.fetch(contig, expanded_start, expanded_end): no read filter here (pairflanks.py:311).
    // Two phases per chunk of reads so that the LDS atomics of one wave instruction go to 50 different columns instead
    // of 64 reads hammering the same column: (1) one thread per read resolves its pool offset and aligned blocks into
    // window-clipped segments in LDS; (2) one wave per segment, lane = window position.
    for (int64_t rbase = r_begin; rbase < r_end; rbase += blockDim.x) {
        if (threadIdx.x == 0) nseg = 0;
        __syncthreads();
        const int64_t r = rbase + threadIdx.x;
        // the read's columns together, before the first test (one round trip instead of a chain of four)
        int32_t x_rs = 0, x_re = 0; uint32_t x_ql = 0, x_f = 0, x_fo = 0; uint64_t x_po = 0;
        const bool all_pooled = R.n_pool == R.n;
        if (r < r_end) {
            x_rs = R.ref_start[r]; x_re = R.ref_end[r]; x_ql = R.qlen[r]; x_f = R.flags[r]; x_fo = R.first_op_len[r];
            if (all_pooled) x_po = R.pool_off[r];
            asm volatile("" : "+v"(x_rs), "+v"(x_re), "+v"(x_ql), "+v"(x_f), "+v"(x_fo));
        }
        if (r < r_end && (x_re <= x_rs ? x_rs + 1 : x_re) > (int32_t)es && x_ql != 0) {
            const uint8_t f = (uint8_t)x_f;
            int64_t pi = -1;
            {   // the pool usually holds every read of the window: then read r sits at a fixed offset from the range start
                const int64_t guess = prange[0] + (r - range[0]);
                if (R.n_pool == R.n) pi = r;
                else if (guess < prange[1] && (int64_t)R.pool_read[guess] == r) pi = guess;
                else { int64_t lo = prange[0], hi = prange[1]; while (lo < hi) { int64_t m = (lo + hi) >> 1; if ((int64_t)R.pool_read[m] < r) lo = m + 1; else hi = m; } if (lo < prange[1] && (int64_t)R.pool_read[lo] == r) pi = lo; }
            }
            if (pi < 0) err = 1;
            else {
                const uint64_t o0 = all_pooled ? x_po : R.pool_off[pi];
                DevBlocks bl = d_blocks(R, r, f);
                if (!(f & MFK_FLAG_COMPLEX)) { bl.s1 = x_rs; bl.e1 = x_re; }   // (the same values, already here)
                int64_t qpos = d_lkind(f) == 1 ? (int64_t)x_fo : 0;
                for (int bi = 0; bi < bl.n; bi++) {
                    qpos += bl.bi(bi);
                    const int32_t bs = bl.bs(bi), be = bl.be(bi);
                    const int32_t lo = bs > (int32_t)es ? bs : (int32_t)es, hi = be < (int32_t)ee ? be : (int32_t)ee;
                    if (lo < hi) {
                        const uint64_t q0 = o0 + (uint64_t)(qpos + (lo - bs));
                        uint32_t slot = atomicAdd(&nseg, 1u);
                        if (slot < SEGS) { seg_q[slot] = q0; seg_lo[slot] = lo; seg_hi[slot] = hi; seg_r[slot] = (uint32_t)r; }
                        else for (int32_t rp = lo; rp < hi; rp++) {   // segment list full: add it here (slow, rare)
                            const uint64_t qi = q0 + (uint64_t)(rp - lo);
                            const uint32_t code = d_pool_base(R, qi);
                            atomicAdd(&sum[rp - es][code], (uint32_t)R.pool_quals[qi]);
                            atomicMin(&firstr[rp - es][code], (uint32_t)r);
                        }
                    }
                    qpos += be - bs;
                }
            }
        }
        __syncthreads();
        const uint32_t ns = nseg < SEGS ? nseg : SEGS;
        const int lane = threadIdx.x & 63;
        const uint32_t nw = blockDim.x >> 6;
        // Segments in flight per wave.  Only the two loaded bytes of a segment stay in registers while the loads are out (the
        // segment's column, read and nibble are looked up in LDS again afterwards), so the 256-thread variant can keep eight in
        // flight without losing occupancy (with five registers per segment it lost two waves per SIMD: 0.85 -> 0.96 ms at C5).
        constexpr int FL = SEGS == DR_SEGS ? DR_FLIGHT : 2 * DR_FLIGHT;
        for (uint32_t sgi = threadIdx.x >> 6; sgi < ns; sgi += FL * nw) {
            uint32_t bb[FL], qq[FL], on = 0;
#pragma unroll
            for (int u = 0; u < FL; u++) {
                const uint32_t sg = sgi + u * nw;
                bb[u] = 0; qq[u] = 0;
                if (sg < ns && seg_lo[sg] + lane < seg_hi[sg]) {
                    const uint64_t qi = seg_q[sg] + (uint64_t)lane;
                    bb[u] = R.pool_bases[qi >> 1]; qq[u] = R.pool_quals[qi];
                    on |= 1u << u;
                }
            }
#pragma unroll
            for (int u = 0; u < FL; u++)
                if (on & (1u << u)) {
                    const uint32_t sg = sgi + u * nw;
                    const uint32_t odd = (uint32_t)(seg_q[sg] + (uint64_t)lane) & 1u;
                    const uint32_t code = odd ? (bb[u] & 15u) : (bb[u] >> 4);     // even base index = high nibble (BAM order)
                    const int col = (int)(seg_lo[sg] + lane - es);
                    atomicAdd(&sum[col][code], qq[u]); atomicMin(&firstr[col][code], seg_r[sg]);
                }
        }
        __syncthreads();
    }
    if (MODE == 1) {   // this share's sums into the pair's accumulators; MODE 2 takes it from there
        for (int i = threadIdx.x; i < DR_ACC; i += blockDim.x) {
            const uint32_t v = (&sum[0][0])[i], fr = (&firstr[0][0])[i];
            if (v) atomicAdd(&g_sum[(size_t)c * DR_ACC + i], v);
            if (fr != 0xFFFFFFFFu) atomicMin(&g_first[(size_t)c * DR_ACC + i], fr);
        }
        if (threadIdx.x == 0 && err) atomicOr(&g_err[c], 1u);
        return;
    }
    if (MODE == 2) {
        for (int i = threadIdx.x; i < DR_ACC; i += blockDim.x) { (&sum[0][0])[i] = g_sum[(size_t)c * DR_ACC + i]; (&firstr[0][0])[i] = g_first[(size_t)c * DR_ACC + i]; }
        if (threadIdx.x == 0) err = (int)g_err[c];
        __syncthreads();
    }
    int32_t *meta = dr_meta + 3 * (size_t)k;
    char *out_ref = dr_out + (size_t)k * stride, *out_reads = out_ref + stride / 2;
    // get_reference_direct_repeats pairflanks.py:356-367: genome[start+1:end] with Python slice clamping
    int64_t a = (int64_t)start + 1, b = end;
    if (a < 0) a = 0;
    if (b > glen) b = glen;
    if (b < a) b = a;
    for (int64_t i = a + threadIdx.x; i < b; i += blockDim.x) out_ref[i - a] = (char)g[i];
    // .index() raises outside the window (pairflanks.py:323)
    bool bad = (start + 1 < es || start + 1 >= ee || end < es || end >= ee);
    if (threadIdx.x == 0) { meta[0] = (int32_t)(b - a); meta[1] = bad ? 0 : end - start - 1; meta[2] = bad ? 2 : err; }
    if (bad) return;
    // get_consensus_target_region pairflanks.py:332-344: keys in insertion order (reference base first, then read bases
    // by first contributing read), strict '>' from 0
    for (int64_t p = (int64_t)start + 1 + threadIdx.x; p < end; p += blockDim.x) {
        int col = (int)(p - es);
        uint8_t rc = refch[col];
        int refcode = -1;
        for (int q = 0; q < 16; q++) if ((uint8_t)d_code_char(q) == rc) refcode = q;
        // reference key first: weight 1 (+ read qualities when a read shows the same character); always > 0
        uint64_t best_q = 1 + (refcode >= 0 ? sum[col][refcode] : 0);
        char best_n = (char)rc;
        uint32_t best_first = 0;
        bool best_is_ref = true;
        for (int q = 0; q < 16; q++) {
            if (q == refcode) continue;
            uint32_t fr = firstr[col][q];
            if (fr == 0xFFFFFFFFu) continue;
            uint64_t v = sum[col][q];
            // a later key replaces the best only with a strictly larger sum; among equal sums the earlier key stays
            if (v > best_q || (v == best_q && !best_is_ref && fr < best_first)) { best_q = v; best_n = d_code_char(q); best_first = fr; best_is_ref = false; }
        }
        out_reads[p - (start + 1)] = best_n;
    }
}

// ---- pair rows on the device (thousands of pairs): the frame of run_pair_flanks pairflanks.py:96-122 ---------------------------
// As for the flank rows (k_rows_fill): with ten thousand pairs the host loop over nine read-back arrays costs a quarter of a
// millisecond.  k_prow_count gives every final pair its string bytes (IR_3p, IR_5p, reference repeat, reads repeat, in that
// order) and raises a flag if a pair carries a direct-repeat error (then the host takes its own loop, which words the error);
// a scan places the strings; k_prow_fill writes the mfk_pair records and the strings.
__global__ void k_prow_count(FlankDev F, const uint32_t *fin, const uint32_t *n_fin, const int32_t *ir, const int32_t *meta, uint32_t *bytes, uint32_t *err) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= *n_fin) return;
    const uint32_t c = fin[k];
    const int32_t *m = meta + 3 * (size_t)c, *r = ir + 6 * (size_t)c;
    if (m[2]) atomicOr(err, 1u);
    uint32_t b = (uint32_t)m[0] + (uint32_t)m[1];
    if (r[1]) b += (uint32_t)(r[2] - r[1] + 1) + (uint32_t)(r[4] - r[3] + 1);
    bytes[k] = b;
}
__global__ void k_prow_fill(FlankDev F, const uint32_t *fin, const uint32_t *fin_span, const uint32_t *n_fin, const uint32_t *c5, const uint32_t *c3, const int32_t *ir,
                            const int32_t *meta, const char *dr, uint32_t stride, const uint64_t *off, mfk_pair *pairs, char *strs) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= *n_fin) return;
    const uint32_t c = fin[k], a = c5[c], b = c3[c];
    const int32_t *m = meta + 3 * (size_t)c, *r = ir + 6 * (size_t)c;
    mfk_pair o;
    o.contig_id = F.contig[a]; o.pos_5p = F.pos[a]; o.pos_3p = F.pos[b]; o.index_5p = (int32_t)a; o.index_3p = (int32_t)b;
    o.softclip_count_5p = F.ccount[a]; o.softclip_count_3p = F.ccount[b];
    o.total_count_5p = F.total[a]; o.total_count_3p = F.total[b]; o.spanning_count = (int32_t)fin_span[k];
    o.has_ir = r[0]; o.ir5_len = o.ir3_len = -1; o.ir_length = 0;
    o.ir5_off = 0; o.ir3_off = 0;
    int64_t at = (int64_t)off[k];
    if (r[1]) {
        const int l5 = F.seq_len[a], l3 = F.seq_len[b];
        const int n5 = l5 > 40 ? 40 : l5, n3 = l3 > 40 ? 40 : l3;
        const char *s5 = F.seqs + F.seq_off[a], *s3 = F.seqs + F.seq_off[b] + (l3 - n3);
        // character q (0-based) of seq5[:40] + 'N' * 20 + seq3[-40:] (pairflanks.py:183-186)
        auto comb = [&](int q) -> char { return q < n5 ? s5[q] : (q < n5 + 20 ? 'N' : s3[q - n5 - 20]); };
        o.ir3_len = r[2] - r[1] + 1; o.ir3_off = at;       // IR_3p = ir1.seq
        for (int i = 0; i < o.ir3_len; i++) strs[at + i] = comb(r[1] - 1 + i);
        at += o.ir3_len;
        o.ir5_len = r[4] - r[3] + 1; o.ir5_off = at;       // IR_5p = ir2.seq
        for (int i = 0; i < o.ir5_len; i++) strs[at + i] = comb(r[3] - 1 + i);
        at += o.ir5_len;
        o.ir_length = o.ir5_len;
    }
    o.dr_ref_len = m[0]; o.dr_len = m[1];
    const char *d = dr + (size_t)c * stride;
    o.dr_ref_off = at;
    for (int i = 0; i < m[0]; i++) strs[at + i] = d[i];
    at += m[0];
    o.dr_reads_off = at;
    for (int i = 0; i < m[1]; i++) strs[at + i] = d[stride / 2 + i];
    pairs[k] = o;
}

// =====================================================================================================================
extern "C" int mfk_ir_detect(mfk_ctx *ctx, const char *seqs, const int64_t *off, int64_t n, int32_t *out) {
    if (!ctx || n < 0) return 1;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->err[0] = 0;
    if (n == 0) return 0;
    arena_reset(ctx);
    hipStream_t st = ctx->stream;
    int64_t total = off[n];
    char *d_seqs = aalloc<char>(ctx, (size_t)total + 1); int64_t *d_off = aalloc<int64_t>(ctx, (size_t)n + 1); int32_t *d_out = aalloc<int32_t>(ctx, 6 * (size_t)n);
    if (!d_seqs || !d_off || !d_out) return 1;
    HIPCHK(hipMemcpyAsync(d_seqs, seqs, (size_t)total, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_off, off, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, st));
    SPAN_EVENT(ev0);
    PROF_LAUNCH(k_ir_seqs, dim3((unsigned)n), dim3(64), 0, st, (const char *)d_seqs, (const int64_t *)d_off, (uint32_t)n, d_out);
    KCHK();
    SPAN_EVENT(ev1);
    HIPCHK(hipMemcpyAsync(out, d_out, 6 * (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    float ms = 0.f;
    if (ctx->span_on && hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ctx->last_ms = ms;
    ctx->call_ok = true;
    return 0;
}

extern "C" int mfk_pairflanks(mfk_ctx *ctx, const mfk_reads *reads, const mfk_params *params, const mfk_flank *flanks, const char *flank_seqs, int64_t n_flanks,
                              const int32_t *h_contig_rank, const uint8_t *genome, const int64_t *h_genome_off, int64_t *n_pairs_out, int64_t *str_bytes_out) {
    return pairflanks_impl(ctx, reads, params, flanks, flank_seqs, n_flanks, h_contig_rank, genome, h_genome_off, n_pairs_out, str_bytes_out, nullptr);
}

// fh (mfk_flankpairs, thousands of sites): the flank table is on the device, behind findflanks kernels that are still running;
// this call continues in the same arena and on the same stream, and reads findflanks' results back with its own.
int pairflanks_impl(mfk_ctx *ctx, const mfk_reads *reads, const mfk_params *params, const mfk_flank *flanks, const char *flank_seqs, int64_t n_flanks,
                    const int32_t *h_contig_rank, const uint8_t *genome, const int64_t *h_genome_off, int64_t *n_pairs_out, int64_t *str_bytes_out, FlankHandoff *fh) {
    if (!ctx || !reads || !params || n_flanks < 0) return 1;
    const bool dev_in = fh && fh->active;
    std::vector<int32_t> ir_tab;
    ir_tab.swap(ctx->ir_ov);                              // the external table (if any) serves exactly this call
    HIPCHK(hipSetDevice(ctx->device));
    ctx->err[0] = 0;
    ctx->pairs.clear(); ctx->strs.clear();
    memset(ctx->pstats, 0, sizeof ctx->pstats);
    if (n_pairs_out) *n_pairs_out = 0;
    if (str_bytes_out) *str_bytes_out = 0;
    if (n_flanks == 0 && !dev_in) return 0;
    HostTimer ht("pairflanks");
    if (params->max_direct_repeat_length > 65000 || params->max_direct_repeat_length < 0) { snprintf(ctx->err, sizeof ctx->err, "max_direct_repeat_length out of range"); return 1; }
    if (!dev_in) arena_reset(ctx);
    DevReads R; DevParams P;
    if (int rc = upload_reads(ctx, reads, params, &R, &P)) return rc;   // (hand-over: same header bytes as findflanks uploaded, nothing is copied)
    const size_t pin_off = upload_reads_pinned_bytes(reads->n_contigs);   // the flank staging block goes behind upload_reads' bytes
    hipStream_t st = ctx->stream;
    const uint32_t nF = dev_in ? 2u * fh->n_act : (uint32_t)n_flanks;   // hand-over: an upper bound, the rows behind the last one are inert
    const int nc = reads->n_contigs;
    // findflanks' results that still sit on the device (hand-over): sizes known after the first read-back below, tables added
    // to the last one
    bool ff_pending = false; size_t ff_ns = 0, ff_nf = 0, ff_nb = 0;
    auto add_ff_tables = [&](ReadBatch &rb) {
        if (!ff_pending) return;
        rb.add(fh->d_sites, ctx->sites.data(), ff_ns * sizeof(mfk_site)); rb.add(fh->d_flanks, ctx->flanks.data(), ff_nf * sizeof(mfk_flank));
        if (ff_nb) rb.add(fh->d_seqs, &ctx->seqs[0], ff_nb);
        ff_pending = false;
    };
    // ---- flank table to the device: one staging block in pinned memory, one copy --------------------------------------
    int sorted = 1; int64_t seq_total = 0;
    for (uint32_t i = 0; i < nF && !dev_in; i++) {
        const mfk_flank &f = flanks[i];
        if (f.contig_id < 0 || f.contig_id >= nc) { snprintf(ctx->err, sizeof ctx->err, "flank %u: contig id %d out of range", i, f.contig_id); return 1; }
        if (f.seq_off < 0 || f.seq_len < 0) { snprintf(ctx->err, sizeof ctx->err, "flank %u: negative sequence offset or length", i); return 1; }
        if (f.seq_off + f.seq_len > seq_total) seq_total = f.seq_off + f.seq_len;
        if (i && (flanks[i - 1].contig_id > f.contig_id || (flanks[i - 1].contig_id == f.contig_id && flanks[i - 1].pos > f.pos))) sorted = 0;
    }
    // Few flanks (a single isolate has a few dozen): pair_all_nearby_flanks runs right here on the host table (same
    // order as k_pair_window: 5p rows in file order, their 3p partners in file order) and the candidate list travels with
    // the flank table.  That takes two kernels, a scan and a read-back of the candidate count off the head of the call.
    std::vector<uint32_t> host_c5, host_c3;
    const char *nh_env = getenv("MFK_NO_HOST_PAIRS");   // tests: force the device path for small tables too
    const bool host_pairs = !dev_in && nF <= PW_HOST_MAX && !(nh_env && *nh_env && *nh_env != '0');
    if (host_pairs)
        for (uint32_t a = 0; a < nF; a++) {
            if (flanks[a].orient != 5) continue;
            const int64_t min_pos = (int64_t)flanks[a].pos - params->max_direct_repeat_length - 1;
            for (uint32_t b = 0; b < nF; b++)
                if (flanks[b].orient == 3 && flanks[b].contig_id == flanks[a].contig_id && (int64_t)flanks[b].pos >= min_pos && flanks[b].pos < flanks[a].pos) { host_c5.push_back(a); host_c3.push_back(b); }
        }
    // (hand-over: only the genome offsets and the contig ranks come from the host; the flank columns are made on the device)
    const size_t nFu = dev_in ? 0 : (size_t)nF;
    const size_t o_soff = 0, o_goff = o_soff + nFu * 8, o_tab = o_goff + ((size_t)nc + 1) * 8, o_rank = o_tab + 6 * nFu * 4;
    const size_t o_c5 = o_rank + (size_t)nc * 4, o_c3 = o_c5 + host_c5.size() * 4;
    const size_t o_seqs = o_c3 + host_c3.size() * 4, up_bytes = ((o_seqs + (size_t)seq_total + 1) + 255) & ~(size_t)255;
    char *hp = (char *)pinned_buf(ctx, pin_off + up_bytes);
    char *d_up = aalloc<char>(ctx, up_bytes);
    if (!hp || !d_up) return 1;
    hp += pin_off;
    if (dev_in) {
        memcpy(hp + o_goff, h_genome_off, ((size_t)nc + 1) * 8);
        memcpy(hp + o_rank, h_contig_rank, (size_t)nc * 4);
    } else {
        int64_t *hoff = (int64_t *)(hp + o_soff); int32_t *h = (int32_t *)(hp + o_tab);
        for (uint32_t i = 0; i < nF; i++) {
            const mfk_flank &f = flanks[i];
            h[i] = f.contig_id; h[nF + i] = f.pos; h[2 * (size_t)nF + i] = f.orient; h[3 * (size_t)nF + i] = f.consensus_softclip_count;
            h[4 * (size_t)nF + i] = f.total_count; h[5 * (size_t)nF + i] = f.seq_len; hoff[i] = f.seq_off;
        }
        memcpy(hp + o_goff, h_genome_off, ((size_t)nc + 1) * 8);
        memcpy(hp + o_rank, h_contig_rank, (size_t)nc * 4);
        if (seq_total) memcpy(hp + o_seqs, flank_seqs, (size_t)seq_total);
        if (!host_c5.empty()) { memcpy(hp + o_c5, host_c5.data(), host_c5.size() * 4); memcpy(hp + o_c3, host_c3.data(), host_c3.size() * 4); }
    }
    ht.lap("staging");
    if (int rc = upload_pinned(ctx, d_up, hp, up_bytes)) return rc;   // pinned bytes stay untouched until the final read-back, which is ordered behind this copy
    int32_t *d_tab = (int32_t *)(d_up + o_tab); int64_t *d_soff = (int64_t *)(d_up + o_soff); const char *d_seqs = d_up + o_seqs;
    int32_t *d_rank = (int32_t *)(d_up + o_rank); int64_t *d_goff = (int64_t *)(d_up + o_goff);
    if (dev_in) {
        d_tab = aalloc<int32_t>(ctx, 6 * (size_t)nF); d_soff = aalloc<int64_t>(ctx, nF); d_seqs = fh->d_seqs;
        if (!d_tab || !d_soff) return 1;
        PROF_LAUNCH(k_flank_cols, dim3(nblocks(nF, 256)), dim3(256), 0, st, fh->d_flanks, fh->d_rows_total, nF, d_tab, d_soff);
        KCHK();
    } else
        SPAN_EVENT(ev0);   // (hand-over: the call's device time runs from findflanks' first kernel)
    FlankDev F; F.n = nF; F.contig = d_tab; F.pos = d_tab + nF; F.orient = d_tab + 2 * (size_t)nF; F.ccount = d_tab + 3 * (size_t)nF;
    F.total = d_tab + 4 * (size_t)nF; F.seq_len = d_tab + 5 * (size_t)nF; F.seq_off = d_soff; F.seqs = d_seqs;
    // ---- candidate pairs ----------------------------------------------------------------------------------------------
    const unsigned fb = nblocks(nF, 256);
    uint32_t *cnt = nullptr, *coff = nullptr;
    uint32_t nP = 0;
    if (host_pairs) nP = (uint32_t)host_c5.size();
    else {
        cnt = aalloc<uint32_t>(ctx, nF); coff = aalloc<uint32_t>(ctx, (size_t)nF + 1);
        if (!cnt || !coff) return 1;
        PROF_LAUNCH(k_pair_window<false>, dim3(fb), dim3(256), 0, st, F, P.maxdr, sorted, cnt, (const uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr);
        KCHK();
        if (int rc = dev_excl_scan32(ctx, cnt, coff, nF)) return rc;
        if (!dev_in) { if (int rc = read_back(ctx, coff + nF, &nP, 4)) return rc; }
        else {
            // one read-back for the candidate count and for everything findflanks would have read back by now
            uint32_t f_stats[MFK_N_STATS], f_scan_err = 0, f_cons[4] = {0, 0, 0, 0}; uint64_t t_rows = 0, t_bytes = 0;
            {
                ReadBatch rb(ctx);
                rb.add(coff + nF, &nP, 4); rb.add(fh->d_stats, f_stats, sizeof f_stats); rb.add(ctx->scan_err, &f_scan_err, 4);
                if (fh->d_cons_err) rb.add(fh->d_cons_err, f_cons, 16);
                rb.add(fh->d_rows_total, &t_rows, 8); rb.add(fh->d_bytes_total, &t_bytes, 8);
                if (int rc = rb.flush()) return rc;
            }
            if (int rc = check_scan_error(ctx, f_scan_err)) return rc;
            if (fh->sure_small && f_cons[3]) { snprintf(ctx->err, sizeof ctx->err, "internal: a consensus job did not finish in k_job_small although its size bound said it would"); return 11; }
            if (f_cons[0] & 2u) { snprintf(ctx->err, sizeof ctx->err, "a clipped read at a surviving site has no sequence in the pool (pack with pool='clipped')"); return 2; }
            ff_ns = (size_t)(t_rows & 0xFFFFFFull); ff_nf = (size_t)(t_rows >> 24); ff_nb = (size_t)t_bytes;
            ctx->sites.resize(ff_ns); ctx->flanks.resize(ff_nf); ctx->seqs.resize(ff_nb);
            ff_pending = true;
            for (int i = 0; i < 10; i++) ctx->stats[i] = f_stats[i];
            ctx->stats[12] = (int64_t)((uint64_t)f_stats[12] | ((uint64_t)f_stats[13] << 32));
            ctx->stats[11] = (int64_t)ff_nf;
        }
    }
    ctx->pstats[0] = nP;
    ht.lap("candidates + read-back");
    uint32_t nK = 0, nFin = 0;
    uint32_t h_ps[MFK_N_STATS], h_scan_err = 0; memset(h_ps, 0, sizeof h_ps);
    std::vector<uint32_t> h_fin, h_span, h_c5, h_c3; std::vector<int32_t> h_ir, h_meta; std::vector<char> h_dr; uint32_t stride = 0;
    if (nP > 0) {
        // everything downstream is sized by the candidate count; the kept / final counts stay on the device
        uint32_t *c5 = host_pairs ? (uint32_t *)(d_up + o_c5) : aalloc<uint32_t>(ctx, nP), *c3 = host_pairs ? (uint32_t *)(d_up + o_c3) : aalloc<uint32_t>(ctx, nP);
        int32_t *ir = aalloc<int32_t>(ctx, 6 * (size_t)nP);
        uint64_t *keys = aalloc<uint64_t>(ctx, nP), *keys2 = aalloc<uint64_t>(ctx, nP), *kk2 = aalloc<uint64_t>(ctx, nP), *kk = nullptr;
        uint32_t *vals = aalloc<uint32_t>(ctx, nP), *order = aalloc<uint32_t>(ctx, nP);
        uint32_t *best5 = aalloc<uint32_t>(ctx, nF), *best3 = aalloc<uint32_t>(ctx, nF);
        uint8_t *used5 = nullptr, *used3 = nullptr, *state = aalloc<uint8_t>(ctx, nP);   // used5 / used3: zero-filled below (k_assign_seg takes them as they are)
        uint32_t *kflag = aalloc<uint32_t>(ctx, nP), *kscan = aalloc<uint32_t>(ctx, (size_t)nP + 1), *jscan = aalloc<uint32_t>(ctx, (size_t)nP + 1);
        uint32_t *kept = aalloc<uint32_t>(ctx, nP), *spanning = aalloc<uint32_t>(ctx, nP), *span_all = aalloc<uint32_t>(ctx, nP);
        uint32_t *fin = aalloc<uint32_t>(ctx, nP), *fin_span = aalloc<uint32_t>(ctx, nP);
        int32_t *meta = aalloc<int32_t>(ctx, 3 * (size_t)nP);
        stride = 2 * (((uint32_t)P.maxdr + 1 + 15) & ~15u);
        uint32_t *d_pstats = nullptr, *kv = nullptr, *jflag = nullptr; char *dr = nullptr;
        // few pairs under deep coverage: the reads over a pair's direct-repeat window (about coverage x (150 + 50) / 150 of
        // them) are split over several workgroups
        uint32_t dr_split = 1;
        uint32_t *g_sum = nullptr, *g_first = nullptr, *g_err = nullptr;
        if (nP <= 128) {
            int64_t ref_total = 0;
            for (int c = 0; c < nc; c++) ref_total += reads->h_contig_len[c];
            const double per_window = ref_total > 0 ? (double)reads->n_reads * (200.0 + R.max_span) / (double)ref_total : 0.0;
            dr_split = (uint32_t)(per_window / 192.0);
            if (dr_split > 8) dr_split = 8;
            if (dr_split < 1) dr_split = 1;
        }
        FillGroup z(ctx, 0), f(ctx, 0xFF);
        z.want(&d_pstats, MFK_N_STATS); z.want(&kv, nP); z.want(&jflag, nP); z.want(&dr, (size_t)nP * stride); z.want(&used5, nF); z.want(&used3, nF);
        f.want(&kk, nP);   // unused sort keys stay ~0 and sort behind the kept pairs
        if (dr_split > 1) { z.want(&g_sum, (size_t)nP * DR_ACC); z.want(&g_err, nP); f.want(&g_first, (size_t)nP * DR_ACC); }
        if (!c5 || !c3 || !ir || !keys || !keys2 || !kk2 || !vals || !order || !best5 || !best3 || !state || !kflag || !kscan || !jscan ||
            !kept || !spanning || !span_all || !fin || !fin_span || !meta || fill_commit2(z, f)) return 1;
        const unsigned pb = nblocks(nP, 256);
        if (!host_pairs) {
            PROF_LAUNCH(k_pair_window<true>, dim3(fb), dim3(256), 0, st, F, P.maxdr, sorted, (uint32_t *)nullptr, (const uint32_t *)coff, c5, c3);
            KCHK();
        }
        // Side streams: direct repeats and spanning reads of every candidate (they depend on the positions only), each on
        // its own stream.  Main stream: inverted repeats -> assign_pairs.  (Timing events of a launch go to the stream of that launch.)
        // A few dozen candidates (a single isolate): the inverted-repeat alignment is as long as the two other kernels together, so
        // ONE side stream carries both (three stream operations less for the host, which sets the pace of so short a call).
        const bool one_side = nP <= PA_SMALL;
        hipStream_t sb = ctx->stream2, sc = one_side ? ctx->stream2 : ctx->stream3;
        HIPCHK(hipEventRecord(ctx->ev_fork, st));
        HIPCHK(hipStreamWaitEvent(sb, ctx->ev_fork, 0));
        if (!one_side) HIPCHK(hipStreamWaitEvent(sc, ctx->ev_fork, 0));
        // the inverted-repeat alignment first: it is the longest chain of the stage and everything on this stream waits for it
        if (ir_tab.empty()) {
            PROF_LAUNCH(k_inverted_repeat, dim3(nP), dim3(64), 0, st, F, (const uint32_t *)c5, (const uint32_t *)c3, nP, ir, keys, vals);
            KCHK();
        } else {
            int32_t *d_tab_ir = aalloc<int32_t>(ctx, ir_tab.size());
            if (!d_tab_ir) return 1;
            HIPCHK(hipMemcpyAsync(d_tab_ir, ir_tab.data(), ir_tab.size() * 4, hipMemcpyHostToDevice, st));   // ir_tab lives until the call returns
            PROF_LAUNCH(k_ir_lookup, dim3(pb), dim3(256), 0, st, F, (const uint32_t *)c5, (const uint32_t *)c3, nP, (const int32_t *)d_tab_ir, (uint32_t)(ir_tab.size() / 7), ir, keys, vals, d_pstats + 15);
            KCHK();
        }
        if (dr_split > 1) {
            prof_begin(ctx, "k_direct_repeats<1>", sb);
                hipLaunchKernelGGL((k_direct_repeats<1, DR_SEGS>), dim3(nP, dr_split), dim3(1024), 0, sb, R, F, nP, (const uint32_t *)c5, (const uint32_t *)c3,
                        genome, (const int64_t *)d_goff, dr, stride, meta, g_sum, g_first, g_err);
                prof_end(ctx, sb);
            KCHK();
            prof_begin(ctx, "k_direct_repeats<2>", sb);
                hipLaunchKernelGGL((k_direct_repeats<2, DR_SEGS>), dim3(nP), dim3(64), 0, sb, R, F, nP, (const uint32_t *)c5, (const uint32_t *)c3,
                        genome, (const int64_t *)d_goff, dr, stride, meta, g_sum, g_first, g_err);
                prof_end(ctx, sb);
            KCHK();
        } else {
            if (nP >= 2048) {
                prof_begin(ctx, "k_direct_repeats<0>", sb);
                hipLaunchKernelGGL((k_direct_repeats<0, DR_SEGS_SMALL>), dim3(nP), dim3(256), 0, sb, R, F, nP, (const uint32_t *)c5, (const uint32_t *)c3,
                            genome, (const int64_t *)d_goff, dr, stride, meta, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr);
                prof_end(ctx, sb);
            } else {
                prof_begin(ctx, "k_direct_repeats<0>", sb);
                hipLaunchKernelGGL((k_direct_repeats<0, DR_SEGS>), dim3(nP), dim3(1024), 0, sb, R, F, nP, (const uint32_t *)c5, (const uint32_t *)c3,
                            genome, (const int64_t *)d_goff, dr, stride, meta, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr);
                prof_end(ctx, sb);
            }
            KCHK();
        }
        if (!one_side) HIPCHK(hipEventRecord(ctx->ev_join, sb));
        PROF_LAUNCH(k_spanning, dim3(nP), dim3(nP >= 2048 ? 256 : 1024), 0, sc, R, P, F, (const uint32_t *)c5, (const uint32_t *)c3, nP, span_all);
        KCHK();
        HIPCHK(hipEventRecord(one_side ? ctx->ev_join : ctx->ev_join3, sc));
        const uint32_t *nK_p = kscan + nP;
        if (nP <= PA_SMALL) {
            // few candidates (a deep isolate): assign_pairs in one launch, the junction filter in another
            PROF_LAUNCH(k_assign_small, dim3(1), dim3(PA_MAX), 0, st, F, (const uint64_t *)keys, (const uint32_t *)c5, (const uint32_t *)c3, nP, nF, (const int32_t *)d_rank,
                        best5, best3, used5, used3, kept, kscan + nP);
            KCHK();
            HIPCHK(hipStreamWaitEvent(st, ctx->ev_join, 0));
            if (!one_side) HIPCHK(hipStreamWaitEvent(st, ctx->ev_join3, 0));
            PROF_LAUNCH(k_pair_finish_small, dim3(1), dim3(PA_MAX), 0, st, P, F, (const uint32_t *)kept, (const uint32_t *)c5, (const uint32_t *)c3, nK_p, (const int32_t *)ir,
                        (const uint32_t *)span_all, fin, fin_span, jscan + nP, d_pstats);
            KCHK();
        } else {
        const char *na_env = getenv("MFK_NO_SEG_ASSIGN");   // test hook: the sort + one-workgroup rounds for a sorted table too
        if (sorted && !(na_env && *na_env && *na_env != '0')) {
            // (used5 / used3 start at zero: part of the zero-filled block above)
            PROF_LAUNCH(k_assign_seg, dim3(pb), dim3(256), 0, st, (const uint64_t *)keys, (const uint32_t *)c5, (const uint32_t *)c3, nP, used5, used3, state, kflag);
            KCHK();
            if (int rc = dev_excl_scan32(ctx, kflag, kscan, nP)) return rc;
            PROF_LAUNCH(k_kept_keys_seg, dim3(pb), dim3(256), 0, st, F, (const uint32_t *)c5, (const uint32_t *)c3, (const uint32_t *)kflag, (const uint32_t *)kscan, nP, (const int32_t *)d_rank, kk, kv);
            KCHK();
        } else {
        if (int rc = dev_sort_pairs64(ctx, keys, keys2, vals, order, nP, 0, 56)) return rc;
        PROF_LAUNCH(k_assign, dim3(1), dim3(1024), 0, st, (const uint32_t *)order, (const uint32_t *)c5, (const uint32_t *)c3, nP, nF, best5, best3, used5, used3, state, kflag);
        KCHK();
        if (int rc = dev_excl_scan32(ctx, kflag, kscan, nP)) return rc;
        PROF_LAUNCH(k_kept_keys, dim3(pb), dim3(256), 0, st, F, (const uint32_t *)order, (const uint32_t *)c5, (const uint32_t *)c3, (const uint8_t *)state, (const uint32_t *)kscan, nP, (const int32_t *)d_rank, kk, kv);
        KCHK();
        }
        if (int rc = dev_sort_pairs64(ctx, kk, kk2, kv, kept, nP, 0, 64)) return rc;
        HIPCHK(hipStreamWaitEvent(st, ctx->ev_join, 0));
        HIPCHK(hipStreamWaitEvent(st, ctx->ev_join3, 0));
        PROF_LAUNCH(k_pair_finish, dim3(pb), dim3(256), 0, st, P, F, (const uint32_t *)kept, (const uint32_t *)c5, (const uint32_t *)c3, nK_p, (const int32_t *)ir,
                    (const uint32_t *)span_all, spanning, jflag, d_pstats);
        KCHK();
        if (int rc = dev_excl_scan32(ctx, jflag, jscan, nP)) return rc;
        PROF_LAUNCH(k_pair_final, dim3(pb), dim3(256), 0, st, (const uint32_t *)kept, nK_p, (const uint32_t *)jflag, (const uint32_t *)jscan, (const uint32_t *)spanning, fin, fin_span);
        KCHK();
        }
        // thousands of candidates: the rows are assembled on the device (unless a pair carries a direct-repeat error)
        if (nP >= 2048) {
            uint32_t *pbytes = nullptr, *perr = nullptr;
            { FillGroup z2(ctx, 0); z2.want(&pbytes, nP); z2.want(&perr, 64); if (z2.commit()) return 1; }
            uint64_t *poff = aalloc<uint64_t>(ctx, (size_t)nP + 1);
            mfk_pair *d_pairs = aalloc<mfk_pair>(ctx, nP);
            const size_t str_cap = (size_t)nP * (2 * (size_t)IR_MAXLEN + stride) + 16;
            char *d_strs = aalloc<char>(ctx, str_cap);
            if (!poff || !d_pairs || !d_strs) return 1;
            const uint32_t *nfin_p = jscan + nP;
            PROF_LAUNCH(k_prow_count, dim3(pb), dim3(256), 0, st, F, (const uint32_t *)fin, nfin_p, (const int32_t *)ir, (const int32_t *)meta, pbytes, perr);
            KCHK();
            if (int rc = dev_excl_scan64(ctx, pbytes, poff, nP)) return rc;
            PROF_LAUNCH(k_prow_fill, dim3(pb), dim3(256), 0, st, F, (const uint32_t *)fin, (const uint32_t *)fin_span, nfin_p, (const uint32_t *)c5, (const uint32_t *)c3, (const int32_t *)ir,
                        (const int32_t *)meta, (const char *)dr, stride, (const uint64_t *)poff, d_pairs, d_strs);
            KCHK();
            SPAN_EVENT(ev1);
            uint32_t h_perr = 0; uint64_t h_total = 0;
            {
                ReadBatch rb(ctx);
                rb.add(nK_p, &nK, 4); rb.add(jscan + nP, &nFin, 4); rb.add(d_pstats, h_ps, sizeof h_ps); rb.add(ctx->scan_err, &h_scan_err, 4);
                rb.add(perr, &h_perr, 4); rb.add(poff + nP, &h_total, 8);
                ht.lap("enqueue");
                if (int rc = rb.flush()) return rc;
            }
            if (int rc = check_scan_error(ctx, h_scan_err)) return rc;
            if (!h_perr && !(!ir_tab.empty() && h_ps[15])) {
                ctx->pairs.resize(nFin); ctx->strs.resize((size_t)h_total);
                ReadBatch rb(ctx);
                rb.add(d_pairs, ctx->pairs.data(), (size_t)nFin * sizeof(mfk_pair)); if (h_total) rb.add(d_strs, &ctx->strs[0], (size_t)h_total);
                add_ff_tables(rb);
                if (int rc = rb.flush()) return rc;
                ht.lap("results read-back (rows made on the device)");
                ctx->pstats[1] = nK; ctx->pstats[2] = h_ps[2]; ctx->pstats[3] = h_ps[3]; ctx->pstats[4] = nFin;
                prof_collect(ctx);
                float ms = 0.f;
                if (ctx->span_on && hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ctx->last_ms = ms;
                ctx->call_ok = true;
                if (n_pairs_out) *n_pairs_out = (int64_t)ctx->pairs.size();
                if (str_bytes_out) *str_bytes_out = (int64_t)ctx->strs.size();
                return 0;
            }
            // an error among the final pairs (rare): the host loop below reports it in the reference's words
        }
        SPAN_EVENT(ev1);
        h_fin.resize(nP); h_span.resize(nP); h_meta.resize(3 * (size_t)nP); h_dr.resize((size_t)nP * stride);
        h_c5.resize(nP); h_c3.resize(nP); h_ir.resize(6 * (size_t)nP);
        ReadBatch rb(ctx);
        rb.add(nK_p, &nK, 4); rb.add(jscan + nP, &nFin, 4);
        rb.add(fin, h_fin.data(), nP * 4); rb.add(fin_span, h_span.data(), nP * 4);
        rb.add(meta, h_meta.data(), 3 * (size_t)nP * 4); rb.add(dr, h_dr.data(), (size_t)nP * stride);
        rb.add(c5, h_c5.data(), nP * 4); rb.add(c3, h_c3.data(), nP * 4); rb.add(ir, h_ir.data(), 6 * (size_t)nP * 4);
        rb.add(d_pstats, h_ps, sizeof h_ps);
        rb.add(ctx->scan_err, &h_scan_err, 4);
        add_ff_tables(rb);
        ht.lap("enqueue");
        if (int rc = rb.flush()) return rc;
        ht.lap("results read-back");
    } else {
        SPAN_EVENT(ev1);
        if (ff_pending) { ReadBatch rb(ctx); add_ff_tables(rb); if (int rc = rb.flush()) return rc; }
    }
    if (dev_in) { flanks = ctx->flanks.data(); flank_seqs = ctx->seqs.data(); }   // the host loop below reads the table findflanks made
    if (int rc = check_scan_error(ctx, h_scan_err)) return rc;
    ctx->pstats[1] = nK;
    const bool had_table = !ir_tab.empty();
    if (had_table && h_ps[15]) { snprintf(ctx->err, sizeof ctx->err, "a candidate pair has no entry in the inverted-repeat table given to mfk_pairflanks_set_ir"); return 8; }
    ctx->pstats[2] = h_ps[2]; ctx->pstats[3] = h_ps[3]; ctx->pstats[4] = nFin;
    prof_collect(ctx);
    float ms = 0.f;
    if (ctx->span_on && hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ctx->last_ms = ms;
    // ---- rows ------------------------------------------------------------------------------------------------------------
    ctx->pairs.reserve(nFin);
    ctx->strs.reserve((size_t)nFin * 64);
    for (uint32_t k = 0; k < nFin; k++) {
        uint32_t c = h_fin[k], a = h_c5[c], b = h_c3[c];
        const mfk_flank &f5 = flanks[a], &f3 = flanks[b];
        const int32_t *m = &h_meta[3 * (size_t)c];
        if (m[2] == 2) { snprintf(ctx->err, sizeof ctx->err, "ValueError: direct-repeat bounds %d..%d fall outside the target window (pairflanks.py:323)", f3.pos + 1, f5.pos); return 5; }
        if (m[2] == 1) { snprintf(ctx->err, sizeof ctx->err, "a read over a direct-repeat window has no sequence in the pool"); return 4; }
        mfk_pair o; memset(&o, 0, sizeof o);
        o.contig_id = f5.contig_id; o.pos_5p = f5.pos; o.pos_3p = f3.pos; o.index_5p = (int32_t)a; o.index_3p = (int32_t)b;
        o.softclip_count_5p = f5.consensus_softclip_count; o.softclip_count_3p = f3.consensus_softclip_count;
        o.total_count_5p = f5.total_count; o.total_count_3p = f3.total_count; o.spanning_count = (int32_t)h_span[k];
        const int32_t *r = &h_ir[6 * (size_t)c];
        o.has_ir = r[0]; o.ir5_len = o.ir3_len = -1; o.ir_length = 0;
        if (r[1]) {
            int n5 = f5.seq_len > 40 ? 40 : f5.seq_len, n3 = f3.seq_len > 40 ? 40 : f3.seq_len;
            char comb[IR_MAXLEN + 4];                      // seq5[:40] + 'N' * 20 + seq3[-40:] (pairflanks.py:183-186)
            memcpy(comb, flank_seqs + f5.seq_off, (size_t)n5);
            memset(comb + n5, 'N', 20);
            memcpy(comb + n5 + 20, flank_seqs + f3.seq_off + (f3.seq_len - n3), (size_t)n3);
            o.ir3_len = r[2] - r[1] + 1; o.ir3_off = (int64_t)ctx->strs.size(); ctx->strs.append(comb + r[1] - 1, (size_t)o.ir3_len);   // IR_3p = ir1.seq
            o.ir5_len = r[4] - r[3] + 1; o.ir5_off = (int64_t)ctx->strs.size(); ctx->strs.append(comb + r[3] - 1, (size_t)o.ir5_len);   // IR_5p = ir2.seq
            o.ir_length = o.ir5_len;
        }
        o.dr_ref_len = m[0]; o.dr_len = m[1];
        o.dr_ref_off = (int64_t)ctx->strs.size(); ctx->strs.append(h_dr.data() + (size_t)c * stride, (size_t)m[0]);
        o.dr_reads_off = (int64_t)ctx->strs.size(); ctx->strs.append(h_dr.data() + (size_t)c * stride + stride / 2, (size_t)m[1]);
        ctx->pairs.push_back(o);
    }
    ht.lap("host table");
    ctx->call_ok = true;
    if (n_pairs_out) *n_pairs_out = (int64_t)ctx->pairs.size();
    if (str_bytes_out) *str_bytes_out = (int64_t)ctx->strs.size();
    return 0;
}

extern "C" int mfk_pairflanks_fetch(mfk_ctx *ctx, mfk_pair *pairs, char *strs, int64_t *stats) {
    if (!ctx) return 1;
    if (pairs && !ctx->pairs.empty()) memcpy(pairs, ctx->pairs.data(), ctx->pairs.size() * sizeof(mfk_pair));
    if (strs && !ctx->strs.empty()) memcpy(strs, ctx->strs.data(), ctx->strs.size());
    if (stats) memcpy(stats, ctx->pstats, sizeof ctx->pstats);
    return 0;
}

// both stages in one call: the host flank table of the first feeds the second directly (no caller in between)
extern "C" int mfk_flankpairs(mfk_ctx *ctx, const mfk_reads *reads, const mfk_params *params, const int32_t *h_contig_rank, const uint8_t *genome,
                              const int64_t *h_genome_off, int64_t *n_sites, int64_t *n_flanks, int64_t *seq_bytes, int64_t *n_pairs, int64_t *str_bytes) {
    if (!ctx) return 1;
    if (n_pairs) *n_pairs = 0;
    if (str_bytes) *str_bytes = 0;
    const char *nh = getenv("MFK_NO_HANDOFF");   // test hook: always through the host table
    const bool no_handoff = nh && *nh && *nh != '0';
    FlankHandoff fh; memset(&fh, 0, sizeof fh);
    if (int rc = findflanks_impl(ctx, reads, params, MFK_STAGE_ALL, n_sites, n_flanks, seq_bytes, no_handoff ? nullptr : &fh)) return rc;
    const double ms_find = fh.active ? 0.0 : ctx->last_ms;
    ctx->pairs.clear(); ctx->strs.clear();
    memset(ctx->pstats, 0, sizeof ctx->pstats);
    int rc = pairflanks_impl(ctx, reads, params, ctx->flanks.data(), ctx->seqs.data(), (int64_t)ctx->flanks.size(), h_contig_rank, genome, h_genome_off, n_pairs, str_bytes, &fh);
    ctx->last_ms += ms_find;
    if (fh.active) {   // findflanks' sizes are known only now
        if (n_sites) *n_sites = (int64_t)ctx->sites.size();
        if (n_flanks) *n_flanks = (int64_t)ctx->flanks.size();
        if (seq_bytes) *seq_bytes = (int64_t)ctx->seqs.size();
    }
    return rc;
}

// External inverted-repeat results for the NEXT mfk_pairflanks call on this ctx (e.g. parsed from EMBOSS einverted, which
// is what the reference runs: embosstools.py:8-26, pairflanks.py:172-219).  One row per candidate pair.
extern "C" int mfk_pairflanks_set_ir(mfk_ctx *ctx, int64_t n, const int32_t *index_5p, const int32_t *index_3p, const int32_t *ir) {
    if (!ctx || n < 0 || (n > 0 && (!index_5p || !index_3p || !ir))) return 1;
    ctx->ir_ov.clear();
    std::vector<int64_t> order((size_t)n);
    for (int64_t k = 0; k < n; k++) order[(size_t)k] = k;
    std::sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return index_5p[x] != index_5p[y] ? index_5p[x] < index_5p[y] : index_3p[x] < index_3p[y]; });
    ctx->ir_ov.reserve((size_t)n * 7);
    for (int64_t q : order) {
        ctx->ir_ov.push_back(index_5p[q]); ctx->ir_ov.push_back(index_3p[q]);
        for (int z = 0; z < 5; z++) ctx->ir_ov.push_back(ir[5 * q + z]);
    }
    return 0;
}

// Fused window attention (Swin W-MSA / SW-MSA core), forward and backward, fp32.
// Reference: WindowAttention.forward src/nsbench/models/swintransformer/swin_transformer.py:123-155
// (dlwpbench twin :122-154):  attn = softmax(q*scale @ k^T + rel_pos_bias[index] + mask) @ v.
// The reference materialises the [B_, heads, N, N] score tensor (1 GB at the dlwpbench default
// N = 2048), gathers the bias through an N x N index tensor and adds an N x N mask per window.
//
// MI355X design: flash-style, S never leaves the CU.  One workgroup = 64 queries of one (window,
// head); keys/values stream through LDS in tiles of 64.  All products run on the exact-f32 MFMA in
// the TRANSPOSED orientation (S^T = K Q^T: rows = keys, columns = queries), so that the score
// accumulator is directly the B operand of O^T = V^T P^T (permuted-k trick, common.hip.h) and the
// softmax statistics of a query live in one lane.  The position bias is looked up from a per-(head,
// window type) LDS copy of the table; its index is additive in the two tokens, idx = ia[q] + ib[k]
// (Swin: relative offsets; Pangu: earth-specific absolute pressure/latitude + relative longitude), so
// two N-vectors replace the N x N index tensor; the shift mask is `label[q] != label[k] ? -100 : 0` from a per-window label
// vector (no N x N mask tensor).  Backward: kernel Q (dQ, dBias, D = rowsum(dO*O)) and kernel KV
// (dK, dV), each recomputing the score tile it needs.
#include <cstdlib>
#include "common.hip.h"
#include "dlwpmi_internal.h"

namespace {

constexpr int QT = 64, KT = 64;
constexpr int LDV = 64 + 4;    // transposed tiles [dd][token]: 16-byte aligned rows

struct WaDev {
    const float* qkv;          // [B_, N, 3, heads, d]
    const float* table;        // [(2Wh-1)(2Ww-1), heads]
    const float* table_t;      // optional: the same table packed as [ntypes][heads][TB] (dlwp_window_attn_pack_table)
    const int* labels;         // [nW, N] or nullptr
    const int *ia, *ib;        // [N] each: bias index of (query q, key k) = ia[q] + ib[k]
    int ntypes;                // bias table is [TB, ntypes, heads]; window type = window index % ntypes
    float* out;                // fwd: [B_, N, heads*d]
    float* lse;                // fwd: [B_, heads, N]
    const float* lse_in;       // bwd: saved log-sum-exp
    const float* gout;         // bwd: [B_, N, heads*d]
    const float* o;            // bwd: forward output
    float* dsum;               // bwd: [B_, heads, N]  D = rowsum(gout * out)
    float* gqkv;               // bwd: [B_, N, 3, heads, d]
    float* gtable;             // bwd: accumulated
    float* slab;               // bwd: optional [workgroups][TB] bias-gradient partials (folded by winattn_fold_kernel)
    int B_, nW, N, heads, d, dp16, TB;
    float scale;
    FastDiv dd;                // division by d
};

__device__ __forceinline__ float wave_col_max(float v) {   // reduce over the 4 lane groups (same r)
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float wave_col_sum(float v) {
    v += __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
}

// 64 token rows x d columns of q / k / v / dO, fetched with all loads in flight (issue) and written to LDS
// later (commit), so that the next tile's HBM latency hides behind the current tile's matrix work.
// Out-of-range rows are clamped for the load and zeroed at commit.  NL = ceil(64 * d / 256) loads per thread.
template <int NL>
struct RowTile {
    float v[NL];
    __device__ __forceinline__ void issue(const float* __restrict__ base, long long row_stride, int row0, int N, int d,
                                          FastDiv dd) {
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const int idx = threadIdx.x + 256 * q;
            const int row = fastdiv(idx, dd), c = idx - row * d;
            const int grow = min(row0 + min(row, 63), N - 1);
            v[q] = base[(long long)grow * row_stride + c];
        }
    }
    // rows[token][dd] (ld LDT) and / or cols[dd][token] (ld LDV); either may be nullptr
    __device__ __forceinline__ void commit(float* rows, int LDT, float* cols, int row0, int N, int d, FastDiv dd,
                                           float scale) const {
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const int idx = threadIdx.x + 256 * q;
            const int row = fastdiv(idx, dd), c = idx - row * d;
            if (row < 64) {
                const float val = row0 + row < N ? v[q] * scale : 0.f;
                if (rows) rows[row * LDT + c] = val;
                if (cols) cols[c * LDV + row] = val;
            }
        }
    }
};

// per-token scalars of a 64-token tile (labels, bias-index parts, lse, D), one per thread of the first wave
struct TokVals {
    int lab, bidx;
    float lse, ds;
};

// this (window type, head)'s slice of the bias table [TB][ntypes][heads] -> LDS.  The slice is a strided gather
// (one cache line per entry): keep eight loads in flight per thread instead of one.
__device__ __forceinline__ void load_table(float* tb, const float* __restrict__ table, int TB, long long tstr,
                                           long long tofs, const float* __restrict__ packed = nullptr) {
    if (packed) {          // contiguous slice of the packed table: coalesced, TB * 4 bytes instead of TB cache lines
        const float* src = packed + tofs * TB;
        for (int i = threadIdx.x; i < TB; i += 256) tb[i] = src[i];
        return;
    }
    for (int i0 = threadIdx.x; i0 < TB; i0 += 256 * 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = table[(long long)min(i0 + 256 * u, TB - 1) * tstr + tofs];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (i0 + 256 * u < TB) tb[i0 + 256 * u] = v[u];
    }
}

// Bias-gradient partials are accumulated in LDS as 64-bit fixed point (2^-40 resolution, |sum| < 2^23):
// gfx950's LDS float atomic add runs at ~0.3 lane-ops/clk/CU, the 64-bit integer add at 5-7 (tools/micro/
// lds_atomics.hip, profiles/README.md), and integer addition makes the partial independent of the order in
// which the waves arrive.  The rounding step (4.5e-13 absolute per term) is below fp32 accumulation error for
// any gradient that matters.
constexpr float FX_SCALE = 1099511627776.f;   // 2^40
__device__ __forceinline__ void fx_add(unsigned long long* acc, float v) {
    atomicAdd(acc, (unsigned long long)(long long)llrintf(v * FX_SCALE));
}
__device__ __forceinline__ float fx_get(unsigned long long v) { return (float)(long long)v * (1.f / FX_SCALE); }

__device__ __forceinline__ f32x4 ldsv(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ int4 ldsi(const int* p) { return *reinterpret_cast<const int4*>(p); }

// score tile of one 16-row chunk against this wave's 16 columns: acc[j] = sum_dd A[16c + 4g + j][dd] * Bfrag[dd][r]
template <int NDB, bool BF>
__device__ __forceinline__ f32x4 score_chunk(const float* rows_a, int LDT, const f32x4 (&bfrag)[NDB], int r, int g) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < NDB; ++cc) acc = mfma16_chunk_p<BF>(ldsv(&rows_a[r * LDT + 16 * cc + 4 * g]), bfrag[cc], acc);
    return acc;
}

// ------------------------------------------------------------------------------------------------
// NBUF = 2: key / value tiles double-buffered (the next tile's LDS commit overlaps the current tile's matrix work, one barrier
// per tile).  NBUF = 1 (windows of at most 128 tokens, i.e. one or two tiles): a single buffer and two barriers per tile --
// the workgroup needs ~40 % less LDS, so twice as many of them are resident per CU, which is what hides latency when every
// workgroup is this short (Swin 7 x 7 = 49, Pangu 2 x 7 x 7 = 98 tokens per window: profiles/r02_winattn_probe.txt).
template <int NDB, int NBUF, bool BF>
__global__ __launch_bounds__(256) void winattn_fwd_kernel(WaDev a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NL = 4 * NDB, LDT = 16 * NDB + 4, DP = 16 * NDB;
    float* Qs = smem;                         // [64][LDT] (scaled)
    float* Kb = Qs + 64 * LDT;                // [NBUF][64][LDT]
    float* Vt = Kb + NBUF * 64 * LDT;         // [NBUF][DP][LDV]   v transposed
    int* klab = reinterpret_cast<int*>(Vt + NBUF * DP * LDV);   // [2][64]
    int* kbs = klab + 128;                                   // [2][64] key part of the bias index
    float* tb = reinterpret_cast<float*>(kbs + 128);         // [TB] bias table of this head
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id(), r = lane & 15, g = lane >> 4;
    const int nqt = (a.N + QT - 1) / QT;
    const int qt = blockIdx.x % nqt, bh = blockIdx.x / nqt, head = bh % a.heads, b = bh / a.heads;
    const int wdw = b % a.nW;
    const long long rs = 3LL * a.heads * a.d;
    const float* qb = a.qkv + (long long)b * a.N * rs + head * a.d;
    const float* kbase = qb + a.heads * a.d;
    const float* vbase = qb + 2 * a.heads * a.d;
    const int* labw = a.labels ? a.labels + (long long)wdw * a.N : nullptr;

    RowTile<NL> tq, tk, tv;
    TokVals kv;
    auto issue_tile = [&](int kt0) {
        tk.issue(kbase, rs, kt0, a.N, a.d, a.dd);
        tv.issue(vbase, rs, kt0, a.N, a.d, a.dd);
        if (tid < KT) {
            const int key = min(kt0 + tid, a.N - 1);
            kv.lab = labw ? labw[key] : 0;
            kv.bidx = a.ib[key];
        }
    };
    auto commit_tile = [&](int kt0, int buf) {
        tk.commit(Kb + buf * 64 * LDT, LDT, nullptr, kt0, a.N, a.d, a.dd, 1.f);
        tv.commit(nullptr, LDT, Vt + buf * DP * LDV, kt0, a.N, a.d, a.dd, 1.f);
        if (tid < KT) { klab[buf * 64 + tid] = kv.lab; kbs[buf * 64 + tid] = kv.bidx; }
    };
    tq.issue(qb, rs, qt * QT, a.N, a.d, a.dd);
    issue_tile(0);
    // zero the tiles once: padded columns (dd >= d) stay zero for the whole kernel
    for (int i = tid; i < ((1 + NBUF) * 64 * LDT + NBUF * DP * LDV) / 4; i += 256) reinterpret_cast<f32x4*>(smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long tofs = (long long)(wdw % a.ntypes) * a.heads + head, tstr = (long long)a.ntypes * a.heads;
    load_table(tb, a.table, a.TB, tstr, tofs, a.table_t);
    const int q = qt * QT + w * 16 + r;                       // this lane's query (column)
    const int qc = q < a.N ? q : a.N - 1;                      // clamped for table indexing
    const int qa = a.ia[qc];
    const int qlab = labw ? labw[qc] : 0;
    __syncthreads();
    tq.commit(Qs, LDT, nullptr, qt * QT, a.N, a.d, a.dd, a.scale);
    commit_tile(0, 0);
    __syncthreads();
    f32x4 qf[NDB];
#pragma unroll
    for (int cc = 0; cc < NDB; ++cc) qf[cc] = ldsv(&Qs[(w * 16 + r) * LDT + 16 * cc + 4 * g]);
    float m = -1e30f, l = 0.f;
    f32x4 oacc[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) oacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};

    int buf = 0;
    for (int kt0 = 0; kt0 < a.N; kt0 += KT, buf = NBUF == 2 ? buf ^ 1 : 0) {
        const bool more = kt0 + KT < a.N;
        if (more) issue_tile(kt0 + KT);
        const float* Kc = Kb + buf * 64 * LDT;
        const float* Vc = Vt + buf * DP * LDV;
        f32x4 s[4];
        float mx = -1e30f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            s[c] = score_chunk<NDB, BF>(Kc + 16 * c * LDT, LDT, qf, r, g);
            const int4 kb4 = ldsi(&kbs[buf * 64 + 16 * c + 4 * g]);
            const int4 kl4 = ldsi(&klab[buf * 64 + 16 * c + 4 * g]);
            const int kbv[4] = {kb4.x, kb4.y, kb4.z, kb4.w}, klv[4] = {kl4.x, kl4.y, kl4.z, kl4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = kt0 + 16 * c + 4 * g + j;
                float v = s[c][j] + tb[qa + kbv[j]];
                if (labw && klv[j] != qlab) v -= 100.f;
                v = key < a.N ? v : -1e30f;
                s[c][j] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = wave_col_max(mx);
        const float m_new = fmaxf(m, mx);
        const float corr = __expf(m - m_new);
        float psum = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float p = __expf(s[c][j] - m_new);
                s[c][j] = p;
                psum += p;
            }
        l = l * corr + wave_col_sum(psum);
        m = m_new;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
#pragma unroll
            for (int j = 0; j < 4; ++j) oacc[db][j] *= corr;
#pragma unroll
            for (int c = 0; c < 4; ++c) oacc[db] = mfma16_chunk_p<BF>(ldsv(&Vc[(16 * db + r) * LDV + 16 * c + 4 * g]), s[c], oacc[db]);
        }
        if (NBUF == 1) __syncthreads();          // every wave is done reading the tile that is about to be overwritten
        if (more) commit_tile(kt0 + KT, NBUF == 2 ? buf ^ 1 : 0);
        __syncthreads();
    }
    if (q < a.N) {
        const float inv = 1.f / l;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int dd = db * 16 + 4 * g + j;
                if (dd < a.d) a.out[((long long)b * a.N + q) * a.heads * a.d + head * a.d + dd] = oacc[db][j] * inv;
            }
        if (g == 0) a.lse[((long long)b * a.heads + head) * a.N + q] = m + __logf(l);
    }
}

// ------------------------------------------------------------------------------------------------
// backward, query side: dQ, dBias, D.  Same tiling as forward.
template <int NDB, int NBUF, bool BF>
__global__ __launch_bounds__(256) void winattn_bwd_q_kernel(WaDev a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NL = 4 * NDB, LDT = 16 * NDB + 4, DP = 16 * NDB;
    float* Kb = smem;                         // [NBUF][64][LDT]
    float* Vb = Kb + NBUF * 64 * LDT;         // [NBUF][64][LDT]
    float* Kt = Vb + NBUF * 64 * LDT;         // [NBUF][DP][LDV]  k transposed (A operand of dQ)
    // q (scaled) and dO rows of the query tile are only needed until their MFMA fragments sit in registers: they
    // borrow the second K / V buffers, which the pipeline first writes at the end of iteration 0 (NBUF = 1: the only
    // buffers, and key tile 0 is committed after the fragments are out)
    float* Qs = Kb + (NBUF - 1) * 64 * LDT;   // [64][LDT]
    float* Gs = Vb + (NBUF - 1) * 64 * LDT;   // [64][LDT]
    int* klab = reinterpret_cast<int*>(Kt + NBUF * DP * LDV);
    int* kbs = klab + 128;
    float* tb = reinterpret_cast<float*>(kbs + 128);     // [TB] bias table
    unsigned long long* gtb = reinterpret_cast<unsigned long long*>(tb + ((a.TB + 1) & ~1));   // [TB] fixed-point partial
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id(), r = lane & 15, g = lane >> 4;
    const int nqt = (a.N + QT - 1) / QT;
    const int qt = blockIdx.x % nqt, bh = blockIdx.x / nqt, head = bh % a.heads, b = bh / a.heads;
    const int wdw = b % a.nW;
    const long long rs = 3LL * a.heads * a.d, os = (long long)a.heads * a.d;
    const float* qb = a.qkv + (long long)b * a.N * rs + head * a.d;
    const float* kbase = qb + a.heads * a.d;
    const float* vbase = qb + 2 * a.heads * a.d;
    const float* gbase = a.gout + (long long)b * a.N * os + head * a.d;
    const int* labw = a.labels ? a.labels + (long long)wdw * a.N : nullptr;

    DLWP_STAMP(0);
    RowTile<NL> tq, tg, tk, tv;
    TokVals kv;
    auto issue_tile = [&](int kt0) {
        tk.issue(kbase, rs, kt0, a.N, a.d, a.dd);
        tv.issue(vbase, rs, kt0, a.N, a.d, a.dd);
        if (tid < KT) {
            const int key = min(kt0 + tid, a.N - 1);
            kv.lab = labw ? labw[key] : 0;
            kv.bidx = a.ib[key];
        }
    };
    auto commit_tile = [&](int kt0, int buf) {
        tk.commit(Kb + buf * 64 * LDT, LDT, Kt + buf * DP * LDV, kt0, a.N, a.d, a.dd, 1.f);
        tv.commit(Vb + buf * 64 * LDT, LDT, nullptr, kt0, a.N, a.d, a.dd, 1.f);
        if (tid < KT) { klab[buf * 64 + tid] = kv.lab; kbs[buf * 64 + tid] = kv.bidx; }
    };
    tq.issue(qb, rs, qt * QT, a.N, a.d, a.dd);
    tg.issue(gbase, os, qt * QT, a.N, a.d, a.dd);
    issue_tile(0);
    DLWP_STAMP(13);
    for (int i = tid; i < (2 * NBUF * 64 * LDT + NBUF * DP * LDV) / 4; i += 256) reinterpret_cast<f32x4*>(smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    DLWP_STAMP(14);
    const long long tofs = (long long)(wdw % a.ntypes) * a.heads + head, tstr = (long long)a.ntypes * a.heads;
    load_table(tb, a.table, a.TB, tstr, tofs, a.table_t);
    DLWP_STAMP(15);
    for (int i = tid; i < a.TB; i += 256) gtb[i] = 0ull;
    DLWP_STAMP(16);
    const int q = qt * QT + w * 16 + r;
    const int qc = q < a.N ? q : a.N - 1;
    const int qa = a.ia[qc];
    const int qlab = labw ? labw[qc] : 0;
    const float lse = a.lse_in[((long long)b * a.heads + head) * a.N + qc];
    float orow[4 * NDB];                      // O[q][g + 4 u]: this lane's share of D = sum_dd dO[q][dd] O[q][dd]
#pragma unroll
    for (int u = 0; u < 4 * NDB; ++u) orow[u] = a.o[((long long)b * a.N + qc) * os + head * a.d + min(g + 4 * u, a.d - 1)];
    DLWP_STAMP(1);
    __syncthreads();
    DLWP_STAMP(2);
    tq.commit(Qs, LDT, nullptr, qt * QT, a.N, a.d, a.dd, a.scale);
    tg.commit(Gs, LDT, nullptr, qt * QT, a.N, a.d, a.dd, 1.f);
    if (NBUF == 2) commit_tile(0, 0);
    __syncthreads();
    DLWP_STAMP(3);
    // D[q] = sum_dd dO[q][dd] * O[q][dd]
    float dpart = 0.f;
#pragma unroll
    for (int u = 0; u < 4 * NDB; ++u)
        if (g + 4 * u < a.d) dpart += Gs[(w * 16 + r) * LDT + g + 4 * u] * orow[u];     // rows q >= N of Gs are zero
    const float D = wave_col_sum(dpart);
    if (q < a.N && g == 0) a.dsum[((long long)b * a.heads + head) * a.N + q] = D;
    f32x4 qf[NDB], gf[NDB], dq[NDB];
#pragma unroll
    for (int cc = 0; cc < NDB; ++cc) {
        qf[cc] = ldsv(&Qs[(w * 16 + r) * LDT + 16 * cc + 4 * g]);
        gf[cc] = ldsv(&Gs[(w * 16 + r) * LDT + 16 * cc + 4 * g]);
        dq[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();                          // Qs / Gs are dead from here on (their storage is tile buffer 1)
    if (NBUF == 1) {                          // ... or the only buffer: key tile 0 goes in now
        commit_tile(0, 0);
        __syncthreads();
    }

    DLWP_STAMP(4);
    int buf = 0;
    for (int kt0 = 0; kt0 < a.N; kt0 += KT, buf = NBUF == 2 ? buf ^ 1 : 0) {
        const bool more = kt0 + KT < a.N;
        if (kt0 == 0) DLWP_STAMP(5);
        if (more) issue_tile(kt0 + KT);
        if (kt0 == 0) DLWP_STAMP(6);
        const float* Kc = Kb + buf * 64 * LDT;
        const float* Vc = Vb + buf * 64 * LDT;
        const float* Ktc = Kt + buf * DP * LDV;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 s = score_chunk<NDB, BF>(Kc + 16 * c * LDT, LDT, qf, r, g);
            const f32x4 dp = score_chunk<NDB, BF>(Vc + 16 * c * LDT, LDT, gf, r, g);   // dP^T = V dO^T
            const int4 kb4 = ldsi(&kbs[buf * 64 + 16 * c + 4 * g]);
            const int4 kl4 = ldsi(&klab[buf * 64 + 16 * c + 4 * g]);
            const int kbv[4] = {kb4.x, kb4.y, kb4.z, kb4.w}, klv[4] = {kl4.x, kl4.y, kl4.z, kl4.w};
            f32x4 ds;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = kt0 + 16 * c + 4 * g + j;
                const int bi = qa + kbv[j];
                float sc = s[j] + tb[bi];
                if (labw && klv[j] != qlab) sc -= 100.f;
                const float p = __expf(sc - lse);
                const float v = (key < a.N && q < a.N) ? p * (dp[j] - D) : 0.f;
                if (key < a.N && q < a.N) fx_add(&gtb[bi], v);
                ds[j] = v;
            }
#pragma unroll
            for (int db = 0; db < NDB; ++db) dq[db] = mfma16_chunk_p<BF>(ldsv(&Ktc[(16 * db + r) * LDV + 16 * c + 4 * g]), ds, dq[db]);
        }
        if (kt0 == 0) DLWP_STAMP(7);
        if (NBUF == 1) __syncthreads();
        if (more) commit_tile(kt0 + KT, NBUF == 2 ? buf ^ 1 : 0);
        if (kt0 == 0) DLWP_STAMP(8);
        __syncthreads();
        if (kt0 == 0) DLWP_STAMP(9);
    }
    DLWP_STAMP(10);
    if (q < a.N) {
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int dd = db * 16 + 4 * g + j;
                if (dd < a.d) a.gqkv[((long long)b * a.N + q) * rs + head * a.d + dd] = dq[db][j] * a.scale;
            }
    }
    DLWP_STAMP(11);
    if (a.slab) {
        // plain coalesced stores; winattn_fold_kernel sums the partials of every (head, window type)
        for (int i = tid; i < a.TB; i += 256) a.slab[(long long)blockIdx.x * a.TB + i] = fx_get(gtb[i]);
    } else {
        for (int i = tid; i < a.TB; i += 256) {
            if (gtb[i] != 0ull) atomic_add_f32(&a.gtable[i * tstr + tofs], fx_get(gtb[i]));
        }
    }
    DLWP_STAMP(12);
}

// gtable[t][type][head] += sum over the workgroups (window b, query tile) with b % ntypes == type
constexpr int FOLD_CH = 32;
__global__ __launch_bounds__(256) void winattn_fold_kernel(const float* __restrict__ slab, float* gtable, int TB, int ntypes,
                                                           int heads, int nqt, int n_items) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int ty = blockIdx.y % ntypes, head = blockIdx.y / ntypes;
    if (t >= TB) return;
    const int i0 = blockIdx.z * FOLD_CH, i1 = min(n_items, i0 + FOLD_CH);
    float acc = 0.f;
#pragma unroll 8
    for (int it = i0; it < i1; ++it) {
        const int m = it / nqt, qt = it - m * nqt;
        const long long wg = ((long long)(ty + ntypes * m) * heads + head) * nqt + qt;
        acc += slab[wg * TB + t];
    }
    atomic_add_f32(&gtable[((long long)t * ntypes + ty) * heads + head], acc);
}

// ------------------------------------------------------------------------------------------------
// backward, key side: dK, dV.  One workgroup = 64 keys (wave = 16 keys as columns), loops over query tiles.
template <int NDB, int NBUF, bool BF>
__global__ __launch_bounds__(256) void winattn_bwd_kv_kernel(WaDev a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NL = 4 * NDB, LDT = 16 * NDB + 4, DP = 16 * NDB;
    float* Qb = smem;                         // [NBUF][64][LDT] scaled q rows of the current query tile
    float* Gb = Qb + NBUF * 64 * LDT;         // [NBUF][64][LDT]
    // this workgroup's k / v rows are only needed until their fragments sit in registers: they borrow the second
    // q / dO buffers, which the pipeline first writes at the end of iteration 0 (NBUF = 1: the only buffers; query tile 0
    // is committed after the fragments are out)
    float* Ks = Qb + (NBUF - 1) * 64 * LDT;   // [64][LDT]
    float* Vs = Gb + (NBUF - 1) * 64 * LDT;   // [64][LDT]
    float* Qt = Gb + NBUF * 64 * LDT;         // [NBUF][DP][LDV]
    float* Gt = Qt + NBUF * DP * LDV;         // [NBUF][DP][LDV]
    float* lses = Gt + NBUF * DP * LDV;       // [2][64]
    float* dss = lses + 128;                  // [2][64]
    int* qlabs = reinterpret_cast<int*>(dss + 128);   // [2][64]
    int* qas = qlabs + 128;                           // [2][64] query part of the bias index
    float* tb = reinterpret_cast<float*>(qas + 128);
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id(), r = lane & 15, g = lane >> 4;
    const int nkt = (a.N + KT - 1) / KT;
    const int kt = blockIdx.x % nkt, bh = blockIdx.x / nkt, head = bh % a.heads, b = bh / a.heads;
    const int wdw = b % a.nW;
    const long long rs = 3LL * a.heads * a.d, os = (long long)a.heads * a.d;
    const float* qb = a.qkv + (long long)b * a.N * rs + head * a.d;
    const float* gbase = a.gout + (long long)b * a.N * os + head * a.d;
    const int* labw = a.labels ? a.labels + (long long)wdw * a.N : nullptr;
    const long long stat = ((long long)b * a.heads + head) * a.N;

    RowTile<NL> tk, tv, tq, tg;
    TokVals qv;
    auto issue_tile = [&](int qt0) {
        tq.issue(qb, rs, qt0, a.N, a.d, a.dd);
        tg.issue(gbase, os, qt0, a.N, a.d, a.dd);
        if (tid < QT) {
            const int qq = min(qt0 + tid, a.N - 1);
            qv.lab = labw ? labw[qq] : 0;
            qv.bidx = a.ia[qq];
            qv.lse = a.lse_in[stat + qq];
            qv.ds = a.dsum[stat + qq];
        }
    };
    auto commit_tile = [&](int qt0, int buf) {
        tq.commit(Qb + buf * 64 * LDT, LDT, Qt + buf * DP * LDV, qt0, a.N, a.d, a.dd, a.scale);
        tg.commit(Gb + buf * 64 * LDT, LDT, Gt + buf * DP * LDV, qt0, a.N, a.d, a.dd, 1.f);
        if (tid < QT) {
            lses[buf * 64 + tid] = qv.lse; dss[buf * 64 + tid] = qv.ds;
            qlabs[buf * 64 + tid] = qv.lab; qas[buf * 64 + tid] = qv.bidx;
        }
    };
    tk.issue(qb + a.heads * a.d, rs, kt * KT, a.N, a.d, a.dd);
    tv.issue(qb + 2 * a.heads * a.d, rs, kt * KT, a.N, a.d, a.dd);
    issue_tile(0);
    for (int i = tid; i < (2 * NBUF * 64 * LDT + 2 * NBUF * DP * LDV) / 4; i += 256) reinterpret_cast<f32x4*>(smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long tofs = (long long)(wdw % a.ntypes) * a.heads + head, tstr = (long long)a.ntypes * a.heads;
    load_table(tb, a.table, a.TB, tstr, tofs, a.table_t);
    const int key = kt * KT + w * 16 + r;                     // this lane's key (column)
    const int kc = key < a.N ? key : a.N - 1;
    const int kbv = a.ib[kc];
    const int klabel = labw ? labw[kc] : 0;
    __syncthreads();
    tk.commit(Ks, LDT, nullptr, kt * KT, a.N, a.d, a.dd, 1.f);
    tv.commit(Vs, LDT, nullptr, kt * KT, a.N, a.d, a.dd, 1.f);
    if (NBUF == 2) commit_tile(0, 0);
    __syncthreads();
    f32x4 kf[NDB], vf[NDB], dk[NDB], dv[NDB];
#pragma unroll
    for (int cc = 0; cc < NDB; ++cc) {
        kf[cc] = ldsv(&Ks[(w * 16 + r) * LDT + 16 * cc + 4 * g]);
        vf[cc] = ldsv(&Vs[(w * 16 + r) * LDT + 16 * cc + 4 * g]);
        dk[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv[cc] = dk[cc];
    }
    __syncthreads();                          // Ks / Vs are dead from here on (their storage is tile buffer 1)
    if (NBUF == 1) {
        commit_tile(0, 0);
        __syncthreads();
    }

    int buf = 0;
    for (int qt0 = 0; qt0 < a.N; qt0 += QT, buf = NBUF == 2 ? buf ^ 1 : 0) {
        const bool more = qt0 + QT < a.N;
        if (more) issue_tile(qt0 + QT);
        const float* Qc = Qb + buf * 64 * LDT;
        const float* Gc = Gb + buf * 64 * LDT;
        const float* Qtc = Qt + buf * DP * LDV;
        const float* Gtc = Gt + buf * DP * LDV;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // S[query 4g+j][key r], dP[query][key]
            const f32x4 s = score_chunk<NDB, BF>(Qc + 16 * c * LDT, LDT, kf, r, g);
            const f32x4 dp = score_chunk<NDB, BF>(Gc + 16 * c * LDT, LDT, vf, r, g);
            const int o4 = buf * 64 + 16 * c + 4 * g;
            const f32x4 ls4 = ldsv(&lses[o4]), dd4 = ldsv(&dss[o4]);
            const int4 ql4 = ldsi(&qlabs[o4]), qa4 = ldsi(&qas[o4]);
            const int qlv[4] = {ql4.x, ql4.y, ql4.z, ql4.w}, qav[4] = {qa4.x, qa4.y, qa4.z, qa4.w};
            f32x4 p, ds;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = qt0 + 16 * c + 4 * g + j;
                float sc = s[j] + tb[qav[j] + kbv];
                if (labw && qlv[j] != klabel) sc -= 100.f;
                const float pv = (q < a.N && key < a.N) ? __expf(sc - ls4[j]) : 0.f;
                p[j] = pv;
                ds[j] = pv * (dp[j] - dd4[j]);
            }
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                dv[db] = mfma16_chunk_p<BF>(ldsv(&Gtc[(16 * db + r) * LDV + 16 * c + 4 * g]), p, dv[db]);    // dV^T += dO^T P
                dk[db] = mfma16_chunk_p<BF>(ldsv(&Qtc[(16 * db + r) * LDV + 16 * c + 4 * g]), ds, dk[db]);   // dK^T += (scale q)^T dS
            }
        }
        if (NBUF == 1) __syncthreads();
        if (more) commit_tile(qt0 + QT, NBUF == 2 ? buf ^ 1 : 0);
        __syncthreads();
    }
    if (key < a.N) {
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int dd = db * 16 + 4 * g + j;
                if (dd < a.d) {
                    float* base = a.gqkv + ((long long)b * a.N + key) * rs + head * a.d + dd;
                    base[a.heads * a.d] = dk[db][j];
                    base[2 * a.heads * a.d] = dv[db][j];
                }
            }
    }
}

int wa_setup(WaDev& a, int B_, int nW, int N, int TB, int ntypes, int heads, int d, float scale, const char* who) {
    DLWP_REQUIRE(B_ > 0 && nW > 0 && B_ % nW == 0 && N > 0 && heads > 0 && d > 0 && TB > 0 && ntypes > 0,
                 DLWP_E_INVALID, "%s: bad shape", who);
    DLWP_REQUIRE(nW % ntypes == 0, DLWP_E_INVALID, "%s: nW (%d) must be a multiple of ntypes (%d)", who, nW, ntypes);
    DLWP_REQUIRE(d <= 64, DLWP_E_UNSUPPORTED, "%s: head_dim %d > 64: use the GEMM form (dlwp_window_softmax_fwd/bwd)", who, d);
    DLWP_REQUIRE(TB <= 9000, DLWP_E_UNSUPPORTED, "%s: bias table slice of %d entries does not fit LDS", who, TB);
    a.B_ = B_; a.nW = nW; a.N = N; a.TB = TB; a.ntypes = ntypes; a.heads = heads; a.d = d; a.scale = scale;
    a.dp16 = round_up(d, 16);
    a.dd = make_fastdiv(d);
    return DLWP_OK;
}

}  // namespace

namespace {
__global__ __launch_bounds__(256) void pack_table_kernel(const float* __restrict__ table, float* __restrict__ packed, int TB, int TH) {
    // packed[th][t] = table[t][th]; a 32 x 32 tile through LDS so that both sides are coalesced
    __shared__ float tile[32][33];
    const int t0 = blockIdx.x * 32, h0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int t = t0 + j, h = h0 + tx;
        tile[j][tx] = (t < TB && h < TH) ? table[(long long)t * TH + h] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int h = h0 + j, t = t0 + tx;
        if (t < TB && h < TH) packed[(long long)h * TB + t] = tile[tx][j];
    }
}
}  // namespace

extern "C" int dlwp_window_attn_pack_table(const float* bias_table, float* packed, int TB, int ntypes, int heads, void* stream) {
    DLWP_REQUIRE(bias_table && packed && TB > 0 && ntypes > 0 && heads > 0, DLWP_E_INVALID, "window_attn_pack_table: bad argument");
    const int TH = ntypes * heads;
    hipLaunchKernelGGL(pack_table_kernel, dim3((TB + 31) / 32, (TH + 31) / 32), dim3(256), 0, (hipStream_t)stream, bias_table,
                       packed, TB, TH);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_window_attn_fwd_packed(const float* qkv, const float* bias_table, const float* packed_table, const int* ia,
                                           const int* ib, const int* labels, float* out, float* lse, int B_, int nW, int N,
                                           int TB, int ntypes, int heads, int d, float scale, void* stream) {
    return dlwp_window_attn_fwd_qrange(qkv, bias_table, packed_table, ia, ib, labels, out, lse, B_, nW, N, TB, ntypes, heads, d, scale,
                                       0, N, stream);
}

// As dlwp_window_attn_fwd_packed, for windows whose tokens outside [q_lo, q_hi) are PADDING that the caller crops afterwards
// (Pangu's pressure-level pad: one of the two planes of every (2,7,7) window): those tokens still act as keys / values, but
// their own attention rows are neither needed (forward) nor do they receive a gradient (backward), so the kernels that can
// skip them do; rows of `out` / `lse` outside the range are then left unwritten.  The tiled kernels compute everything.
extern "C" int dlwp_window_attn_fwd_qrange(const float* qkv, const float* bias_table, const float* packed_table, const int* ia,
                                           const int* ib, const int* labels, float* out, float* lse, int B_, int nW, int N,
                                           int TB, int ntypes, int heads, int d, float scale, int q_lo, int q_hi, void* stream) {
    DLWP_REQUIRE(qkv && bias_table && ia && ib && out && lse, DLWP_E_INVALID, "window_attn_fwd: NULL argument");
    DLWP_REQUIRE(q_lo >= 0 && q_hi <= N && q_lo < q_hi, DLWP_E_INVALID, "window_attn_fwd: query range [%d, %d) outside [0, %d)", q_lo, q_hi, N);
    WaDev a{};
    a.table_t = packed_table;
    int rc = wa_setup(a, B_, nW, N, TB, ntypes, heads, d, scale, "window_attn_fwd");
    if (rc) return rc;
    if (dlwp_winattn_small_applies(N, d, (long long)B_ * heads) && !dlwp_tune_on("WINATTN_TILED"))
        return dlwp_winattn_small_fwd(qkv, bias_table, packed_table, ia, ib, labels, out, lse, B_, nW, N, TB, ntypes, heads, d,
                                      scale, q_lo, q_hi, stream);
    a.qkv = qkv; a.table = bias_table; a.ia = ia; a.ib = ib; a.labels = labels; a.out = out; a.lse = lse;
    const int nbuf = N <= 128 ? 1 : 2;         // short windows: single-buffered tiles, more workgroups per CU
    const size_t lds = sizeof(float) * ((size_t)(1 + nbuf) * 64 * (a.dp16 + 4) + (size_t)nbuf * a.dp16 * LDV + 256 + a.TB);
    const dim3 grid(B_ * heads * ((N + QT - 1) / QT)), block(256);
    const bool bf = dlwp_get_gemm_precision() == 1;      // bf16 matrix arithmetic asked for: bf16 MFMA operands, fp32 everything else
#define WA_FWD_B(NDB, NB, BFV)                                                                                                \
    do {                                                                                                                  \
        if ((rc = dlwp_ensure_lds(reinterpret_cast<const void*>(winattn_fwd_kernel<NDB, NB, BFV>), lds, "window_attn_fwd"))) return rc; \
        dlwp_prof_scope prof((hipStream_t)stream, 2 * 2.0 * B_ * heads * (double)N * N * d, 4.0 * B_ * N * heads * d * 4 + 4.0 * B_ * heads * N, \
                             "winattn_fwd_kernel<%d, %d, %s>", NDB, NB, BFV ? "true" : "false");                             \
        hipLaunchKernelGGL((winattn_fwd_kernel<NDB, NB, BFV>), grid, block, lds, (hipStream_t)stream, a);               \
    } while (0)
#define WA_FWD(NDB)                                                                                                       \
    do {                                                                                                                  \
        if (bf) { if (nbuf == 1) WA_FWD_B(NDB, 1, true); else WA_FWD_B(NDB, 2, true); }                                   \
        else { if (nbuf == 1) WA_FWD_B(NDB, 1, false); else WA_FWD_B(NDB, 2, false); }                                    \
    } while (0)
    switch (a.dp16 / 16) {
        case 1: WA_FWD(1); break;
        case 2: WA_FWD(2); break;
        case 3: WA_FWD(3); break;
        default: WA_FWD(4); break;
    }
#undef WA_FWD_B
#undef WA_FWD
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

// bf16 tensors in the window layout (round 6): qkv [B_, N, 3, heads, d], out / gout [B_, N, heads, d] and the qkv gradient are bf16 arrays
// (statistics, bias table and its gradient stay fp32) -- the projection that makes qkv writes bf16, the output projection reads bf16, and
// the casts between them and the attention kernels disappear (Swin C4: eight cast launches over 20 - 80 MB each per step).  Where
// dlwp_window_attn_io_bf16_supported() says so: windows of at most 64 tokens, head_dim % 4 == 0, bf16 matrix mode, the whole query range.
extern "C" int dlwp_window_attn_io_bf16_supported(int N, int d, int TB, long long pairs) {
    return dlwp_winattn_io_bf16_applies(N, d, TB, pairs) ? 1 : 0;
}
extern "C" int dlwp_window_attn_fwd_bf16(const void* qkv, const float* bias_table, const float* packed_table, const int* ia, const int* ib,
                                         const int* labels, void* out, float* lse, int B_, int nW, int N, int TB, int ntypes, int heads,
                                         int d, float scale, void* stream) {
    DLWP_REQUIRE(qkv && bias_table && ia && ib && out && lse, DLWP_E_INVALID, "window_attn_fwd_bf16: NULL argument");
    WaDev a{};
    int rc = wa_setup(a, B_, nW, N, TB, ntypes, heads, d, scale, "window_attn_fwd_bf16");
    if (rc) return rc;
    DLWP_REQUIRE(dlwp_winattn_io_bf16_applies(N, d, TB, (long long)B_ * heads), DLWP_E_UNSUPPORTED,
                 "window_attn_fwd_bf16: needs windows of at most 64 tokens, head_dim %% 4 == 0 and the bf16 matrix mode (N %d, d %d)", N, d);
    DLWP_REQUIRE((uintptr_t)qkv % 8 == 0 && (uintptr_t)out % 8 == 0, DLWP_E_INVALID, "window_attn_fwd_bf16: 8-byte aligned tensors");
    return dlwp_winattn_small_fwd(static_cast<const float*>(qkv), bias_table, packed_table, ia, ib, labels, static_cast<float*>(out), lse, B_, nW, N,
                                  TB, ntypes, heads, d, scale, 0, N, stream, 1);
}
extern "C" int dlwp_window_attn_bwd_bf16(const void* qkv, const float* bias_table, const float* packed_table, const int* ia, const int* ib,
                                         const int* labels, const void* out, const float* lse, const void* gout, void* gqkv,
                                         float* gbias_table, int B_, int nW, int N, int TB, int ntypes, int heads, int d, float scale,
                                         void* stream) {
    DLWP_REQUIRE(qkv && bias_table && ia && ib && out && lse && gout && gqkv && gbias_table, DLWP_E_INVALID, "window_attn_bwd_bf16: NULL argument");
    WaDev a{};
    int rc = wa_setup(a, B_, nW, N, TB, ntypes, heads, d, scale, "window_attn_bwd_bf16");
    if (rc) return rc;
    DLWP_REQUIRE((uintptr_t)qkv % 8 == 0 && (uintptr_t)out % 8 == 0 && (uintptr_t)gout % 8 == 0 && (uintptr_t)gqkv % 8 == 0, DLWP_E_INVALID,
                 "window_attn_bwd_bf16: 8-byte aligned tensors");
    return dlwp_winattn_small_bwd(static_cast<const float*>(qkv), bias_table, packed_table, ia, ib, labels, static_cast<const float*>(out), lse,
                                  static_cast<const float*>(gout), static_cast<float*>(gqkv), gbias_table, B_, nW, N, TB, ntypes, heads, d, scale,
                                  0, N, stream, 1);
}

extern "C" int dlwp_window_attn_fwd(const float* qkv, const float* bias_table, const int* ia, const int* ib,
                                    const int* labels, float* out, float* lse, int B_, int nW, int N, int TB,
                                    int ntypes, int heads, int d, float scale, void* stream) {
    return dlwp_window_attn_fwd_packed(qkv, bias_table, nullptr, ia, ib, labels, out, lse, B_, nW, N, TB, ntypes, heads, d, scale,
                                       stream);
}

extern "C" int dlwp_window_attn_bwd_packed(const float* qkv, const float* bias_table, const float* packed_table, const int* ia,
                                           const int* ib, const int* labels, const float* out, const float* lse,
                                           const float* gout, float* gqkv, float* gbias_table, float* dsum, float* slab, int B_,
                                           int nW, int N, int TB, int ntypes, int heads, int d, float scale, void* stream) {
    return dlwp_window_attn_bwd_qrange(qkv, bias_table, packed_table, ia, ib, labels, out, lse, gout, gqkv, gbias_table, dsum, slab,
                                       B_, nW, N, TB, ntypes, heads, d, scale, 0, N, stream);
}

// backward of dlwp_window_attn_fwd_qrange: `gout` rows outside [q_lo, q_hi) are taken as zero (they are: the caller crops those
// tokens); the query gradient of those rows is written as zero, their key / value gradients are complete.  The tiled kernels
// read every row of `out` / `lse` / `gout`: with them the forward must have been the tiled one too (same dispatch rule).
extern "C" int dlwp_window_attn_bwd_qrange(const float* qkv, const float* bias_table, const float* packed_table, const int* ia,
                                           const int* ib, const int* labels, const float* out, const float* lse,
                                           const float* gout, float* gqkv, float* gbias_table, float* dsum, float* slab, int B_,
                                           int nW, int N, int TB, int ntypes, int heads, int d, float scale, int q_lo, int q_hi,
                                           void* stream) {
    DLWP_REQUIRE(qkv && bias_table && ia && ib && out && lse && gout && gqkv && gbias_table && dsum, DLWP_E_INVALID,
                 "window_attn_bwd: NULL argument");
    DLWP_REQUIRE(q_lo >= 0 && q_hi <= N && q_lo < q_hi, DLWP_E_INVALID, "window_attn_bwd: query range [%d, %d) outside [0, %d)", q_lo, q_hi, N);
    WaDev a{};
    a.table_t = packed_table;
    int rc = wa_setup(a, B_, nW, N, TB, ntypes, heads, d, scale, "window_attn_bwd");
    if (rc) return rc;
    if (dlwp_winattn_small_applies(N, d, (long long)B_ * heads) && !dlwp_tune_on("WINATTN_TILED"))
        return dlwp_winattn_small_bwd(qkv, bias_table, packed_table, ia, ib, labels, out, lse, gout, gqkv, gbias_table, B_, nW, N,
                                      TB, ntypes, heads, d, scale, q_lo, q_hi, stream);
    a.qkv = qkv; a.table = bias_table; a.ia = ia; a.ib = ib; a.labels = labels; a.o = out; a.lse_in = lse; a.gout = gout; a.gqkv = gqkv;
    a.gtable = gbias_table; a.dsum = dsum; a.slab = slab;
    const int LDT = a.dp16 + 4;
    const int nbuf = N <= 128 ? 1 : 2;
    const size_t lds_q = sizeof(float) * ((size_t)2 * nbuf * 64 * LDT + (size_t)nbuf * a.dp16 * LDV + 256 + 3 * (size_t)((a.TB + 1) & ~1));
    const size_t lds_kv = sizeof(float) * ((size_t)2 * nbuf * 64 * LDT + (size_t)2 * nbuf * a.dp16 * LDV + 512 + a.TB);
    const dim3 grid(B_ * heads * ((N + QT - 1) / QT)), block(256);
    const bool bf = dlwp_get_gemm_precision() == 1;
#define WA_BWD_B(NDB, NB, BFV)                                                                                                   \
    do {                                                                                                                     \
        if ((rc = dlwp_ensure_lds(reinterpret_cast<const void*>(winattn_bwd_q_kernel<NDB, NB, BFV>), lds_q, "window_attn_bwd"))) return rc;  \
        if ((rc = dlwp_ensure_lds(reinterpret_cast<const void*>(winattn_bwd_kv_kernel<NDB, NB, BFV>), lds_kv, "window_attn_bwd"))) return rc; \
        {   /* live accounting: the pair computes the five products of the backward once; q reads qkv, out, gout, writes gq */ \
            dlwp_prof_scope prof((hipStream_t)stream, 2.5 * 2.0 * B_ * heads * (double)N * N * d, 4.0 * B_ * N * heads * d * 6,    \
                                 "winattn_bwd_q_kernel<%d, %d, %s>", NDB, NB, BFV ? "true" : "false");                          \
            hipLaunchKernelGGL((winattn_bwd_q_kernel<NDB, NB, BFV>), grid, block, lds_q, (hipStream_t)stream, a);          \
        }                                                                                                                    \
        {                                                                                                                    \
            dlwp_prof_scope prof((hipStream_t)stream, 2.5 * 2.0 * B_ * heads * (double)N * N * d, 4.0 * B_ * N * heads * d * 6,    \
                                 "winattn_bwd_kv_kernel<%d, %d, %s>", NDB, NB, BFV ? "true" : "false");                         \
            hipLaunchKernelGGL((winattn_bwd_kv_kernel<NDB, NB, BFV>), grid, block, lds_kv, (hipStream_t)stream, a);        \
        }                                                                                                                    \
    } while (0)
#define WA_BWD(NDB)                                                                                                          \
    do {                                                                                                                     \
        if (bf) { if (nbuf == 1) WA_BWD_B(NDB, 1, true); else WA_BWD_B(NDB, 2, true); }                                      \
        else { if (nbuf == 1) WA_BWD_B(NDB, 1, false); else WA_BWD_B(NDB, 2, false); }                                       \
    } while (0)
    switch (a.dp16 / 16) {
        case 1: WA_BWD(1); break;
        case 2: WA_BWD(2); break;
        case 3: WA_BWD(3); break;
        default: WA_BWD(4); break;
    }
#undef WA_BWD_B
#undef WA_BWD
    if (slab) {
        const int nqt = (N + QT - 1) / QT, n_items = (B_ / ntypes) * nqt;
        hipLaunchKernelGGL(winattn_fold_kernel, dim3((TB + 255) / 256, heads * ntypes, (n_items + FOLD_CH - 1) / FOLD_CH),
                           dim3(256), 0, (hipStream_t)stream, slab, gbias_table, TB, ntypes, heads, nqt, n_items);
    }
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_window_attn_bwd(const float* qkv, const float* bias_table, const int* ia, const int* ib,
                                    const int* labels, const float* out, const float* lse, const float* gout,
                                    float* gqkv, float* gbias_table, float* dsum, float* slab, int B_, int nW, int N,
                                    int TB, int ntypes, int heads, int d, float scale, void* stream) {
    return dlwp_window_attn_bwd_packed(qkv, bias_table, nullptr, ia, ib, labels, out, lse, gout, gqkv, gbias_table, dsum, slab, B_,
                                       nW, N, TB, ntypes, heads, d, scale, stream);
}

extern "C" long long dlwp_window_attn_bwd_slab_floats(int B_, int N, int heads, int TB) {
    if (B_ <= 0 || N <= 0 || heads <= 0 || TB <= 0) return -1;
    return (long long)B_ * heads * ((N + QT - 1) / QT) * TB;
}

// ------------------------------------------------------------------------------------------------
// Head dimensions above 64 (deep stages of the nsbench Swin U-Net: the shipped swintransformer.yaml reaches head_dim 192
// and the paper's 4-stage runs 112 -- always on the 8 x 8 / 4 x 4 token maps of the last stages, src/nsbench/configs/model/
// swintransformer.yaml:8-14, scripts/train_commands.txt:105-109): q k^T and p v run as strided-batched MFMA GEMMs
// (dlwp_gemm_batched) around these two row kernels, with the [B_, heads, N, N] scores in HBM (N <= 1024 tokens).
//   fwd: P = softmax(scale S + table[ia[q] + ib[k]][type][head] + mask)        (in place)
//   bwd: dS = P (dP - rowsum(P dP)); gtable += dS; dP <- scale dS              (in place)
namespace {
struct RowSmDev {
    float* s; const float* p; const float* table; float* gtable; const int *ia, *ib, *labels;
    int B_, nW, N, ntypes, heads;
    float scale;
};
template <bool BWD>
__global__ __launch_bounds__(256) void winattn_rows_kernel(RowSmDev a) {
    const int lane = lane_id();
    const long long row = (long long)blockIdx.x * 4 + wave_id();
    const long long nrows = (long long)a.B_ * a.heads * a.N;
    if (row >= nrows) return;
    const int q = (int)(row % a.N);
    const long long bh = row / a.N;
    const int head = (int)(bh % a.heads), b = (int)(bh / a.heads), wdw = b % a.nW;
    const long long tstr = (long long)a.ntypes * a.heads, tofs = (long long)(wdw % a.ntypes) * a.heads + head;
    const int* labw = a.labels ? a.labels + (long long)wdw * a.N : nullptr;
    float* srow = a.s + row * a.N;
    constexpr int NQ = 16;                       // N <= 1024
    float v[NQ];
    if (!BWD) {
        const int qa = a.ia[q], qlab = labw ? labw[q] : 0;
        float m = -1e30f;
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int k = lane + 64 * u;
            v[u] = -1e30f;
            if (k < a.N) {
                float x = srow[k] * a.scale + a.table[(long long)(qa + a.ib[k]) * tstr + tofs];
                if (labw && labw[k] != qlab) x -= 100.f;
                v[u] = x;
                m = fmaxf(m, x);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float l = 0.f;
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            v[u] = lane + 64 * u < a.N ? __expf(v[u] - m) : 0.f;
            l += v[u];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
        const float inv = 1.f / l;
#pragma unroll
        for (int u = 0; u < NQ; ++u)
            if (lane + 64 * u < a.N) srow[lane + 64 * u] = v[u] * inv;
    } else {
        const float* prow = a.p + row * a.N;
        const int qa = a.ia[q];
        float pv[NQ], dot = 0.f;
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int k = lane + 64 * u;
            pv[u] = k < a.N ? prow[k] : 0.f;
            v[u] = k < a.N ? srow[k] : 0.f;
            dot += pv[u] * v[u];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int k = lane + 64 * u;
            if (k < a.N) {
                const float ds = pv[u] * (v[u] - dot);
                atomic_add_f32(&a.gtable[(long long)(qa + a.ib[k]) * tstr + tofs], ds);
                srow[k] = ds * a.scale;
            }
        }
    }
}
}  // namespace

extern "C" int dlwp_window_softmax_fwd(float* s, const float* bias_table, const int* ia, const int* ib, const int* labels,
                                       int B_, int nW, int N, int ntypes, int heads, float scale, void* stream) {
    DLWP_REQUIRE(s && bias_table && ia && ib && B_ > 0 && nW > 0 && B_ % nW == 0 && N > 0 && heads > 0 && ntypes > 0,
                 DLWP_E_INVALID, "window_softmax_fwd: bad argument");
    DLWP_REQUIRE(N <= 1024, DLWP_E_UNSUPPORTED, "window_softmax_fwd: at most 1024 tokens per window (got %d)", N);
    RowSmDev a{s, nullptr, bias_table, nullptr, ia, ib, labels, B_, nW, N, ntypes, heads, scale};
    const long long rows = (long long)B_ * heads * N;
    hipLaunchKernelGGL(winattn_rows_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_window_softmax_bwd(const float* p, float* dp, float* gbias_table, const int* ia, const int* ib, int B_,
                                       int nW, int N, int ntypes, int heads, float scale, void* stream) {
    DLWP_REQUIRE(p && dp && gbias_table && ia && ib && B_ > 0 && nW > 0 && B_ % nW == 0 && N > 0 && heads > 0 && ntypes > 0,
                 DLWP_E_INVALID, "window_softmax_bwd: bad argument");
    DLWP_REQUIRE(N <= 1024, DLWP_E_UNSUPPORTED, "window_softmax_bwd: at most 1024 tokens per window (got %d)", N);
    RowSmDev a{dp, p, nullptr, gbias_table, ia, ib, nullptr, B_, nW, N, ntypes, heads, scale};
    const long long rows = (long long)B_ * heads * N;
    hipLaunchKernelGGL(winattn_rows_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

#ifdef DLWP_STAMPS
extern "C" int dlwp_debug_stamps_winattn(unsigned long long* host_out) {
    DLWP_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dlwp_stamps), sizeof(unsigned long long) * 32));
    return DLWP_OK;
}
#endif

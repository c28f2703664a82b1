// Window attention for SHORT windows (N <= 128 tokens: Swin 7 x 7 = 49, Pangu 2 x 7 x 7 = 98), forward and backward, fp32:
// one WAVE owns one (window, head) and keeps everything in registers.
//
// Reference: WindowAttention.forward src/nsbench/models/swintransformer/swin_transformer.py:123-155 (dlwpbench twin :122-154),
// EarthAttention3D src/dlwpbench/models/panguweather/panguweather.py:166-246:
//   attn = softmax(q * scale @ k^T + bias_table[index] + mask) @ v.
//
// Why a second kernel family next to winattn.hip (64 queries per workgroup, keys streamed through LDS tiles): at these window
// sizes a workgroup of that design spends its life in prologue latency (LDS zero-fill, tile commits, three barriers per 64
// keys) for ~100 matrix instructions, 23-59 % of its 64 x 64 tiles are padding, and its 38-86 KB of LDS keep 1-4 workgroups
// per CU (profiles/r02_winattn_probe.txt: 8436 workgroups, 193 us forward at the Pangu C4 shape = 3 % of the fp32 MFMA rate).
// Here the operands of every product are MFMA fragments loaded straight from global memory (a window's q, k, v are 5-12 KB:
// L1 / L2 resident after the first touch), there is no barrier after the shared bias-table slice is staged, a workgroup is
// four independent waves = four windows of the same (window type, head), and LDS holds the table slice only.
//
// Products, all on v_mfma_f32_16x16x4_f32 through the permuted-k chunk of common.hip.h, tokens padded to NC chunks of 16:
//   S^T[key][q] = K Q^T           A = K row fragment, B = Q row fragment          (accumulator: 4 keys x 1 query per lane)
//   O^T[dd][q]  = V^T P^T         A = V column fragment (4 strided loads), B = the S^T accumulator itself
// so that a query's softmax statistics live in the four lanes that share r = lane & 15.  The backward pass evaluates the score
// tile in both orientations (S^T for dQ and dBias, S for dK and dV; swapping the A and B fragments transposes the accumulator),
// which costs matrix work that this regime has to spare and avoids any transposition through LDS:
//   dP^T = V dO^T,  dS^T = P^T (dP^T - D),  dQ^T += K^T dS^T ;   dP = dO V^T,  dS = P (dP - D),  dV^T += dO^T P,  dK^T += Q^T dS.
// dBias partials are summed per workgroup in LDS (64-bit fixed point, see winattn.hip) and flushed with float atomics.
//
// Families in this file (dispatch: dlwp_winattn_small_fwd / _bwd and the token-layout entries at the end):
//   winattn_small_fwd_kernel<NC, NDB, VEC, BF, TOK, IOBF>   forward, one wave per (window, head); TOK: operands in the token layout
//                                                           through position maps; IOBF: bf16 token tensors
//   winattn_small_bwd_kernel                                backward, register fragments (fp32 mode, or forced)
//   winattn_lds_bwd_kernel<NDB>             (round 3)       backward, LDS-staged, two passes: windows of <= 64 tokens (Swin)
//   winattn_lds_fwd_tok_kernel<NC, NDB, IOBF> (round 4)     forward, LDS-staged, next window's rows prefetched: token-layout operands
//   winattn_lds_bwd1p_kernel<NDB, NW, IOBF> (round 4)       backward, LDS-staged, every score tile evaluated once: 65 - 128 tokens
//                                                           (Pangu); window- or token-layout gradients and operands
#include <algorithm>
#include <type_traits>
#include "common.hip.h"
#include "dlwpmi_internal.h"

namespace {

struct WsDev {
    const float* qkv;          // [B_, N, 3, heads, d]
    const float* table;        // [TB, ntypes, heads]
    const float* table_t;      // optional packed copy [ntypes][heads][TB]
    const int* labels;         // [nW, N] or nullptr
    const int *ia, *ib;        // bias index of (q, k) = ia[q] + ib[k]
    float *out, *lse;          // fwd: [B_, N, heads*d], [B_, heads, N]
    const float *lse_in, *gout, *o;
    float *gqkv, *gtable;
    int B_, nW, N, heads, d, TB, ntypes, M, groups;
    // token-layout mode of the one-pass backward (dlwp_window_attn_bwd_tokens): the upstream gradient and the qkv gradient live in
    // the UNPARTITIONED layout [batch][Ltok][.]; window position n of window `wdw` reads its gout row at token dst_map[wdw][n] and
    // writes its gqkv row at token src_map[wdw][n] (-1: a padded position -- zero upstream gradient / its gradient is the fill's)
    const int *src_map, *dst_map;
    float* gfill;              // [3 heads d]: sum of the qkv gradient over the padded positions (accumulated)
    int Ltok;
    // `fill` != NULL: the OPERANDS live in the token layout too -- qkv [batch][Ltok][3 heads d] read through src_map (padded
    // positions hold fill [3 heads d], the qkv bias), out [batch][Ltok][heads d] written / read through dst_map; no window-layout
    // copy of either exists (dlwp_window_attn_fwd_tokens; dlwp_window_attn_bwd_tokens with fill)
    const float* fill;
    // io_bf16 (token-layout mode with fill): out / gout / gqkv are bf16 arrays (the `float*` fields then point at bf16 storage):
    // the proj and qkv Linear layers around the attention take their operands as bf16 without a cast pass
    int io_bf16;
    int dbg;                   // measurement switches of the one-pass backward (DLWP_WINATTN_DBG): results are wrong when set
    int qc_lo, qc_hi;          // query chunks (of 16 tokens) to compute; the rest are padded positions whose outputs nobody reads
                               // and whose upstream gradient is zero (Pangu: half of every window, dlwp_window_attn_fwd_qrange)
    float scale;
};

__device__ __forceinline__ float col_max4(float v) { v = fmaxf(v, __shfl_xor(v, 16)); return fmaxf(v, __shfl_xor(v, 32)); }
__device__ __forceinline__ float col_sum4(float v) { v += __shfl_xor(v, 16); return v + __shfl_xor(v, 32); }

// four consecutive channels dd0 .. dd0+3 of token `tok` (zero beyond the window / the head dimension)
template <bool VEC>
__device__ __forceinline__ f32x4 row_frag(const float* __restrict__ base, long long stride, int tok, int N, int dd0, int d) {
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tok < N) {
        const float* p = base + (long long)tok * stride + dd0;
        if (VEC) {
            if (dd0 < d) v = *reinterpret_cast<const f32x4*>(p);       // d % 4 == 0: a vector is whole or absent
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (dd0 + s < d) v[s] = p[s];
        }
    }
    return v;
}
// channel dd of the four consecutive tokens tok0 .. tok0+3
__device__ __forceinline__ f32x4 col_frag(const float* __restrict__ base, long long stride, int tok0, int N, int dd, int d) {
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (dd < d) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (tok0 + s < N) v[s] = base[(long long)(tok0 + s) * stride + dd];
    }
    return v;
}

// the same fragments of a bf16 tensor in the window layout (round 6: dlwp_window_attn_fwd_bf16 / _bwd_bf16), RAW: the four bf16 values
// as they are (the bf16 MFMA takes them; the softmax scale goes on the fp32 scores).  base is a bf16 address, stride in elements, d % 4 == 0.
__device__ __forceinline__ s16x4 row_frag_raw(const float* __restrict__ base, long long stride, int tok, int N, int dd0, int d) {
    s16x4 v = s16x4{0, 0, 0, 0};
    if (tok < N && dd0 < d) v = *reinterpret_cast<const s16x4*>(reinterpret_cast<const short*>(base) + (long long)tok * stride + dd0);
    return v;
}
__device__ __forceinline__ s16x4 col_frag_raw(const float* __restrict__ base, long long stride, int tok0, int N, int dd, int d) {
    s16x4 v = s16x4{0, 0, 0, 0};
    if (dd < d) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (tok0 + s < N) v[s] = reinterpret_cast<const short*>(base)[(long long)(tok0 + s) * stride + dd];
    }
    return v;
}

constexpr float FXS = 1099511627776.f;   // 2^40 (fixed-point dBias partials, see winattn.hip)
__device__ __forceinline__ void fx_add_s(unsigned long long* acc, float v) {
    atomicAdd(acc, (unsigned long long)(long long)llrintf(v * FXS));
}

// workgroup -> (window type, head, group of four windows of that type); wave -> window
struct Who {
    int head, ty, b, wdw;
    bool valid;
    long long tofs, tstr;
};
__device__ __forceinline__ Who who_am_i(const WsDev& a) {
    Who w;
    const int grp = blockIdx.x % a.groups, th = blockIdx.x / a.groups;
    w.ty = th % a.ntypes;
    w.head = th / a.ntypes;
    const int m = grp * 4 + wave_id();
    w.valid = m < a.M;
    w.b = w.ty + a.ntypes * min(m, a.M - 1);
    w.wdw = w.b % a.nW;
    w.tofs = (long long)w.ty * a.heads + w.head;
    w.tstr = (long long)a.ntypes * a.heads;
    return w;
}
__device__ __forceinline__ void stage_table(float* tb, const WsDev& a, const Who& w) {
    if (a.table_t) {
        for (int i = threadIdx.x; i < a.TB; i += 256) tb[i] = a.table_t[w.tofs * a.TB + i];
    } else {
        for (int i = threadIdx.x; i < a.TB; i += 256) tb[i] = a.table[(long long)i * w.tstr + w.tofs];
    }
}

// ------------------------------------------------------------------------------------------------ forward
// TOK: operands in the token layout (WsDev::fill): a wave stages its window's two position-map rows in LDS; a fragment row is then
// the token's row of the unpartitioned qkv tensor or, for a padded position, the fill vector; output rows go to the token the
// reverse map names (positions it drops are not written).  Needs d % 4 == 0.
__device__ __forceinline__ f32x4 mfma_bf(const s16x4 a, const s16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ s16x4 pack_bf(const f32x4 v) {
    typedef __bf16 bh4 __attribute__((ext_vector_type(4)));
    const bh4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    return __builtin_bit_cast(s16x4, h);
}
// Token-layout fragments (TOK): a row is the token's row of the unpartitioned qkv tensor or, for a padded position, the fill vector
// (LDS copy `fl` [d] of this head's slice).  IOBF: the tensor is bf16 and the fragment stays RAW bf16 (s16x4: no widening, the
// bf16 MFMA takes it as it is; the softmax scale is applied to the scores instead of to q).  (A branch-free form of these loaders
// -- clamped rows, unconditional fill reads, selects -- was measured: 83 - 87 us against 45 us for the forward at the Pangu C4
// shapes; the wave-per-window forward has no loads in flight across these branches to protect.)
template <bool IOBF> struct tokfrag { typedef f32x4 type; };
template <> struct tokfrag<true> { typedef s16x4 type; };
template <bool IOBF>
__device__ __forceinline__ typename tokfrag<IOBF>::type tok_row(const float* __restrict__ base, long long stride, const float* __restrict__ fl,
                                                                const int* __restrict__ srcl, int tok, int N, int dd0, int d) {
    if constexpr (IOBF) {
        s16x4 v = s16x4{0, 0, 0, 0};
        if (tok < N && dd0 < d) {
            const int s_ = srcl[tok];
            if (s_ >= 0) v = *reinterpret_cast<const s16x4*>(reinterpret_cast<const short*>(base) + (long long)s_ * stride + dd0);
            else v = pack_bf(*reinterpret_cast<const f32x4*>(fl + dd0));
        }
        return v;
    } else {
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tok < N && dd0 < d) {
            const int s_ = srcl[tok];
            v = s_ >= 0 ? *reinterpret_cast<const f32x4*>(base + (long long)s_ * stride + dd0) : *reinterpret_cast<const f32x4*>(fl + dd0);
        }
        return v;
    }
}
template <bool IOBF>
__device__ __forceinline__ typename tokfrag<IOBF>::type tok_col(const float* __restrict__ base, long long stride, const float* __restrict__ fl,
                                                                const int* __restrict__ srcl, int tok0, int N, int dd, int d) {
    if constexpr (IOBF) {
        s16x4 v = s16x4{0, 0, 0, 0};
        if (dd < d) {
            const short fb = __builtin_bit_cast(short, (__bf16)fl[dd]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (tok0 + s < N) {
                    const int s_ = srcl[tok0 + s];
                    v[s] = s_ >= 0 ? reinterpret_cast<const short*>(base)[(long long)s_ * stride + dd] : fb;
                }
        }
        return v;
    } else {
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (dd < d) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (tok0 + s < N) {
                    const int s_ = srcl[tok0 + s];
                    v[s] = s_ >= 0 ? base[(long long)s_ * stride + dd] : fl[dd];
                }
        }
        return v;
    }
}
template <int NC, int NDB, bool VEC, bool BF, bool TOK = false, bool IOBF = false>
__global__ __launch_bounds__(256) void winattn_small_fwd_kernel(WsDev a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tb = smem;                                         // [TB] bias-table slice of this (type, head)
    const int lane = lane_id(), r = lane & 15, g = lane >> 4;
    const Who w = who_am_i(a);
    stage_table(tb, a, w);
    int* srcl = reinterpret_cast<int*>(tb + ((a.TB + 3) & ~3)) + 256 * wave_id();      // TOK: [128] source rows | [128] destination rows
    int* dstl = srcl + 128;
    float* fills = reinterpret_cast<float*>(reinterpret_cast<int*>(tb + ((a.TB + 3) & ~3)) + 4 * 256);      // TOK: [3][32] fill of this head
    if (TOK) {
        if (w.valid) {
            for (int t = lane; t < a.N; t += 64) {
                srcl[t] = a.src_map[(long long)w.wdw * a.N + t];
                dstl[t] = a.dst_map[(long long)w.wdw * a.N + t];
            }
        }
        if (threadIdx.x < 96) {
            const int part = threadIdx.x >> 5, c = threadIdx.x & 31;
            fills[threadIdx.x] = c < a.d ? a.fill[part * a.heads * a.d + w.head * a.d + c] : 0.f;
        }
    }
    __syncthreads();
    if (!w.valid) return;
    const int N = a.N, d = a.d;
    const long long rs = 3LL * a.heads * d, os = (long long)a.heads * d;
    // IOBF: the three bases are bf16 addresses carried in float pointers (frag_row / frag_col index them in bf16 elements)
    const long long qoff = TOK ? (long long)(w.b / a.nW) * a.Ltok * rs + w.head * d : (long long)w.b * N * rs + w.head * d;
    const float* qb = IOBF ? reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(a.qkv) + qoff) : a.qkv + qoff;
    const float* kb = IOBF ? reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(qb) + a.heads * d) : qb + a.heads * d;
    const float* vb = IOBF ? reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(qb) + 2 * a.heads * d) : qb + 2 * a.heads * d;
    const float* fq = fills;            // LDS copies (TOK only)
    const float* fk = fills + 32;
    const float* fv = fills + 64;
    const int* labw = a.labels ? a.labels + (long long)w.wdw * N : nullptr;

    typedef typename tokfrag<IOBF>::type frag_t;             // raw bf16 fragments when the tensors are bf16 (either layout)
    frag_t kf[NC][NDB], vt[NDB][NC];
    int kbi[NC][4], klb[NC][4];
#pragma unroll
    for (int kc = 0; kc < NC; ++kc) {
#pragma unroll
        for (int cc = 0; cc < NDB; ++cc) {
            if constexpr (TOK) kf[kc][cc] = tok_row<IOBF>(kb, rs, fk, srcl, 16 * kc + r, N, 16 * cc + 4 * g, d);
            else if constexpr (IOBF) kf[kc][cc] = row_frag_raw(kb, rs, 16 * kc + r, N, 16 * cc + 4 * g, d);
            else kf[kc][cc] = row_frag<VEC>(kb, rs, 16 * kc + r, N, 16 * cc + 4 * g, d);
        }
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            if constexpr (TOK) vt[db][kc] = tok_col<IOBF>(vb, rs, fv, srcl, 16 * kc + 4 * g, N, 16 * db + r, d);
            else if constexpr (IOBF) vt[db][kc] = col_frag_raw(vb, rs, 16 * kc + 4 * g, N, 16 * db + r, d);
            else vt[db][kc] = col_frag(vb, rs, 16 * kc + 4 * g, N, 16 * db + r, d);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = min(16 * kc + 4 * g + j, N - 1);
            kbi[kc][j] = a.ib[key];
            klb[kc][j] = labw ? labw[key] : 0;
        }
    }
    if (TOK) {          // token-layout entry: lse arrives uninitialised; rows outside the computed chunks read as zero in the backward
        for (int t = lane; t < N; t += 64)
            if ((t >> 4) < a.qc_lo || (t >> 4) >= a.qc_hi) a.lse[((long long)w.b * a.heads + w.head) * N + t] = 0.f;
    }
#pragma unroll 1
    for (int qc = a.qc_lo; qc < a.qc_hi; ++qc) {
        if (16 * qc >= N) break;
        const int q = 16 * qc + r, qcl = min(q, N - 1);
        const int qa = a.ia[qcl], qlab = labw ? labw[qcl] : 0;
        constexpr bool RAW = IOBF;
        frag_t qf[NDB];
#pragma unroll
        for (int cc = 0; cc < NDB; ++cc) {
            if constexpr (TOK) qf[cc] = tok_row<IOBF>(qb, rs, fq, srcl, q, N, 16 * cc + 4 * g, d);
            else if constexpr (IOBF) qf[cc] = row_frag_raw(qb, rs, q, N, 16 * cc + 4 * g, d);
            else qf[cc] = row_frag<VEC>(qb, rs, q, N, 16 * cc + 4 * g, d);
            if constexpr (!RAW) {
#pragma unroll
                for (int s = 0; s < 4; ++s) qf[cc][s] *= a.scale;
            }
        }
        const float sscale = RAW ? a.scale : 1.f;          // raw bf16 q: the scale goes on the fp32 scores
        f32x4 s[NC];
        float mx = -1e30f;
#pragma unroll
        for (int kc = 0; kc < NC; ++kc) {
            s[kc] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int cc = 0; cc < NDB; ++cc) {
                if constexpr (RAW) s[kc] = mfma_bf(kf[kc][cc], qf[cc], s[kc]);
                else s[kc] = mfma16_chunk_p<BF>(kf[kc][cc], qf[cc], s[kc]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = s[kc][j] * sscale + tb[qa + kbi[kc][j]];
                if (labw && klb[kc][j] != qlab) v -= 100.f;
                v = 16 * kc + 4 * g + j < N ? v : -1e30f;
                s[kc][j] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = col_max4(mx);
        float l = 0.f;
#pragma unroll
        for (int kc = 0; kc < NC; ++kc)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float p = __expf(s[kc][j] - mx);
                s[kc][j] = p;
                l += p;
            }
        l = col_sum4(l);
        const float inv = 1.f / l;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < NC; ++kc) {
                if constexpr (RAW) o = mfma_bf(vt[db][kc], pack_bf(s[kc]), o);
                else o = mfma16_chunk_p<BF>(vt[db][kc], s[kc], o);
            }
            const int dd = 16 * db + 4 * g;
            const int orow = TOK ? (q < N ? dstl[q] : -1) : q;
            if (IOBF) {
                if (q < N && orow >= 0 && dd < d) {
                    typedef __bf16 bh4 __attribute__((ext_vector_type(4)));
                    __bf16* dst = reinterpret_cast<__bf16*>(a.out) + (TOK ? (long long)(w.b / a.nW) * a.Ltok + orow : (long long)w.b * N + q) * os + w.head * d + dd;
                    *reinterpret_cast<bh4*>(dst) = bh4{(__bf16)(o[0] * inv), (__bf16)(o[1] * inv), (__bf16)(o[2] * inv), (__bf16)(o[3] * inv)};
                }
            } else if (q < N && orow >= 0) {
                float* dst = a.out + (TOK ? (long long)(w.b / a.nW) * a.Ltok + orow : (long long)w.b * N + q) * os + w.head * d + dd;
                if (VEC) {
                    if (dd < d) *reinterpret_cast<f32x4*>(dst) = f32x4{o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (dd + j < d) dst[j] = o[j] * inv;
                }
            }
        }
        if (g == 0 && q < N) a.lse[((long long)w.b * a.heads + w.head) * N + q] = mx + __logf(l);
    }
}

// ------------------------------------------------------------------------------------------------ backward
// Two passes over the (query chunk, key chunk) pairs, both with runtime loops so that only one pair's fragments are live
// (a single fused pass holds dK^T and dV^T of every key chunk and lands at 256 VGPRs):
//   pass Q: for each query chunk, over the key chunks: S^T, dP^T -> dS^T -> dQ^T and dBias; D[q] goes to a wave-private LDS row
//   pass K: for each key chunk, over the query chunks: S, dP -> P, dS -> dV^T, dK^T
template <int NDB, bool VEC, bool BF>
__global__ __launch_bounds__(256) void winattn_small_bwd_kernel(WsDev a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tb = smem;                                                                        // [TB]
    unsigned long long* gtb = reinterpret_cast<unsigned long long*>(tb + ((a.TB + 1) & ~1));   // [TB] fixed-point dBias partial
    float* Dw = reinterpret_cast<float*>(gtb + a.TB) + 128 * wave_id();                       // [128] D of this wave's window
    const int lane = lane_id(), r = lane & 15, g = lane >> 4;
    const Who w = who_am_i(a);
    stage_table(tb, a, w);
    for (int i = threadIdx.x; i < a.TB; i += 256) gtb[i] = 0ull;
    __syncthreads();
    const int N = a.N, d = a.d, NCr = (N + 15) / 16;
    const long long rs = 3LL * a.heads * d, os = (long long)a.heads * d;
    if (w.valid) {
        const float* qb = a.qkv + (long long)w.b * N * rs + w.head * d;
        const float* kb = qb + a.heads * d;
        const float* vb = qb + 2 * a.heads * d;
        const float* gb = a.gout + (long long)w.b * N * os + w.head * d;
        const float* ob = a.o + (long long)w.b * N * os + w.head * d;
        float* gq = a.gqkv + (long long)w.b * N * rs + w.head * d;
        const int* labw = a.labels ? a.labels + (long long)w.wdw * N : nullptr;
        const long long stat = ((long long)w.b * a.heads + w.head) * N;

        // Both passes prefetch the next pair's fragments into registers while the current pair runs on the matrix cores:
        // the operands come from L1 / L2 (a window is 5-12 KB per matrix), whose latency one wave cannot hide otherwise.
        struct KeySide {       // everything pass Q needs from key chunk kc
            f32x4 kf[NDB], vf[NDB], kt[NDB];
            int kb[4], kl[4];
        };
        auto load_keys = [&](int kc, KeySide& t) {
#pragma unroll
            for (int cc = 0; cc < NDB; ++cc) {
                t.kf[cc] = row_frag<VEC>(kb, rs, 16 * kc + r, N, 16 * cc + 4 * g, d);
                t.vf[cc] = row_frag<VEC>(vb, rs, 16 * kc + r, N, 16 * cc + 4 * g, d);
                t.kt[cc] = col_frag(kb, rs, 16 * kc + 4 * g, N, 16 * cc + r, d);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int keyc = min(16 * kc + 4 * g + j, N - 1);
                t.kb[j] = a.ib[keyc];
                t.kl[j] = labw ? labw[keyc] : 0;
            }
        };
        struct QuerySide {     // everything pass K needs from query chunk qc
            f32x4 qf[NDB], gf[NDB], qt[NDB], gt[NDB];
            int qa[4], ql[4];
            float lse[4];
        };
        auto load_queries = [&](int qc, QuerySide& t) {
#pragma unroll
            for (int cc = 0; cc < NDB; ++cc) {
                t.qf[cc] = row_frag<VEC>(qb, rs, 16 * qc + r, N, 16 * cc + 4 * g, d);
                t.gf[cc] = row_frag<VEC>(gb, os, 16 * qc + r, N, 16 * cc + 4 * g, d);
                t.qt[cc] = col_frag(qb, rs, 16 * qc + 4 * g, N, 16 * cc + r, d);
                t.gt[cc] = col_frag(gb, os, 16 * qc + 4 * g, N, 16 * cc + r, d);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int qqc = min(16 * qc + 4 * g + j, N - 1);
                t.qa[j] = a.ia[qqc];
                t.ql[j] = labw ? labw[qqc] : 0;
                t.lse[j] = a.lse_in[stat + qqc];
            }
        };

        // ---- pass Q  (query chunks outside [qc_lo, qc_hi) carry no upstream gradient: their dQ rows are zero)
        const int qlo = a.qc_lo, qhi = min(a.qc_hi, NCr);
        for (int qc = 0; qc < NCr; ++qc) {
            if (qc >= qlo && qc < qhi) continue;
            const int q = 16 * qc + r;
            if (q < N) {
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const int dd = 16 * db + 4 * g;
                    float* dst = gq + (long long)q * rs + dd;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (dd + j < d) dst[j] = 0.f;
                }
            }
        }
#pragma unroll 1
        for (int qc = qlo; qc < qhi; ++qc) {
            const int q = 16 * qc + r, qcl = min(q, N - 1);
            const int qa = a.ia[qcl], ql = labw ? labw[qcl] : 0;
            const float lse = a.lse_in[stat + qcl];
            KeySide cur, nxt;
            load_keys(0, cur);
            f32x4 qf[NDB], gf[NDB], dq[NDB];
            float dpart = 0.f;
#pragma unroll
            for (int cc = 0; cc < NDB; ++cc) {
                qf[cc] = row_frag<VEC>(qb, rs, q, N, 16 * cc + 4 * g, d);
                gf[cc] = row_frag<VEC>(gb, os, q, N, 16 * cc + 4 * g, d);
                const f32x4 of = row_frag<VEC>(ob, os, q, N, 16 * cc + 4 * g, d);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    dpart += gf[cc][s] * of[s];
                    qf[cc][s] *= a.scale;
                }
                dq[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const float D = col_sum4(dpart);                          // D[q] = sum_dd dO[q][dd] O[q][dd]
            if (g == 0) Dw[q] = D;                                     // q < 128 always
#pragma unroll 1
            for (int kc = 0; kc < NCr; ++kc) {
                if (kc + 1 < NCr) load_keys(kc + 1, nxt);
                f32x4 sT = f32x4{0.f, 0.f, 0.f, 0.f}, dpT = sT;
#pragma unroll
                for (int cc = 0; cc < NDB; ++cc) {
                    sT = mfma16_chunk_p<BF>(cur.kf[cc], qf[cc], sT);
                    dpT = mfma16_chunk_p<BF>(cur.vf[cc], gf[cc], dpT);
                }
                f32x4 dsT;
#pragma unroll
                for (int j = 0; j < 4; ++j) {                          // rows = keys 16 kc + 4g + j, column = query r
                    const int key = 16 * kc + 4 * g + j;
                    const int bi = qa + cur.kb[j];
                    float sc = sT[j] + tb[bi];
                    if (labw && cur.kl[j] != ql) sc -= 100.f;
                    const bool ok = key < N && q < N;
                    const float v = ok ? __expf(sc - lse) * (dpT[j] - D) : 0.f;
                    if (ok) fx_add_s(&gtb[bi], v);
                    dsT[j] = v;
                }
#pragma unroll
                for (int db = 0; db < NDB; ++db) dq[db] = mfma16_chunk_p<BF>(cur.kt[db], dsT, dq[db]);   // dQ^T[dd][q] += K^T dS^T
                cur = nxt;
            }
            if (q < N) {
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const int dd = 16 * db + 4 * g;
                    float* dst = gq + (long long)q * rs + dd;
                    if (VEC) {
                        if (dd < d) *reinterpret_cast<f32x4*>(dst) = f32x4{dq[db][0] * a.scale, dq[db][1] * a.scale, dq[db][2] * a.scale, dq[db][3] * a.scale};
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (dd + j < d) dst[j] = dq[db][j] * a.scale;
                    }
                }
            }
        }
        // ---- pass K (Dw was written by this wave: LDS operations of a wave complete in order)
#pragma unroll 1
        for (int kc = 0; kc < NCr; ++kc) {
            const int key = 16 * kc + r, keyc = min(key, N - 1);
            const int kbi = a.ib[keyc], kl = labw ? labw[keyc] : 0;
            QuerySide cur, nxt;
            load_queries(qlo, cur);
            f32x4 kf[NDB], vf[NDB], dk[NDB], dv[NDB];
#pragma unroll
            for (int cc = 0; cc < NDB; ++cc) {
                kf[cc] = row_frag<VEC>(kb, rs, key, N, 16 * cc + 4 * g, d);
                vf[cc] = row_frag<VEC>(vb, rs, key, N, 16 * cc + 4 * g, d);
                dk[cc] = dv[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll 1
            for (int qc = qlo; qc < qhi; ++qc) {
                if (qc + 1 < qhi) load_queries(qc + 1, nxt);
                f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll
                for (int cc = 0; cc < NDB; ++cc) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) { cur.qf[cc][t] *= a.scale; cur.qt[cc][t] *= a.scale; }
                    s = mfma16_chunk_p<BF>(cur.qf[cc], kf[cc], s);          // rows = queries 16 qc + 4g + j, column = key r
                    dp = mfma16_chunk_p<BF>(cur.gf[cc], vf[cc], dp);
                }
                f32x4 p, ds;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int qq = 16 * qc + 4 * g + j, qqc = min(qq, N - 1);
                    float sc = s[j] + tb[cur.qa[j] + kbi];
                    if (labw && cur.ql[j] != kl) sc -= 100.f;
                    const float pv = (qq < N && key < N) ? __expf(sc - cur.lse[j]) : 0.f;
                    p[j] = pv;
                    ds[j] = pv * (dp[j] - Dw[qqc]);
                }
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    dv[db] = mfma16_chunk_p<BF>(cur.gt[db], p, dv[db]);     // dV^T += dO^T P
                    dk[db] = mfma16_chunk_p<BF>(cur.qt[db], ds, dk[db]);    // dK^T += (scale Q)^T dS
                }
                cur = nxt;
            }
            if (key < N) {
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const int dd = 16 * db + 4 * g;
                    float* dst = gq + (long long)key * rs + dd;
                    if (VEC) {
                        if (dd < d) {
                            *reinterpret_cast<f32x4*>(dst + a.heads * d) = dk[db];
                            *reinterpret_cast<f32x4*>(dst + 2 * a.heads * d) = dv[db];
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (dd + j < d) { dst[a.heads * d + j] = dk[db][j]; dst[2 * a.heads * d + j] = dv[db][j]; }
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.TB; i += 256)
        if (gtb[i] != 0ull) atomic_add_f32(&a.gtable[(long long)i * w.tstr + w.tofs], (float)(long long)gtb[i] * (1.f / FXS));
}

// ------------------------------------------------------------------------------------------------ LDS-staged bf16 family
// (round 3) The register-fragment kernels above fetch every MFMA fragment of every 16 x 16 tile pair from global memory: a
// row fragment touches 16 rows 2304 B apart, a column fragment is four strided dword loads, and one wave walks 28 - 49 tile
// pairs twice in the backward pass -- the kernel is bound by L1 transactions, not by arithmetic (Pangu C4 layer 1: 638 us for
// 18 GFLOP).  In bf16 matrix mode the operands are rounded to bf16 anyway, so here a WORKGROUP owns one (window, head):
// its four waves stage q (pre-scaled), k, v, dO of the window ONCE, coalesced (128 B per token and matrix), as bf16 rows of
// 72 B in LDS (<= 37 KB), together with D = rowsum(dO o), the row statistics, the bias-index vectors and the labels; the
// tile walk then reads row fragments as one ds_read_b64 and column fragments as four ds_read_u16.  The arithmetic, its order
// and the roundings are those of the register-fragment kernels in bf16 mode.  Waves split the query chunks (pass Q) and the key
// chunks (pass K).  BACKWARD only: the forward pass reads each operand once per query chunk and the same construction measured
// SLOWER there (93 vs 63 us at the C4 layer-1 shape: the staging is 265 MB of HBM traffic per launch either way), so the
// forward stays on the register-fragment kernel.  Where the backward time goes (timing switches, same shape, 272 us): staging
// + zero fill + flush 69 us, pass Q 102 us (49 of them the fixed-point LDS atomics of the bias gradient), pass K 77 us -- the
// passes are bound by VALU bookkeeping per 16 x 16 tile (index vectors, masks, exponentials, fragment packing), ~2 k SIMD cycles
// per tile pair for 6 - 8 MFMAs.
typedef unsigned short bf16_t;
constexpr int LDB = 36;        // bf16 elements per staged row (32 + 4: 72 B, 8-byte aligned, rows 18 banks apart)

// row fragment: four consecutive channels of one staged row; column fragment: one channel of four consecutive rows
// LB = row pitch in bf16 elements: LDB for head dims up to 32; LDB48 (48 + 4: 104-byte rows, still 8-byte aligned) for head dims up to 48
// (round 6: Swin C4 stage 1 has 192 channels on 4 heads -- the two-pass backward and the wave-per-window forward take it; the one-pass and
// token-layout kernels stay at 32)
constexpr int LDB48 = 52;
template <int LB = LDB>
__device__ __forceinline__ s16x4 lds_row(const bf16_t* m, int row, int c0) { return *reinterpret_cast<const s16x4*>(m + row * LB + c0); }
// Four consecutive rows row0 .. row0 + 3 of column c -- the A operand of the transposed products (dV^T += dO^T P, dK^T += Q^T dS,
// dQ^T = K^T dS^T, O^T = V^T P^T), always called with row0 = (a 16-row chunk) + 4 g and c = 16 db + r.  Round 6: ONE transposing
// read (ds_read_b64_tr_b16) instead of four 2-byte reads and three packs per fragment: within a 16-lane group lane r hands the
// hardware the 8-byte piece [row0 + (r >> 2)][16 db + 4 (r & 3) .. + 3] of the 4 x 16 block and receives its column r.  All 64
// lanes must be active (every call site sits under wave-uniform control flow); rows are 8-byte aligned (pitch 72 / 104 bytes).
template <int LB = LDB>
__device__ __forceinline__ s16x4 lds_col(const bf16_t* m, int row0, int c) {
    static_assert((LB * 2) % 8 == 0, "8-byte aligned rows");
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int r = c & 15;
    const bf16_t* p = m + (row0 + (r >> 2)) * LB + (c - r) + 4 * (r & 3);
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
}
// the same with a run-time row pitch (elements; a multiple of 4)
__device__ __forceinline__ s16x4 lds_col_pitch(const bf16_t* m, int pitch, int row0, int c) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int r = c & 15;
    const bf16_t* p = m + (row0 + (r >> 2)) * pitch + (c - r) + 4 * (r & 3);
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
}

struct LdsWin {
    bf16_t *Q, *K, *V, *G;
    float *D, *lse, *tb;
    int *ia, *ib, *lab;
    unsigned long long* gtb;
};
template <int LB = LDB>
__device__ __forceinline__ LdsWin lds_carve(void* smem, int NR, int TB, bool bwd) {
    LdsWin L;
    bf16_t* h = reinterpret_cast<bf16_t*>(smem);
    L.Q = h; L.K = L.Q + NR * LB; L.V = L.K + NR * LB; L.G = L.V + NR * LB;
    float* f = reinterpret_cast<float*>(L.G + (bwd ? NR * LB : 0));
    L.D = f; L.lse = f + NR;
    L.ia = reinterpret_cast<int*>(f + 2 * NR); L.ib = L.ia + NR; L.lab = L.ib + NR;
    L.tb = reinterpret_cast<float*>(L.lab + NR);
    L.gtb = reinterpret_cast<unsigned long long*>(L.tb + ((TB + 1) & ~1));
    return L;
}
static size_t lds_bytes(int NR, int TB, bool bwd, int lb = LDB) {
    return (size_t)(bwd ? 4 : 3) * NR * lb * 2 + (size_t)5 * NR * 4 + (size_t)((TB + 1) & ~1) * 4 + (bwd ? (size_t)TB * 8 : 0);
}
// A workgroup owns one (window type, head) and walks the windows m = grp, grp + groups, ... of that type one after the other:
// the bias-table slice is staged once and the bias-gradient partials are flushed once per workgroup (one workgroup per window
// meant 2548 global float atomics per window on 37-way shared addresses at the Pangu C4 shape: the flush, not the arithmetic,
// set the kernel's time).
__device__ __forceinline__ Who who_lds(const WsDev& a, int& grp) {
    Who w;
    grp = blockIdx.x % a.groups;
    const int th = blockIdx.x / a.groups;
    w.ty = th % a.ntypes;
    w.head = th / a.ntypes;
    w.valid = true;
    w.b = w.ty;
    w.wdw = 0;
    w.tofs = (long long)w.ty * a.heads + w.head;
    w.tstr = (long long)a.ntypes * a.heads;
    return w;
}
__device__ __forceinline__ void who_window(const WsDev& a, Who& w, int m) {
    w.b = w.ty + a.ntypes * m;
    w.wdw = w.b % a.nW;
}
// stage the window: q (x scale), k, v [, dO, D = rowsum(dO o), lse] + index vectors; TPT threads per token (16 B each): 8 for head dims up to
// 32, 16 for head dims up to 48 (twelve of them carry channels)
template <bool BWD, int LB = LDB, int TPT = 8>
__device__ __forceinline__ void lds_stage(const WsDev& a, const Who& w, const LdsWin& L, int NR) {
    const int N = a.N, d = a.d, tid = threadIdx.x, ch = tid & (TPT - 1);
    const long long rs = 3LL * a.heads * d, os = (long long)a.heads * d;
    const float* qb = a.qkv + (long long)w.b * N * rs + w.head * d;
    const float* gb = BWD ? a.gout + (long long)w.b * N * os + w.head * d : nullptr;
    const float* ob = BWD ? a.o + (long long)w.b * N * os + w.head * d : nullptr;
    const long long stat = ((long long)w.b * a.heads + w.head) * N;
    for (int tok = tid / TPT; tok < NR; tok += 256 / TPT) {
        const bool ok = tok < N && 4 * ch < d;
        const bool st = 4 * ch < LB - 4;                    // this lane's chunk exists in the row (TPT = 16: chunks 12 .. 15 do not)
        const int tc = tok < N ? tok : N - 1, cc = 4 * ch < d ? 4 * ch : 0;        // clamped: unconditional loads
        f32x4 q, k, v, g = {0.f, 0.f, 0.f, 0.f}, o = g;
        if (a.io_bf16) {                                    // bf16 tensors (window layout, round 6): 8-byte pieces, widened for the scale and D
            typedef __bf16 bh4 __attribute__((ext_vector_type(4)));
            auto ld = [](const float* base, long long off) {
                const bh4 h = *reinterpret_cast<const bh4*>(reinterpret_cast<const __bf16*>(base) + off);
                return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
            };
            const long long qo = ((long long)w.b * N + tc) * rs + w.head * d + cc, go = ((long long)w.b * N + tc) * os + w.head * d + cc;
            q = ld(a.qkv, qo); k = ld(a.qkv, qo + a.heads * d); v = ld(a.qkv, qo + 2 * a.heads * d);
            if (BWD) { g = ld(a.gout, go); o = ld(a.o, go); }
        } else {
            const float* row = qb + (long long)tc * rs + cc;
            q = *reinterpret_cast<const f32x4*>(row);
            k = *reinterpret_cast<const f32x4*>(row + a.heads * d);
            v = *reinterpret_cast<const f32x4*>(row + 2 * a.heads * d);
            if (BWD) {
                g = *reinterpret_cast<const f32x4*>(gb + (long long)tc * os + cc);
                o = *reinterpret_cast<const f32x4*>(ob + (long long)tc * os + cc);
            }
        }
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (!ok) { q = z; k = z; v = z; g = z; o = z; }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) q[s2] *= a.scale;
        if (st) {
            *reinterpret_cast<s16x4*>(L.Q + tok * LB + 4 * ch) = pack_bf(q);
            *reinterpret_cast<s16x4*>(L.K + tok * LB + 4 * ch) = pack_bf(k);
            *reinterpret_cast<s16x4*>(L.V + tok * LB + 4 * ch) = pack_bf(v);
        }
        if (BWD) {
            if (st) *reinterpret_cast<s16x4*>(L.G + tok * LB + 4 * ch) = pack_bf(g);
            float dp = g[0] * o[0] + g[1] * o[1] + g[2] * o[2] + g[3] * o[3];
            dp += __shfl_xor(dp, 1); dp += __shfl_xor(dp, 2); dp += __shfl_xor(dp, 4);
            if (TPT == 16) dp += __shfl_xor(dp, 8);
            if (ch == 0) L.D[tok] = dp;
        }
    }
    const int* labw = a.labels ? a.labels + (long long)w.wdw * N : nullptr;
    for (int t = tid; t < NR; t += 256) {
        const int tc = t < N ? t : N - 1;
        L.ia[t] = a.ia[tc];
        L.ib[t] = a.ib[tc];
        L.lab[t] = labw ? labw[tc] : 0;
        if (BWD) L.lse[t] = a.lse_in[stat + tc];
    }
}

template <int NDB>
__global__ __launch_bounds__(256) void winattn_lds_bwd_kernel(WsDev a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LB = NDB <= 2 ? LDB : LDB48, TPT = NDB <= 2 ? 8 : 16;
    const int N = a.N, d = a.d, NCr = (N + 15) / 16, NR = 16 * NCr;
    const LdsWin L = lds_carve<LB>(smem, NR, a.TB, true);
    int grp;
    Who w = who_lds(a, grp);
    stage_table(L.tb, a, w);
    for (int i = threadIdx.x; i < a.TB; i += 256) L.gtb[i] = 0ull;
    const int lane = lane_id(), r = lane & 15, g = lane >> 4, wv = wave_id();
    const long long rs = 3LL * a.heads * d;
    const bool masked = a.labels != nullptr;
    const int qlo = a.qc_lo, qhi = min(a.qc_hi, NCr);
    for (int m = grp; m < a.M; m += a.groups) {
    who_window(a, w, m);
    __syncthreads();                       // the previous window's fragments have been read (and tb / gtb are initialised)
    lds_stage<true, LB, TPT>(a, w, L, NR);
    float* gq = a.gqkv + (long long)w.b * N * rs + w.head * d;
    bf16_t* gqh = reinterpret_cast<bf16_t*>(a.gqkv) + (long long)w.b * N * rs + w.head * d;       // io_bf16: the gradient leaves as bf16
    const bool iobf = a.io_bf16 != 0;
    auto put = [&](long long off, const f32x4& val) {      // four channels of one row
        if (iobf) *reinterpret_cast<s16x4*>(gqh + off) = pack_bf(val);
        else *reinterpret_cast<f32x4*>(gq + off) = val;
    };
    // query rows outside the computed chunks: zero query gradient (they carry no upstream gradient)
    for (int e = threadIdx.x; e < N * (d >> 2); e += 256) {
        const int tok = e / (d >> 2), c4 = e - tok * (d >> 2), qc = tok >> 4;
        if (qc < qlo || qc >= qhi) put((long long)tok * rs + 4 * c4, f32x4{0.f, 0.f, 0.f, 0.f});
    }
    __syncthreads();
    // ---- pass Q: dS^T = P^T (dP^T - D), dQ^T += K^T dS^T, dBias
    for (int qc = qlo + wv; qc < qhi; qc += 4) {
        const int q = 16 * qc + r;
        const int qa = L.ia[q], ql = L.lab[q];
        const float lse = L.lse[q], D = L.D[q];
        s16x4 qf[NDB], gf[NDB];
        f32x4 dq[NDB];
#pragma unroll
        for (int cc = 0; cc < NDB; ++cc) {
            qf[cc] = lds_row<LB>(L.Q, q, 16 * cc + 4 * g);
            gf[cc] = lds_row<LB>(L.G, q, 16 * cc + 4 * g);
            dq[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 1
        for (int kc = 0; kc < NCr; ++kc) {
            f32x4 sT = f32x4{0.f, 0.f, 0.f, 0.f}, dpT = sT;
#pragma unroll
            for (int cc = 0; cc < NDB; ++cc) {
                sT = mfma_bf(lds_row<LB>(L.K, 16 * kc + r, 16 * cc + 4 * g), qf[cc], sT);
                dpT = mfma_bf(lds_row<LB>(L.V, 16 * kc + r, 16 * cc + 4 * g), gf[cc], dpT);
            }
            f32x4 dsT;
#pragma unroll
            for (int j = 0; j < 4; ++j) {                          // rows = keys 16 kc + 4g + j, column = query r
                const int key = 16 * kc + 4 * g + j;
                const int bi = qa + L.ib[key];
                float sc = sT[j] + L.tb[bi];
                if (masked && L.lab[key] != ql) sc -= 100.f;
                const bool ok = key < N && q < N;
                const float v = ok ? __expf(sc - lse) * (dpT[j] - D) : 0.f;
                if (ok) fx_add_s(&L.gtb[bi], v);
                dsT[j] = v;
            }
            const s16x4 dsb = pack_bf(dsT);
#pragma unroll
            for (int db = 0; db < NDB; ++db) dq[db] = mfma_bf(lds_col<LB>(L.K, 16 * kc + 4 * g, 16 * db + r), dsb, dq[db]);
        }
        if (q < N) {
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const int dd = 16 * db + 4 * g;
                if (dd < d) put((long long)q * rs + dd, f32x4{dq[db][0] * a.scale, dq[db][1] * a.scale, dq[db][2] * a.scale, dq[db][3] * a.scale});
            }
        }
    }
    // ---- pass K: dS = P (dP - D), dV^T += dO^T P, dK^T += (scale Q)^T dS      (D and the statistics were staged: no dependency on pass Q)
    for (int kc = wv; kc < NCr; kc += 4) {
        const int key = 16 * kc + r;
        const int kbi = L.ib[key], kl = L.lab[key];
        s16x4 kf[NDB], vf[NDB];
        f32x4 dk[NDB], dv[NDB];
#pragma unroll
        for (int cc = 0; cc < NDB; ++cc) {
            kf[cc] = lds_row<LB>(L.K, key, 16 * cc + 4 * g);
            vf[cc] = lds_row<LB>(L.V, key, 16 * cc + 4 * g);
            dk[cc] = dv[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 1
        for (int qc = qlo; qc < qhi; ++qc) {
            f32x4 sc4 = f32x4{0.f, 0.f, 0.f, 0.f}, dp = sc4;
#pragma unroll
            for (int cc = 0; cc < NDB; ++cc) {
                sc4 = mfma_bf(lds_row<LB>(L.Q, 16 * qc + r, 16 * cc + 4 * g), kf[cc], sc4);      // rows = queries 16 qc + 4g + j, column = key r
                dp = mfma_bf(lds_row<LB>(L.G, 16 * qc + r, 16 * cc + 4 * g), vf[cc], dp);
            }
            f32x4 p, ds;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int qq = 16 * qc + 4 * g + j;
                float sc = sc4[j] + L.tb[L.ia[qq] + kbi];
                if (masked && L.lab[qq] != kl) sc -= 100.f;
                const float pv = (qq < N && key < N) ? __expf(sc - L.lse[qq]) : 0.f;
                p[j] = pv;
                ds[j] = pv * (dp[j] - L.D[qq]);
            }
            const s16x4 pb = pack_bf(p), dsb = pack_bf(ds);
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                dv[db] = mfma_bf(lds_col<LB>(L.G, 16 * qc + 4 * g, 16 * db + r), pb, dv[db]);      // dV^T += dO^T P
                dk[db] = mfma_bf(lds_col<LB>(L.Q, 16 * qc + 4 * g, 16 * db + r), dsb, dk[db]);     // dK^T += (scale Q)^T dS
            }
        }
        if (key < N) {
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const int dd = 16 * db + 4 * g;
                if (dd < d) {
                    put((long long)key * rs + dd + a.heads * d, dk[db]);
                    put((long long)key * rs + dd + 2 * a.heads * d, dv[db]);
                }
            }
        }
    }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.TB; i += 256)
        if (L.gtb[i] != 0ull) atomic_add_f32(&a.gtable[(long long)i * w.tstr + w.tofs], (float)(long long)L.gtb[i] * (1.f / FXS));
}

// ---- (round 4) the same backward with ONE evaluation of every score tile.  The two-pass kernel above evaluates every score tile
// twice (bias lookup, mask, exponential, packing: the bookkeeping that bounds it) because dQ wants dS with the queries on the
// fragment's row lanes and dK / dV want it with the keys there, and it pays 4 LDS 64-bit atomics per lane and tile pair for the
// bias gradient.  Here:
//   pass 1   a wave owns a KEY chunk (NW waves, chunks kc = wave, wave + NW ...) and walks the query chunks once:
//              S, dP         two MFMA pairs, rows = queries 4g + j, column = key r                     (as pass K above)
//              P, dS         one exponential per element
//              dV^T, dK^T    accumulate in registers (dO^T P, (scale Q)^T dS), stored by the owning wave at the end
//              dBias         dS added into a DENSE [key][query] fp32 image of the window pairs in LDS: the tile (kc, qc) belongs
//                            to one wave for every window of the workgroup, so the update is a plain 16-byte read-modify-write
//                            per lane (no atomics; row pitch NR + 4: bank-conflict free); the image is folded into the table
//                            slots (index ia[q] + ib[key], ~3.8 pairs per slot) ONCE per workgroup and flushed with float atomics
//              dS (bf16)     written TRANSPOSED into a [query][key] LDS matrix (four 2-byte stores per lane)
//            every operand of tile pair t + 1 (row / column fragments, index vectors, statistics, bias values, the dense tile) is
//            read while tile pair t is computed: with two waves per SIMD the walk is bound by dependent LDS latencies otherwise;
//   pass 2   a wave owns a QUERY chunk: dQ = dS K from the stored matrix, row fragments as one 8-byte read, two MFMAs per tile
//            pair and no score arithmetic; scaled and stored straight from the registers.
// (LDS float atomics for dQ instead of pass 2 were measured first: 8 ds_add_f32 per lane and tile pair cost 343 of 575 us at the
// Pangu C4 layer-1 shape.)  Per tile pair: 10 MFMAs instead of 14 and one score evaluation instead of two.  NW = 8 (512 threads,
// 5 - 8 key chunks: Pangu's 98 tokens) or 4 (<= 4 chunks: Swin's 49); LDS 127 KB (N = 98, table 2548) / 40 KB (N = 49).
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct LdsWin2 {
    LdsWin w;
    float* dense;
    bf16_t* dsm;
    int* src;          // gqkv row of every window position (token-layout mode: the map's row; -1 = padded position)
    float* fillv;      // [3][32]: this head's slice of the fill vector (token-layout operands)
};
__host__ __device__ inline size_t lds2_bytes(int NR, int TB) {
    return (size_t)4 * NR * LDB * 2 + (size_t)5 * NR * 4 + (size_t)((TB + 3) & ~3) * 4 + (size_t)NR * (NR + 4) * 4 + (size_t)NR * (NR + 4) * 2 +
           (size_t)NR * 4 + 96 * 4;
}
__device__ __forceinline__ LdsWin2 lds2_carve(void* smem, int NR, int TB) {
    LdsWin2 L;
    bf16_t* h = reinterpret_cast<bf16_t*>(smem);
    L.w.Q = h; L.w.K = L.w.Q + NR * LDB; L.w.V = L.w.K + NR * LDB; L.w.G = L.w.V + NR * LDB;
    float* f = reinterpret_cast<float*>(L.w.G + NR * LDB);
    L.w.D = f; L.w.lse = f + NR;
    L.w.ia = reinterpret_cast<int*>(f + 2 * NR); L.w.ib = L.w.ia + NR; L.w.lab = L.w.ib + NR;
    L.w.tb = reinterpret_cast<float*>(L.w.lab + NR);
    L.w.gtb = nullptr;
    L.dense = L.w.tb + ((TB + 3) & ~3);
    L.dsm = reinterpret_cast<bf16_t*>(L.dense + NR * (NR + 4));
    L.src = reinterpret_cast<int*>(L.dsm + NR * (NR + 4));
    L.fillv = reinterpret_cast<float*>(L.src + NR);
    return L;
}
// lds_stage<true> in two halves for a workgroup of NT threads: the global loads of window m + 1 (into registers) are issued before the
// passes over window m and land under them; the conversion to the bf16 LDS image follows the barrier that ends window m.  (Measured
// with the in-kernel stamps at the Pangu C4 layer-1 shape: staging in one piece took 7.9 k cycles per window, pass 1 8.1 k.)
// Two (token, 4-channel group) items per thread cover NR <= 128 rows with 512 threads and NR <= 64 with 256.
template <bool IOBF>
struct Stage2 {
    typedef typename std::conditional<IOBF, f32x2, f32x4>::type go_t;      // IOBF: four bf16 = 8 raw bytes per row piece
    go_t q[2], k[2], v[2], g[2], o[2];
    float lse;
    int lab, src, pad;      // pad (IOBF): bit i = item i is a padded position
};
// token-layout mode: the map entries a thread needs for window m + 1 (gout rows of its two items; the LDS copies of both maps) are
// fetched one window earlier still, so that the dependent gout loads do not wait on them
struct Idx2 {
    int d[2], s[2];
};
template <int NT>
__device__ __forceinline__ void lds2_index(Idx2& I, const WsDev& a, const Who& w) {
    const int N = a.N, tid = threadIdx.x;
    // (window-layout mode reads a.ia instead -- any valid int [N] array -- and ignores the values: no branch around the loads)
    const int* dm = a.dst_map ? a.dst_map + (long long)w.wdw * N : a.ia;
    const int* sm = a.src_map ? a.src_map + (long long)w.wdw * N : a.ia;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int tok = (tid >> 3) + i * (NT / 8);
        I.d[i] = dm[tok < N ? tok : N - 1];
        I.s[i] = sm[tok < N ? tok : N - 1];
    }
}
template <int NT, bool IOBF>
__device__ __forceinline__ void lds2_load(Stage2<IOBF>& R, const Idx2& I, const WsDev& a, const Who& w, int NR) {
    // Addresses = a workgroup-uniform base (window / sample, head) + a 32-bit per-thread offset (row * pitch + channel): the loads
    // take the scalar-base form and the per-thread arithmetic stays in 32 bits (a sample's qkv tensor is < 2^31 elements: checked
    // by the launcher).  No branch around a load: clamped rows + selects.
    const int N = a.N, d = a.d, tid = threadIdx.x, ch = tid & 7;
    const int rs = 3 * a.heads * d, os = a.heads * d, hd = a.heads * d;
    const bool tokm = a.dst_map != nullptr, full = a.fill != nullptr;
    const long long tb0 = (long long)(w.b / a.nW) * a.Ltok, wb0 = (long long)w.b * N;
    const int cc = 4 * ch < d ? 4 * ch : 0;              // clamped: unconditional loads
    R.pad = 0;
    if constexpr (IOBF) {
        const __bf16* qb = reinterpret_cast<const __bf16*>(a.qkv) + tb0 * rs + w.head * d;
        const __bf16* gb = reinterpret_cast<const __bf16*>(a.gout) + tb0 * os + w.head * d;
        const __bf16* ob = reinterpret_cast<const __bf16*>(a.o) + tb0 * os + w.head * d;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // bf16 rows (8 raw bytes per piece); a padded position is flagged and takes the fill vector from its LDS copy in
            // lds2_store; a dropped position gets zero gradient
            const int qo = (I.s[i] < 0 ? 0 : I.s[i]) * rs + cc, go = (I.d[i] < 0 ? 0 : I.d[i]) * os + cc;
            R.q[i] = *reinterpret_cast<const f32x2*>(qb + qo);
            R.k[i] = *reinterpret_cast<const f32x2*>(qb + qo + hd);
            R.v[i] = *reinterpret_cast<const f32x2*>(qb + qo + 2 * hd);
            R.g[i] = *reinterpret_cast<const f32x2*>(gb + go);
            R.o[i] = *reinterpret_cast<const f32x2*>(ob + go);
            // flags only: a select on a value that has just been requested would make the wave wait for it HERE (round 6: the stamps
            // showed 5.8 k cycles between the end of the staging and the first tile pair of every window); lds2_store applies them
            R.pad |= ((I.s[i] < 0 ? 1 : 0) << i) | ((I.d[i] < 0 ? 4 : 0) << i);
        }
    } else {
        const float* qb = a.qkv + (full ? tb0 : wb0) * rs + w.head * d;
        const float* gb = a.gout + (tokm ? tb0 : wb0) * os + w.head * d;
        const float* ob = a.o + (full ? tb0 : wb0) * os + w.head * d;
        const float* fl = a.fill ? a.fill + w.head * d : a.qkv;      // (never read through when !full)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int tok = (tid >> 3) + i * (NT / 8);
            const int tc = tok < N ? tok : N - 1;
            const bool padq = full && I.s[i] < 0;
            const float* row = padq ? fl + cc : qb + ((full ? I.s[i] : tc) * rs + cc);
            R.q[i] = *reinterpret_cast<const f32x4*>(row);
            R.k[i] = *reinterpret_cast<const f32x4*>(row + hd);
            R.v[i] = *reinterpret_cast<const f32x4*>(row + 2 * hd);
            const int gt = tokm ? (I.d[i] < 0 ? 0 : I.d[i]) : tc;
            R.g[i] = *reinterpret_cast<const f32x4*>(gb + (gt * os + cc));
            R.o[i] = *reinterpret_cast<const f32x4*>(ob + ((full ? gt : tc) * os + cc));
            R.pad |= ((tokm && I.d[i] < 0) ? 4 : 0) << i;      // bit 2 + i: a dropped position (zero gradient; applied in lds2_store)
        }
    }
    const int tc = tid < N ? tid : N - 1;
    R.lse = (a.lse_in + ((long long)w.b * a.heads + w.head) * N)[tc];
    R.lab = (a.labels ? a.labels + (long long)w.wdw * N : a.ia)[tc];          // raw loads: lds2_store picks (labels ? lab : 0), (tokm ? src : token)
    R.src = (tokm ? a.src_map + (long long)w.wdw * N : a.ia)[tc];
}
template <int NT, bool IOBF>
__device__ __forceinline__ void lds2_store(const Stage2<IOBF>& R, const WsDev& a, const LdsWin& L, int* srcv, const float* fillv, int NR) {
    const int N = a.N, d = a.d, tid = threadIdx.x, ch = tid & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int tok = (tid >> 3) + i * (NT / 8);
        if (tok < NR) {
            const bool ok = tok < N && 4 * ch < d;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            f32x4 g, o;
            if constexpr (IOBF) {
                // raw bf16 rows go to the LDS image as they are (q UNSCALED: the passes scale the scores and dK instead -- the same
                // operands as the raw-fragment forward); only D = rowsum(dO o) needs floats
                typedef __bf16 bh4 __attribute__((ext_vector_type(4)));
                auto widen = [](const f32x2 raw) {
                    const bh4 h = __builtin_bit_cast(bh4, raw);
                    return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
                };
                s16x4 qh = __builtin_bit_cast(s16x4, R.q[i]), kh = __builtin_bit_cast(s16x4, R.k[i]), vh = __builtin_bit_cast(s16x4, R.v[i]);
                s16x4 gh = __builtin_bit_cast(s16x4, R.g[i]);
                if ((R.pad >> i) & 1) {          // the fill (qkv bias) of this head: [3][32] floats in LDS
                    qh = pack_bf(*reinterpret_cast<const f32x4*>(fillv + 4 * ch));
                    kh = pack_bf(*reinterpret_cast<const f32x4*>(fillv + 32 + 4 * ch));
                    vh = pack_bf(*reinterpret_cast<const f32x4*>(fillv + 64 + 4 * ch));
                }
                g = widen(R.g[i]); o = widen(R.o[i]);
                if ((R.pad >> (2 + i)) & 1) { gh = s16x4{0, 0, 0, 0}; g = z; o = z; }      // a dropped position
                if (!ok) { qh = kh = vh = gh = s16x4{0, 0, 0, 0}; g = z; o = z; }
                *reinterpret_cast<s16x4*>(L.Q + tok * LDB + 4 * ch) = qh;
                *reinterpret_cast<s16x4*>(L.K + tok * LDB + 4 * ch) = kh;
                *reinterpret_cast<s16x4*>(L.V + tok * LDB + 4 * ch) = vh;
                *reinterpret_cast<s16x4*>(L.G + tok * LDB + 4 * ch) = gh;
            } else {
                f32x4 q = R.q[i], k = R.k[i], v = R.v[i];
                g = R.g[i]; o = R.o[i];
                if ((R.pad >> (2 + i)) & 1) { g = z; if (a.fill) o = z; }                 // a dropped position
                if (!ok) { q = z; k = z; v = z; g = z; o = z; }
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) q[s2] *= a.scale;
                *reinterpret_cast<s16x4*>(L.Q + tok * LDB + 4 * ch) = pack_bf(q);
                *reinterpret_cast<s16x4*>(L.K + tok * LDB + 4 * ch) = pack_bf(k);
                *reinterpret_cast<s16x4*>(L.V + tok * LDB + 4 * ch) = pack_bf(v);
                *reinterpret_cast<s16x4*>(L.G + tok * LDB + 4 * ch) = pack_bf(g);
            }
            float dp = g[0] * o[0] + g[1] * o[1] + g[2] * o[2] + g[3] * o[3];
            dp += __shfl_xor(dp, 1); dp += __shfl_xor(dp, 2); dp += __shfl_xor(dp, 4);
            if (ch == 0) L.D[tok] = dp;
        }
    }
    if (tid < NR) { L.lse[tid] = R.lse; L.lab[tid] = a.labels ? R.lab : 0; srcv[tid] = a.dst_map ? R.src : (tid < N ? tid : N - 1); }
}

// The bias-table slice of a (type, head) into LDS with the loads of eight strides of the workgroup in flight at once (round 6: the plain
// `for (i = tid; i < TB; i += NT) tb[i] = table[..]` loop is a chain of dependent load -> LDS-store round trips, five of them at
// Pangu's 2548 entries and 512 threads -- part of the ~ 8.5 k cycles a workgroup spends before its first window)
template <int NT>
__device__ __forceinline__ void stage_table_batched(float* tb, const WsDev& a, const Who& w) {
    const int TB = a.TB, tid = threadIdx.x;
    for (int i0 = 0; i0 < TB; i0 += 8 * NT) {
        float tv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(i0 + u * NT + tid, TB - 1);           // clamped: unconditional loads
            tv[u] = a.table_t ? a.table_t[w.tofs * TB + i] : a.table[(long long)i * w.tstr + w.tofs];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * NT + tid;
            if (i < TB) tb[i] = tv[u];
        }
    }
}

template <int NDB, int NW, bool IOBF = false>
__global__ __launch_bounds__(64 * NW) void winattn_lds_bwd1p_kernel(WsDev a) {
    constexpr int NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = a.N, d = a.d, NCr = (N + 15) / 16, NR = 16 * NCr, LDD = NR + 4;
    const LdsWin2 L2 = lds2_carve(smem, NR, a.TB);
    const LdsWin& L = L2.w;
    int grp;
    Who w = who_lds(a, grp);
    stage_table_batched<NT>(L.tb, a, w);
    DLWP_STAMP(0);
    Stage2<IOBF> R;
    Idx2 I;
    who_window(a, w, grp);
    lds2_index<NT>(I, a, w);
    lds2_load<NT, IOBF>(R, I, a, w, NR);           // the first window's loads fly while tb / dense are initialised
    if (grp + a.groups < a.M) {
        Who wn = w;
        who_window(a, wn, grp + a.groups);
        lds2_index<NT>(I, a, wn);
    }
    const bool tokm = a.dst_map != nullptr;
    const float sscale = IOBF ? a.scale : 1.f;      // IOBF: q is staged unscaled (raw bf16 rows), the scores take the scale
    f32x4 padk[NDB], padv[NDB], padq[NDB];   // token-layout mode: this lane's share of the padded positions' gradient (all windows)
#pragma unroll
    for (int db = 0; db < NDB; ++db) padk[db] = padv[db] = padq[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < NR * LDD; i += NT) L2.dense[i] = 0.f;
    for (int t = threadIdx.x; t < NR; t += NT) {      // the bias-index vectors do not depend on the window
        const int tc = t < N ? t : N - 1;
        L.ia[t] = a.ia[tc];
        L.ib[t] = a.ib[tc];
    }
    if (threadIdx.x < 96) {
        const int part = threadIdx.x >> 5, c = threadIdx.x & 31;
        L2.fillv[threadIdx.x] = (a.fill && c < d) ? a.fill[part * a.heads * d + w.head * d + c] : 0.f;
    }
    const int lane = lane_id(), r = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    const long long rs = 3LL * a.heads * d;
    const bool masked = a.labels != nullptr;
    const int qlo = a.qc_lo, qhi = min(a.qc_hi, NCr), nq = qhi - qlo;
    // operands of one tile pair of pass 1 (everything that does not depend on the matrix products)
    // The fragments and the gathered bias values travel one tile pair ahead.  The four 16-byte vectors of a pair -- labels,
    // statistics, D, the dense tile -- travel with them in the 4-wave kernel (FULLPRE); in the 8-wave kernel they are read at the top
    // of the pair's own step, ahead of its first MFMAs: 16 registers less, without which that kernel sits at 256 VGPRs + scratch
    // (measured: 149 -> 199 us at the Pangu C4 layer-1 shape with 60 B of scratch per lane).
    constexpr bool FULLPRE = NW == 4;
    struct Pre {
        s16x4 qr[NDB], gr[NDB], qcol[NDB], gcol[NDB];
        f32x4 tbv;
        i32x4 lab;
        f32x4 lse, D, dense;
    };
    for (int m = grp; m < a.M; m += a.groups) {
        who_window(a, w, m);
        __syncthreads();                   // the previous window's fragments have been read (first turn: tb / dense are initialised)
        if (m == grp) DLWP_STAMP(1);
        lds2_store<NT, IOBF>(R, a, L, L2.src, L2.fillv, NR);
        // gqkv rows: window layout [b][n], or token layout [batch][src token] (row index from the staged map)
        const long long gq_off = (tokm ? (long long)(w.b / a.nW) * a.Ltok : (long long)w.b * N) * rs + w.head * d;      // in elements
        float* gq = a.gqkv + gq_off;
        __syncthreads();
        if (m == grp) DLWP_STAMP(2);
        if (m + a.groups < a.M) {
            Who wn = w;
            who_window(a, wn, m + a.groups);
            lds2_load<NT, IOBF>(R, I, a, wn, NR);
        }
        {
            // UNCONDITIONAL (a clamped window where there is none): under `if (m + 2 groups < M)` the compiler kept the new entries in
            // temporaries and copied them into I's registers right behind the loads -- a wait for every load in flight, the window's
            // rows included: 5.8 k cycles per window between the staging barrier and the first tile pair (stamps, round 6)
            Who wn = w;
            who_window(a, wn, min(m + 2 * a.groups, a.M - 1));
            lds2_index<NT>(I, a, wn);
        }
        // ---- pass 1
        for (int kc = wv; kc < ((a.dbg & 1) ? 0 : NCr); kc += NW) {
            const int key = 16 * kc + r;
            const int kbi = L.ib[key], kl = L.lab[key];
            s16x4 kf[NDB], vf[NDB];
            f32x4 dk[NDB], dv[NDB];
#pragma unroll
            for (int cc = 0; cc < NDB; ++cc) {
                kf[cc] = lds_row(L.K, key, 16 * cc + 4 * g);
                vf[cc] = lds_row(L.V, key, 16 * cc + 4 * g);
                dk[cc] = dv[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            float* drow = L2.dense + (16 * kc + r) * LDD + 4 * g;
            auto load = [&](Pre& P, int qc) {
                const int q0 = 16 * qc + 4 * g;
                const i32x4 qia = *reinterpret_cast<const i32x4*>(L.ia + q0);
#pragma unroll
                for (int cc = 0; cc < NDB; ++cc) {
                    P.qr[cc] = lds_row(L.Q, 16 * qc + r, 16 * cc + 4 * g);
                    P.gr[cc] = lds_row(L.G, 16 * qc + r, 16 * cc + 4 * g);
                    P.qcol[cc] = lds_col(L.Q, q0, 16 * cc + r);
                    P.gcol[cc] = lds_col(L.G, q0, 16 * cc + r);
                }
                if constexpr (FULLPRE) {
                    P.lab = *reinterpret_cast<const i32x4*>(L.lab + q0);
                    P.lse = *reinterpret_cast<const f32x4*>(L.lse + q0);
                    P.D = *reinterpret_cast<const f32x4*>(L.D + q0);
                    P.dense = *reinterpret_cast<const f32x4*>(drow + 16 * qc);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) P.tbv[j] = L.tb[qia[j] + kbi];
            };
            Pre cur, nxt;
            if (nq > 0) load(cur, qlo);
            nxt = cur;
#pragma unroll 1
            for (int qc = qlo; qc < qhi; ++qc) {
                if (m == grp && kc == 0) DLWP_STAMP(24 + qc - qlo);
                const int q0 = 16 * qc + 4 * g;
                i32x4 qlab;
                f32x4 qlse, qD, dtile;
                if constexpr (FULLPRE) {
                    qlab = cur.lab; qlse = cur.lse; qD = cur.D; dtile = cur.dense;
                } else {
                    qlab = *reinterpret_cast<const i32x4*>(L.lab + q0);
                    qlse = *reinterpret_cast<const f32x4*>(L.lse + q0);
                    qD = *reinterpret_cast<const f32x4*>(L.D + q0);
                    dtile = *reinterpret_cast<const f32x4*>(drow + 16 * qc);
                }
                if (qc + 1 < qhi) load(nxt, qc + 1);
                f32x4 sc4 = f32x4{0.f, 0.f, 0.f, 0.f}, dp = sc4;
#pragma unroll
                for (int cc = 0; cc < NDB; ++cc) {
                    sc4 = mfma_bf(cur.qr[cc], kf[cc], sc4);      // rows = queries 16 qc + 4g + j, column = key r
                    dp = mfma_bf(cur.gr[cc], vf[cc], dp);
                }
                f32x4 p, ds;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float sc = sc4[j] * sscale + cur.tbv[j];
                    if (masked && qlab[j] != kl) sc -= 100.f;
                    const float pv = (q0 + j < N && key < N) ? __expf(sc - qlse[j]) : 0.f;
                    p[j] = pv;
                    ds[j] = pv * (dp[j] - qD[j]);
                }
                if (!(a.dbg & 8)) *reinterpret_cast<f32x4*>(drow + 16 * qc) = dtile + ds;      // bias gradient: this wave's own tile
                const s16x4 pb = pack_bf(p), dsb = pack_bf(ds);
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    dv[db] = mfma_bf(cur.gcol[db], pb, dv[db]);      // dV^T += dO^T P
                    dk[db] = mfma_bf(cur.qcol[db], dsb, dk[db]);     // dK^T += (scale Q)^T dS
                }
                if (!(a.dbg & 4)) {
                    // dS with the KEYS on the rows: one 8-byte store of this lane's four queries (round 6; the [query][key] image took
                    // four 2-byte stores); pass 2 gets its [key 4g .. 4g + 3][query r] operand through the transposing read
                    *reinterpret_cast<s16x4*>(L2.dsm + key * LDD + q0) = dsb;
                }
                cur = nxt;
            }
            if (key < N) {
                const int krow = L2.src[key];
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const int dd = 16 * db + 4 * g;
                    if constexpr (IOBF) dk[db] *= a.scale;          // (the LDS image holds q unscaled)
                    if (krow < 0) {
                        padk[db] += dk[db];
                        padv[db] += dv[db];
                    } else if (dd < d) {
                        if constexpr (IOBF) {
                            bf16_t* dst = reinterpret_cast<bf16_t*>(a.gqkv) + gq_off + (long long)krow * rs + dd;
                            *reinterpret_cast<s16x4*>(dst + a.heads * d) = pack_bf(dk[db]);
                            *reinterpret_cast<s16x4*>(dst + 2 * a.heads * d) = pack_bf(dv[db]);
                        } else {
                            float* dst = gq + (long long)krow * rs + dd;
                            *reinterpret_cast<f32x4*>(dst + a.heads * d) = dk[db];
                            *reinterpret_cast<f32x4*>(dst + 2 * a.heads * d) = dv[db];
                        }
                    }
                }
            }
        }
        if (m == grp) DLWP_STAMP_WAVE(8);
        __syncthreads();
        if (m == grp) DLWP_STAMP(3);
        // ---- pass 2: dQ^T = K^T dS^T for the computed query chunks (rows = channels 16 db + 4g + j, column = query r: a lane
        //      holds four consecutive channels of one query row -> one vector store), zeros elsewhere
        for (int qc = wv; qc < ((a.dbg & 32) ? 0 : NCr); qc += NW) {
            f32x4 dq[NDB];
#pragma unroll
            for (int db = 0; db < NDB; ++db) dq[db] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (qc >= qlo && qc < qhi) {
                s16x4 sf = lds_col_pitch(L2.dsm, LDD, 4 * g, 16 * qc + r), sn = sf;
                s16x4 kcol[NDB], kn[NDB];
#pragma unroll
                for (int db = 0; db < NDB; ++db) kn[db] = kcol[db] = lds_col(L.K, 4 * g, 16 * db + r);
#pragma unroll 1
                for (int kc = 0; kc < NCr; ++kc) {
                    if (kc + 1 < NCr) {
                        sn = lds_col_pitch(L2.dsm, LDD, 16 * (kc + 1) + 4 * g, 16 * qc + r);
#pragma unroll
                        for (int db = 0; db < NDB; ++db) kn[db] = lds_col(L.K, 16 * (kc + 1) + 4 * g, 16 * db + r);
                    }
#pragma unroll
                    for (int db = 0; db < NDB; ++db) dq[db] = mfma_bf(kcol[db], sf, dq[db]);
                    sf = sn;
#pragma unroll
                    for (int db = 0; db < NDB; ++db) kcol[db] = kn[db];
                }
            }
            const int q = 16 * qc + r;
            if (q < N) {
                const int qrow = L2.src[q];
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const int dd = 16 * db + 4 * g;
                    const f32x4 v = dq[db] * a.scale;
                    if (qrow < 0) padq[db] += v;
                    else if (dd < d) {
                        if constexpr (IOBF) *reinterpret_cast<s16x4*>(reinterpret_cast<bf16_t*>(a.gqkv) + gq_off + (long long)qrow * rs + dd) = pack_bf(v);
                        else *reinterpret_cast<f32x4*>(gq + (long long)qrow * rs + dd) = v;
                    }
                }
            }
        }
    }
    if (tokm) {
        // the padded positions' gradient = the fill's: every fragment holds channels 16 db + 4g + j of token r -> sum over r
        float* gf = a.gfill + w.head * d;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float sq = padq[db][j], sk = padk[db][j], sv = padv[db][j];
#pragma unroll
                for (int sh = 1; sh < 16; sh <<= 1) { sq += __shfl_xor(sq, sh); sk += __shfl_xor(sk, sh); sv += __shfl_xor(sv, sh); }
                const int dd = 16 * db + 4 * g + j;
                if (r == 0 && dd < d) {
                    if (sq != 0.f) atomic_add_f32(gf + dd, sq);
                    if (sk != 0.f) atomic_add_f32(gf + a.heads * d + dd, sk);
                    if (sv != 0.f) atomic_add_f32(gf + 2 * a.heads * d + dd, sv);
                }
            }
        }
    }
    DLWP_STAMP_WAVE(16);
    __syncthreads();
    DLWP_STAMP(4);
    // fold the dense image into the table slots (the staging area is free now), then flush
    float* tbg = reinterpret_cast<float*>(L.Q);
    for (int i = threadIdx.x; i < a.TB; i += NT) tbg[i] = 0.f;
    __syncthreads();
    // (four entries per thread in flight instead of one were measured in round 6: 18.6 k -> 17.4 k cycles -- the LDS float adds, not the
    // latency of the index reads, bound the fold)
    for (int e = threadIdx.x; e < ((a.dbg & 16) ? 0 : N * N); e += NT) {
        const int key = e / N, q = e - key * N;
        const float v = L2.dense[key * LDD + q];
        if (v != 0.f) atomicAdd(tbg + L.ia[q] + L.ib[key], v);
    }
    __syncthreads();
    DLWP_STAMP(5);
    for (int i = threadIdx.x; i < a.TB; i += NT)
        if (tbg[i] != 0.f) atomic_add_f32(&a.gtable[(long long)i * w.tstr + w.tofs], tbg[i]);
    DLWP_STAMP(6);
}

// windows of one (type, head) per workgroup: enough workgroups to fill the chip about three times over
// (Pangu C4 layer 1, 4218 (window, head) pairs, tools/probe_winattn_c4.py: 768 workgroups 277 us, 1536: 272, 3072: 288, one per
// window: 370; the register-fragment kernel: 338)
// Workgroups of the two-pass LDS-staged backward: ONE resident round (occ workgroups per CU: the kernel's waves per SIMD), every workgroup
// with the same number of windows.  Round 6, Swin C4 in the step: stage 0 (5624 (window, head) pairs) 112.6 us at the 1536 workgroups of
// rounds 3 - 5 (1.5 rounds, 3 or 4 windows each), 101 - 102 us at 960 - 1024; stage 1 (1520 pairs, head dim 48, three waves per SIMD) 64 us
// at 1520 (one window each: the table staging and flush of a workgroup paid per window, a partial second round), 59 at 768.
static int lds_groups(int M, int heads, int ntypes, int occ) {
    const int env = dlwp_tune_or("WINATTN_WG_BWD", 0);
    if (env) {
        const long long g = ((long long)env + heads * ntypes - 1) / (heads * ntypes);
        return g < 1 ? 1 : (g > M ? M : (int)g);
    }
    static const int ncu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    const long long pairs = (long long)M * heads * ntypes, slots = (long long)ncu * occ;
    const int per_wg = (int)((pairs + slots - 1) / slots);          // windows per workgroup
    return (M + per_wg - 1) / per_wg;
}

// the LDS-staged family takes the bf16 matrix mode with 16-byte loadable head slices
static bool lds_family_applies(int N, int d) {
    // head dims 33 .. 48: the two-pass backward only (windows of at most 64 tokens), see LDB48
    return dlwp_get_gemm_precision() == 1 && d % 4 == 0 && (d <= 32 || (d <= 48 && N <= 64)) && N <= 128 && !dlwp_tune_on("WINATTN_NOLDS");
}

int ws_setup(WsDev& a, int B_, int nW, int N, int TB, int ntypes, int heads, int d, float scale, int q_lo, int q_hi) {
    a.B_ = B_; a.nW = nW; a.N = N; a.TB = TB; a.ntypes = ntypes; a.heads = heads; a.d = d; a.scale = scale;
    const int nc = (N + 15) / 16;
    a.qc_lo = q_lo < 0 ? 0 : q_lo / 16;
    a.qc_hi = (q_hi < 0 || q_hi > N) ? nc : (q_hi + 15) / 16;
    if (a.qc_lo >= a.qc_hi) { a.qc_lo = 0; a.qc_hi = nc; }
    a.M = B_ / ntypes;
    a.groups = (a.M + 3) / 4;
    return DLWP_OK;
}

// ---- (round 4) the forward pass on the same construction: a workgroup (four waves) walks the windows of ONE (window type, head);
// the NEXT window's token rows (q, k, v through the source map) are loaded into registers while the current window is computed from
// its bf16 LDS image; a wave owns the query chunks qc = wave, wave + 4 ... of the caller's range and keeps S^T = K Q^T of all key
// chunks in registers (rows = keys 4g + j, column = query r: the softmax statistics of a query live in the four lanes sharing r),
// O^T = V^T P^T comes out as four consecutive channels of one query per lane -> one vector store to the token the reverse map names.
// Token-layout operands only (dlwp_window_attn_fwd_tokens); NC = key chunks (4 or 8).
template <bool IOBF>
struct FwdStage {
    typedef typename std::conditional<IOBF, f32x2, f32x4>::type row_t;
    row_t q[4], k[4], v[4];
    int lab, dst, pad;
};
template <int NC, int NDB, bool IOBF>
__global__ __launch_bounds__(256) void winattn_lds_fwd_tok_kernel(WsDev a) {
    constexpr int NT = 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = a.N, d = a.d, NCr = (N + 15) / 16, NR = 16 * NCr;
    bf16_t* Qs = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Ks = Qs + NR * LDB;
    bf16_t* Vs = Ks + NR * LDB;
    int* ias = reinterpret_cast<int*>(Vs + NR * LDB);
    int* ibs = ias + NR;
    int* labs = ibs + NR;
    int* dsts = labs + NR;
    float* tb = reinterpret_cast<float*>(dsts + NR);
    float* fillv = tb + ((a.TB + 3) & ~3);
    int grp;
    Who w = who_lds(a, grp);
    stage_table_batched<NT>(tb, a, w);
    for (int t = threadIdx.x; t < NR; t += NT) {
        const int tc = t < N ? t : N - 1;
        ias[t] = a.ia[tc];
        ibs[t] = a.ib[tc];
    }
    if (threadIdx.x < 96) {
        const int part = threadIdx.x >> 5, c = threadIdx.x & 31;
        fillv[threadIdx.x] = c < d ? a.fill[part * a.heads * d + w.head * d + c] : 0.f;
    }
    const int tid = threadIdx.x, lane = lane_id(), r = lane & 15, g = lane >> 4, wv = tid >> 6, ch = tid & 7;
    const int rs = 3 * a.heads * d, os = a.heads * d, hd = a.heads * d;
    const int cc = 4 * ch < d ? 4 * ch : 0;
    const int qlo = a.qc_lo, qhi = min(a.qc_hi, NCr);
    const float sscale = IOBF ? a.scale : 1.f;
    const bool masked = a.labels != nullptr;
    int sidx[4];
    auto load_index = [&](int wdw) {
        const int* sm = a.src_map + (long long)wdw * N;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tok = (tid >> 3) + 32 * i;
            sidx[i] = sm[tok < N ? tok : N - 1];
        }
    };
    FwdStage<IOBF> R;
    auto load_rows = [&](const Who& ww) {
        const long long tb0 = (long long)(ww.b / a.nW) * a.Ltok;
        R.pad = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int so = (sidx[i] < 0 ? 0 : sidx[i]) * rs + cc;
            R.pad |= (sidx[i] < 0 ? 1 : 0) << i;
            if constexpr (IOBF) {
                const __bf16* qb = reinterpret_cast<const __bf16*>(a.qkv) + tb0 * rs + ww.head * d;
                R.q[i] = *reinterpret_cast<const f32x2*>(qb + so);
                R.k[i] = *reinterpret_cast<const f32x2*>(qb + so + hd);
                R.v[i] = *reinterpret_cast<const f32x2*>(qb + so + 2 * hd);
            } else {
                const float* qb = a.qkv + tb0 * rs + ww.head * d;
                R.q[i] = *reinterpret_cast<const f32x4*>(qb + so);
                R.k[i] = *reinterpret_cast<const f32x4*>(qb + so + hd);
                R.v[i] = *reinterpret_cast<const f32x4*>(qb + so + 2 * hd);
            }
        }
        const int tc = tid < N ? tid : N - 1;
        R.lab = (a.labels ? a.labels + (long long)ww.wdw * N : a.ia)[tc];      // raw: store_rows picks (labels ? lab : 0) -- no wait here
        R.dst = a.dst_map[(long long)ww.wdw * N + tc];
    };
    auto store_rows = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tok = (tid >> 3) + 32 * i;
            if (tok < NR) {
                const bool ok = tok < N && 4 * ch < d;
                s16x4 qh, kh, vh;
                if constexpr (IOBF) {
                    qh = __builtin_bit_cast(s16x4, R.q[i]); kh = __builtin_bit_cast(s16x4, R.k[i]); vh = __builtin_bit_cast(s16x4, R.v[i]);
                    if ((R.pad >> i) & 1) qh = pack_bf(*reinterpret_cast<const f32x4*>(fillv + 4 * ch));
                } else {
                    f32x4 q = ((R.pad >> i) & 1) ? *reinterpret_cast<const f32x4*>(fillv + 4 * ch) : R.q[i];
                    qh = pack_bf(q * a.scale); kh = pack_bf(R.k[i]); vh = pack_bf(R.v[i]);
                }
                if ((R.pad >> i) & 1) {
                    kh = pack_bf(*reinterpret_cast<const f32x4*>(fillv + 32 + 4 * ch));
                    vh = pack_bf(*reinterpret_cast<const f32x4*>(fillv + 64 + 4 * ch));
                }
                if (!ok) qh = kh = vh = s16x4{0, 0, 0, 0};
                *reinterpret_cast<s16x4*>(Qs + tok * LDB + 4 * ch) = qh;
                *reinterpret_cast<s16x4*>(Ks + tok * LDB + 4 * ch) = kh;
                *reinterpret_cast<s16x4*>(Vs + tok * LDB + 4 * ch) = vh;
            }
        }
        if (tid < NR) { labs[tid] = a.labels ? R.lab : 0; dsts[tid] = R.dst; }
    };
    who_window(a, w, grp);
    load_index(w.wdw);
    load_rows(w);
    if (grp + a.groups < a.M) { Who wn = w; who_window(a, wn, grp + a.groups); load_index(wn.wdw); }
    for (int m = grp; m < a.M; m += a.groups) {
        who_window(a, w, m);
        __syncthreads();                   // the previous window's image has been read (first turn: tb / index vectors are staged)
        store_rows();
        __syncthreads();
        if (m + a.groups < a.M) {
            Who wn = w;
            who_window(a, wn, m + a.groups);
            load_rows(wn);
        }
        {   // unconditional, on a clamped window (see winattn_lds_bwd1p_kernel: a conditional prefetch ends in a wait for every load in flight)
            Who wn = w;
            who_window(a, wn, min(m + 2 * a.groups, a.M - 1));
            load_index(wn.wdw);
        }
        const long long tb0 = (long long)(w.b / a.nW) * a.Ltok;
        if (tid < N && ((tid >> 4) < qlo || (tid >> 4) >= qhi))      // rows outside the computed chunks: the backward stages every row's lse
            a.lse[((long long)w.b * a.heads + w.head) * N + tid] = 0.f;
        for (int qc = qlo + wv; qc < qhi; qc += 4) {
            const int q = 16 * qc + r;
            const int qa = ias[q], qlab = labs[q];
            s16x4 qf[NDB];
#pragma unroll
            for (int c2 = 0; c2 < NDB; ++c2) qf[c2] = lds_row(Qs, q, 16 * c2 + 4 * g);
            f32x4 s[NC];
            float mx = -1e30f;
#pragma unroll
            for (int kc = 0; kc < NC; ++kc) {
                s[kc] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (kc < NCr) {
#pragma unroll
                    for (int c2 = 0; c2 < NDB; ++c2) s[kc] = mfma_bf(lds_row(Ks, 16 * kc + r, 16 * c2 + 4 * g), qf[c2], s[kc]);
                    const i32x4 kb4 = *reinterpret_cast<const i32x4*>(ibs + 16 * kc + 4 * g), kl4 = *reinterpret_cast<const i32x4*>(labs + 16 * kc + 4 * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v = s[kc][j] * sscale + tb[qa + kb4[j]];
                        if (masked && kl4[j] != qlab) v -= 100.f;
                        v = 16 * kc + 4 * g + j < N ? v : -1e30f;
                        s[kc][j] = v;
                        mx = fmaxf(mx, v);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[kc][j] = -1e30f;
                }
            }
            mx = col_max4(mx);
            float l = 0.f;
#pragma unroll
            for (int kc = 0; kc < NC; ++kc)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float p = __expf(s[kc][j] - mx);
                    s[kc][j] = p;
                    l += p;
                }
            l = col_sum4(l);
            const float inv = 1.f / l;
            const int orow = q < N ? dsts[q] : -1;
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kc = 0; kc < NC; ++kc)
                    if (kc < NCr) o = mfma_bf(lds_col(Vs, 16 * kc + 4 * g, 16 * db + r), pack_bf(s[kc]), o);
                const int dd = 16 * db + 4 * g;
                if (orow >= 0 && dd < d) {
                    const long long off = (tb0 + orow) * os + w.head * d + dd;
                    if constexpr (IOBF) *reinterpret_cast<s16x4*>(reinterpret_cast<bf16_t*>(a.out) + off) = pack_bf(o * inv);
                    else *reinterpret_cast<f32x4*>(a.out + off) = o * inv;
                }
            }
            if (g == 0 && q < N) a.lse[((long long)w.b * a.heads + w.head) * N + q] = mx + __logf(l);
        }
    }
}

}  // namespace

// Shapes this family takes: at most 128 tokens per window, head_dim <= 32, and enough (window, head) pairs that one wave per
// pair fills the chip -- a single wave walks its pairs of 16 x 16 tiles serially (L1 / L2 latency per step, one prefetch deep),
// so with few windows the tiled kernels of winattn.hip (a workgroup per 64 queries) finish sooner.  Measured
// (profiles/r02_winattn_probe.txt, forward / backward us, tiled -> this family):  1406 windows x 4 heads, N = 49, d = 24:
// 86 / 264 -> 54 / 249;  703 x 6, N = 98, d = 32: 193 / 727 -> 149 / 643;  100 x 4, N = 49, d = 10: 8 / 26 -> 11 / 51.
// live accounting (prof.hip): products = 2 (forward: Q K^T, P V) or 5 (backward: the scores again, dP, dV, dQ, dK) over the
// computed query chunks; bytes = the rows that exist in HBM read / written once (token layout: the unpartitioned tensors)
static inline double ws_prof_flops(const WsDev& a, int products) {
    const double q = 16.0 * (a.qc_hi - a.qc_lo) < a.N ? 16.0 * (a.qc_hi - a.qc_lo) : a.N;
    return products * 2.0 * a.B_ * a.heads * q * a.N * a.d;
}
static inline double ws_prof_bytes(const WsDev& a, bool backward) {
    const double rows = a.fill ? (double)(a.B_ / a.nW) * a.Ltok : (double)a.B_ * a.N;
    const double C = (double)a.heads * a.d, e = a.io_bf16 ? 2 : 4;      // io_bf16: qkv, out, gout, gqkv are all bf16 arrays
    // forward: qkv read, out + lse written; backward: qkv, out, gout, lse read, gqkv written
    return rows * C * e * (backward ? 3 + 1 + 1 + 3 : 3 + 1) + 4.0 * a.B_ * a.heads * a.N;
}
// Round 6: head dims up to 48 for windows of at most 64 tokens in the bf16 matrix mode (Swin C4 stage 1: 49 tokens, head dim 48, 1520 pairs at
// batch 2: tiled kernels 36 + 106 us forward + backward per block, this family 12 + 31 us) -- from WINATTN_SMALL_MIN_PAIRS pairs (default 1024).
bool dlwp_winattn_small_applies(int N, int d, long long pairs) {
    if (N <= 128 && d <= 32 && pairs >= 2048) return true;
    return dlwp_tune_or("WINATTN_D48", 1) != 0 && N <= 64 && d <= 48 && d % 4 == 0 && dlwp_get_gemm_precision() == 1 && !dlwp_tune_on("WINATTN_NOLDS") &&
           pairs >= dlwp_tune_or("WINATTN_SMALL_MIN_PAIRS", 1024);
}

#define WS_DISPATCH(KERNEL, lds)                                                                                          \
    do {                                                                                                                  \
        const int nc = (N + 15) / 16, ndb = (d + 15) / 16;                                                                \
        const bool vec = d % 4 == 0;                                                                                      \
        const dim3 grid((unsigned)(heads * ntypes * a.groups)), block(256);                                               \
        auto go = [&](auto knl) -> int {                                                                                  \
            int rc2 = dlwp_ensure_lds(reinterpret_cast<const void*>(knl), lds, #KERNEL);                                  \
            if (rc2) return rc2;                                                                                          \
            dlwp_prof_scope prof((hipStream_t)stream, ws_prof_flops(a, 2), ws_prof_bytes(a, false), #KERNEL "<%d, %d>", nc <= 4 ? 4 : 8, ndb); \
            hipLaunchKernelGGL(knl, grid, block, lds, (hipStream_t)stream, a);                                            \
            return DLWP_OK;                                                                                               \
        };                                                                                                                \
        int rc3 = DLWP_OK;                                                                                                \
        const bool bf = dlwp_get_gemm_precision() == 1;      /* bf16 matrix arithmetic asked for: bf16 MFMA operands */   \
        if (bf && ndb == 3) {                       /* head dims 33 .. 48, windows of at most 64 tokens (small_applies) */  \
            rc3 = go(KERNEL<4, 3, true, true>);                                                                           \
        } else if (bf) {                                                                                                  \
            if (ndb == 1) {                                                                                               \
                if (nc <= 4) rc3 = vec ? go(KERNEL<4, 1, true, true>) : go(KERNEL<4, 1, false, true>);                    \
                else rc3 = vec ? go(KERNEL<8, 1, true, true>) : go(KERNEL<8, 1, false, true>);                            \
            } else {                                                                                                      \
                if (nc <= 4) rc3 = vec ? go(KERNEL<4, 2, true, true>) : go(KERNEL<4, 2, false, true>);                    \
                else rc3 = vec ? go(KERNEL<8, 2, true, true>) : go(KERNEL<8, 2, false, true>);                            \
            }                                                                                                             \
        } else if (ndb == 1) {                                                                                            \
            if (nc <= 4) rc3 = vec ? go(KERNEL<4, 1, true, false>) : go(KERNEL<4, 1, false, false>);                      \
            else rc3 = vec ? go(KERNEL<8, 1, true, false>) : go(KERNEL<8, 1, false, false>);                              \
        } else {                                                                                                          \
            if (nc <= 4) rc3 = vec ? go(KERNEL<4, 2, true, false>) : go(KERNEL<4, 2, false, false>);                      \
            else rc3 = vec ? go(KERNEL<8, 2, true, false>) : go(KERNEL<8, 2, false, false>);                              \
        }                                                                                                                 \
        if (rc3) return rc3;                                                                                              \
    } while (0)

int dlwp_winattn_small_fwd(const float* qkv, const float* table, const float* packed, const int* ia, const int* ib,
                           const int* labels, float* out, float* lse, int B_, int nW, int N, int TB, int ntypes, int heads, int d,
                           float scale, int q_lo, int q_hi, void* stream, int io_bf16) {
    WsDev a{};
    ws_setup(a, B_, nW, N, TB, ntypes, heads, d, scale, q_lo, q_hi);
    a.qkv = qkv; a.table = table; a.table_t = packed; a.ia = ia; a.ib = ib; a.labels = labels; a.out = out; a.lse = lse;
    a.io_bf16 = io_bf16 != 0;
    const size_t lds = sizeof(float) * (size_t)TB;
    if (io_bf16) {
        const int ndb = (d + 15) / 16;
        const dim3 grid((unsigned)(heads * ntypes * a.groups)), block(256);
        auto go = [&](auto knl) -> int {
            int rc2 = dlwp_ensure_lds(reinterpret_cast<const void*>(knl), lds, "winattn_small_fwd_kernel");
            if (rc2) return rc2;
            dlwp_prof_scope prof((hipStream_t)stream, ws_prof_flops(a, 2), ws_prof_bytes(a, false), "winattn_small_fwd_kernel<4, %d> (bf16)", ndb);
            hipLaunchKernelGGL(knl, grid, block, lds, (hipStream_t)stream, a);
            return DLWP_OK;
        };
        const int rc3 = ndb == 1 ? go(winattn_small_fwd_kernel<4, 1, true, true, false, true>)
                      : ndb == 2 ? go(winattn_small_fwd_kernel<4, 2, true, true, false, true>) : go(winattn_small_fwd_kernel<4, 3, true, true, false, true>);
        if (rc3) return rc3;
        DLWP_LAUNCH_CHECK();
        return DLWP_OK;
    }
    WS_DISPATCH(winattn_small_fwd_kernel, lds);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

// the one-pass kernel (round 4) wherever its LDS image fits and the table slice fits the staging area it is folded in
// Windows of at most 64 tokens (Swin's 49: four key chunks, the 4-wave instantiation) stay on the two-pass kernel unless asked for:
// with ~4 windows per workgroup the per-workgroup fold of the dense image (2401 LDS float atomics onto 169 table slots) outweighs
// the saved score evaluations -- Swin C4 step 6.74 ms one-pass vs 6.59 ms two-pass, same box, back to back.
static bool one_pass_applies(int N, int d, int TB) {
    const int nc = (N + 15) / 16;
    if (d > 32) return false;
    if (nc <= 4 && !dlwp_tune_on("WINATTN_BWD1P_SMALL")) return false;
    return lds_family_applies(N, d) && lds2_bytes(16 * nc, TB) <= 160 * 1024 && (size_t)TB * 4 <= (size_t)4 * 16 * nc * LDB * 2;
}
// bf16 tensors in the WINDOW layout (round 6: qkv, out, gout, gqkv all bf16 arrays; dlwp_window_attn_fwd_bf16 / _bwd_bf16): windows of at
// most 64 tokens in the bf16 matrix mode -- the wave-per-window forward with raw bf16 fragments and the two-pass LDS-staged backward
bool dlwp_winattn_io_bf16_applies(int N, int d, int TB, long long pairs) {
    return N <= 64 && d % 4 == 0 && dlwp_get_gemm_precision() == 1 && dlwp_winattn_small_applies(N, d, pairs) && lds_family_applies(N, d) &&
           !(one_pass_applies(N, d, TB) && !dlwp_tune_on("WINATTN_BWD2PASS")) && !dlwp_tune_on("WINATTN_TILED");
}

static int one_pass_launch(WsDev& a, void* stream) {
    const int nc = (a.N + 15) / 16, nw1 = nc > 4 ? 8 : 4, heads = a.heads, ntypes = a.ntypes, d = a.d;
    DLWP_REQUIRE((long long)(a.dst_map ? a.Ltok : a.N) * 3 * heads * d < (1LL << 31), DLWP_E_UNSUPPORTED,
                 "window attention backward: a sample's qkv tensor must stay below 2^31 elements (32-bit row offsets)");
    const size_t lb1 = lds2_bytes(16 * nc, a.TB);
    const int want = dlwp_tune_or("WINATTN_WG_BWD", 0);
    a.dbg = dlwp_tune_or("WINATTN_DBG", 0);
    // one round of workgroups: a workgroup's table staging, dense-image fold and flush cost ~36 k cycles (stamps), a window
    // ~12 k, so the windows of a (type, head) are split over as many workgroups as fit on the chip at once and no more
    // (Pangu C4 layer 1, 114 pairs x 37 windows: 228 workgroups 177 us, 456: 202, 798: 252)
    static const int ncu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    const int slots = ncu * (nw1 == 8 ? 1 : 2);
    const long long per = want ? ((long long)want + heads * ntypes - 1) / (heads * ntypes) : slots / (heads * ntypes);
    a.groups = per < 1 ? 1 : (per > a.M ? a.M : (int)per);
    const dim3 grid((unsigned)(heads * ntypes * a.groups));
    auto go1 = [&](auto knl, int nt) -> int {
        int rc2 = dlwp_ensure_lds(reinterpret_cast<const void*>(knl), lb1, "winattn_lds_bwd1p");
        if (rc2) return rc2;
        dlwp_prof_scope prof((hipStream_t)stream, ws_prof_flops(a, 5), ws_prof_bytes(a, true), "winattn_lds_bwd1p_kernel<%d, %d, %s>",
                             d <= 16 ? 1 : 2, nw1, a.io_bf16 ? "true" : "false");
        hipLaunchKernelGGL(knl, grid, dim3(nt), lb1, (hipStream_t)stream, a);
        return DLWP_OK;
    };
    int rc1;
    if (a.io_bf16) {
        if (nw1 == 8) rc1 = d <= 16 ? go1(winattn_lds_bwd1p_kernel<1, 8, true>, 512) : go1(winattn_lds_bwd1p_kernel<2, 8, true>, 512);
        else rc1 = d <= 16 ? go1(winattn_lds_bwd1p_kernel<1, 4, true>, 256) : go1(winattn_lds_bwd1p_kernel<2, 4, true>, 256);
    } else {
        if (nw1 == 8) rc1 = d <= 16 ? go1(winattn_lds_bwd1p_kernel<1, 8>, 512) : go1(winattn_lds_bwd1p_kernel<2, 8>, 512);
        else rc1 = d <= 16 ? go1(winattn_lds_bwd1p_kernel<1, 4>, 256) : go1(winattn_lds_bwd1p_kernel<2, 4>, 256);
    }
    if (rc1) return rc1;
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

int dlwp_winattn_small_bwd(const float* qkv, const float* table, const float* packed, const int* ia, const int* ib,
                           const int* labels, const float* out, const float* lse, const float* gout, float* gqkv, float* gtable,
                           int B_, int nW, int N, int TB, int ntypes, int heads, int d, float scale, int q_lo, int q_hi, void* stream, int io_bf16) {
    WsDev a{};
    ws_setup(a, B_, nW, N, TB, ntypes, heads, d, scale, q_lo, q_hi);
    a.qkv = qkv; a.table = table; a.table_t = packed; a.ia = ia; a.ib = ib; a.labels = labels; a.o = out; a.lse_in = lse;
    a.gout = gout; a.gqkv = gqkv; a.gtable = gtable;
    a.io_bf16 = io_bf16 != 0;
    DLWP_REQUIRE(!io_bf16 || dlwp_winattn_io_bf16_applies(N, d, TB, (long long)B_ * heads), DLWP_E_UNSUPPORTED,
                 "window attention backward: bf16 tensors in the window layout need windows of at most 64 tokens in the bf16 matrix mode (N %d, d %d)", N, d);
    if (lds_family_applies(N, d)) {
        const int nc = (N + 15) / 16;
        if (!io_bf16 && !dlwp_tune_on("WINATTN_BWD2PASS") && one_pass_applies(N, d, TB)) return one_pass_launch(a, stream);
        const size_t lb = lds_bytes(16 * nc, TB, true, d <= 32 ? LDB : LDB48);
        a.groups = lds_groups(a.M, heads, ntypes, d <= 32 ? 4 : 3);
        const dim3 grid((unsigned)(heads * ntypes * a.groups)), block(256);
        auto go = [&](auto knl) -> int {
            int rc2 = dlwp_ensure_lds(reinterpret_cast<const void*>(knl), lb, "winattn_lds_bwd");
            if (rc2) return rc2;
            dlwp_prof_scope prof((hipStream_t)stream, ws_prof_flops(a, 5), ws_prof_bytes(a, true), "winattn_lds_bwd_kernel<%d>", d <= 16 ? 1 : d <= 32 ? 2 : 3);
            hipLaunchKernelGGL(knl, grid, block, lb, (hipStream_t)stream, a);
            return DLWP_OK;
        };
        const int rc3 = d <= 16 ? go(winattn_lds_bwd_kernel<1>) : d <= 32 ? go(winattn_lds_bwd_kernel<2>) : go(winattn_lds_bwd_kernel<3>);
        if (rc3) return rc3;
        DLWP_LAUNCH_CHECK();
        return DLWP_OK;
    }
    const size_t lds = sizeof(float) * (3 * (size_t)((TB + 1) & ~1) + 4 * 128);
    const bool vec = d % 4 == 0;
    const dim3 grid((unsigned)(heads * ntypes * a.groups)), block(256);
    auto go = [&](auto knl) -> int {
        int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(knl), lds, "winattn_small_bwd");
        if (rc) return rc;
        dlwp_prof_scope prof((hipStream_t)stream, ws_prof_flops(a, 5), ws_prof_bytes(a, true), "winattn_small_bwd_kernel<%d>", d <= 16 ? 1 : 2);
        hipLaunchKernelGGL(knl, grid, block, lds, (hipStream_t)stream, a);
        return DLWP_OK;
    };
    int rc;
    if (dlwp_get_gemm_precision() == 1) {
        if (d <= 16) rc = vec ? go(winattn_small_bwd_kernel<1, true, true>) : go(winattn_small_bwd_kernel<1, false, true>);
        else rc = vec ? go(winattn_small_bwd_kernel<2, true, true>) : go(winattn_small_bwd_kernel<2, false, true>);
    } else {
        if (d <= 16) rc = vec ? go(winattn_small_bwd_kernel<1, true, false>) : go(winattn_small_bwd_kernel<1, false, false>);
        else rc = vec ? go(winattn_small_bwd_kernel<2, true, false>) : go(winattn_small_bwd_kernel<2, false, false>);
    }
    if (rc) return rc;
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

// ---- token-layout entries (include/dlwpmi.h: dlwp_window_attn_fwd_tokens / dlwp_window_attn_bwd_tokens)
extern "C" int dlwp_window_attn_bwd_tokens_supported(int N, int d, int TB) {
    return one_pass_applies(N, d, TB) && !dlwp_tune_on("WINATTN_BWD2PASS") ? 1 : 0;
}
extern "C" int dlwp_window_attn_fwd_tokens_supported(int N, int d, long long pairs) {
    return d <= 32 && pairs >= 2048 && dlwp_winattn_small_applies(N, d, pairs) && d % 4 == 0 && dlwp_get_gemm_precision() == 1 && !dlwp_tune_on("WINATTN_TILED") ? 1 : 0;
}
static int tokens_check(const char* who, int B_, int nW, int N, int Ltok, int TB, int ntypes, int heads, int d, int q_lo, int q_hi) {
    DLWP_REQUIRE(B_ > 0 && nW > 0 && B_ % nW == 0 && N > 0 && Ltok > 0 && heads > 0 && d > 0 && ntypes > 0 && TB > 0, DLWP_E_INVALID,
                 "%s: bad shape (B_ %d, nW %d, N %d, Ltok %d, heads %d, d %d)", who, B_, nW, N, Ltok, heads, d);
    DLWP_REQUIRE(B_ % ntypes == 0 && nW % ntypes == 0, DLWP_E_INVALID, "%s: %d window types do not divide nW = %d", who, ntypes, nW);
    DLWP_REQUIRE(q_lo >= 0 && q_hi <= N && q_lo < q_hi, DLWP_E_INVALID, "%s: query range [%d, %d) outside [0, %d)", who, q_lo, q_hi, N);
    return DLWP_OK;
}
extern "C" int dlwp_window_attn_fwd_tokens(const float* qkv_tokens, const float* fill, const float* bias_table, const float* packed_table,
                                           const int* ia, const int* ib, const int* labels, const int* src_map, const int* dst_map,
                                           float* out_tokens, float* lse, int B_, int nW, int N, int Ltok, int TB, int ntypes, int heads,
                                           int d, float scale, int q_lo, int q_hi, int io_bf16, void* stream) {
    DLWP_REQUIRE(qkv_tokens && fill && bias_table && ia && ib && src_map && dst_map && out_tokens && lse, DLWP_E_INVALID,
                 "window_attn_fwd_tokens: NULL argument");
    int rc = tokens_check("window_attn_fwd_tokens", B_, nW, N, Ltok, TB, ntypes, heads, d, q_lo, q_hi);
    if (rc) return rc;
    DLWP_REQUIRE(dlwp_window_attn_fwd_tokens_supported(N, d, (long long)B_ * heads), DLWP_E_UNSUPPORTED,
                 "window_attn_fwd_tokens: needs the bf16 matrix mode, N <= 128, head_dim <= 32 and %% 4 == 0, at least 2048 (window, head) pairs "
                 "(N %d, d %d, pairs %lld)", N, d, (long long)B_ * heads);
    WsDev a{};
    ws_setup(a, B_, nW, N, TB, ntypes, heads, d, scale, q_lo, q_hi);
    a.qkv = qkv_tokens; a.fill = fill; a.table = bias_table; a.table_t = packed_table; a.ia = ia; a.ib = ib; a.labels = labels;
    a.src_map = src_map; a.dst_map = dst_map; a.out = out_tokens; a.lse = lse; a.Ltok = Ltok; a.io_bf16 = io_bf16 != 0;
    const int nc = (N + 15) / 16;
    DLWP_REQUIRE((long long)Ltok * 3 * heads * d < (1LL << 31), DLWP_E_UNSUPPORTED,
                 "window_attn_fwd_tokens: a sample's qkv tensor must stay below 2^31 elements (32-bit row offsets)");
    if (dlwp_tune_or("WINATTN_FWD_LDS", 1)) {
        // the LDS-staged forward (winattn_lds_fwd_tok_kernel; default: 43 vs 56 us per launch in the Pangu C4 step): one round of
        // four-wave workgroups, two per CU (Pangu C4 step: 228 workgroups 10.97 ms, 456: 10.58, 912: 10.64, 1824: 10.72)
        const int NR = 16 * nc;
        const size_t lb = (size_t)3 * NR * LDB * 2 + (size_t)4 * NR * 4 + (size_t)((TB + 3) & ~3) * 4 + 96 * 4;
        static const int ncu = [] {
            int dev = 0, n = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
            return n > 0 ? n : 256;
        }();
        const int per_cu = std::max(1, std::min(2, (int)((size_t)150 * 1024 / lb)));
        const int want = dlwp_tune_or("WINATTN_WG_FWD", 0);
        const long long per = want ? ((long long)want + heads * ntypes - 1) / (heads * ntypes) : (long long)ncu * per_cu / (heads * ntypes);
        a.groups = per < 1 ? 1 : (per > a.M ? a.M : (int)per);
        const dim3 grid1((unsigned)(heads * ntypes * a.groups));
        auto go1 = [&](auto knl) -> int {
            int rc2 = dlwp_ensure_lds(reinterpret_cast<const void*>(knl), lb, "winattn_lds_fwd_tok");
            if (rc2) return rc2;
            dlwp_prof_scope prof((hipStream_t)stream, ws_prof_flops(a, 2), ws_prof_bytes(a, false), "winattn_lds_fwd_tok_kernel<%d, %d, %s>",
                                 nc <= 4 ? 4 : 8, d <= 16 ? 1 : 2, io_bf16 ? "true" : "false");
            hipLaunchKernelGGL(knl, grid1, dim3(256), lb, (hipStream_t)stream, a);
            return DLWP_OK;
        };
        if (io_bf16) {
            if (d <= 16) rc = nc <= 4 ? go1(winattn_lds_fwd_tok_kernel<4, 1, true>) : go1(winattn_lds_fwd_tok_kernel<8, 1, true>);
            else rc = nc <= 4 ? go1(winattn_lds_fwd_tok_kernel<4, 2, true>) : go1(winattn_lds_fwd_tok_kernel<8, 2, true>);
        } else {
            if (d <= 16) rc = nc <= 4 ? go1(winattn_lds_fwd_tok_kernel<4, 1, false>) : go1(winattn_lds_fwd_tok_kernel<8, 1, false>);
            else rc = nc <= 4 ? go1(winattn_lds_fwd_tok_kernel<4, 2, false>) : go1(winattn_lds_fwd_tok_kernel<8, 2, false>);
        }
        if (rc) return rc;
        DLWP_LAUNCH_CHECK();
        return DLWP_OK;
    }
    const size_t lds = sizeof(float) * (size_t)((TB + 3) & ~3) + sizeof(int) * 4 * 256 + sizeof(float) * 96;
    const dim3 grid((unsigned)(heads * ntypes * a.groups)), block(256);
    auto go = [&](auto knl) -> int {
        int rc2 = dlwp_ensure_lds(reinterpret_cast<const void*>(knl), lds, "winattn_small_fwd_tokens");
        if (rc2) return rc2;
        dlwp_prof_scope prof((hipStream_t)stream, ws_prof_flops(a, 2), ws_prof_bytes(a, false), "winattn_small_fwd_kernel<%d, %d> (tokens)",
                             nc <= 4 ? 4 : 8, d <= 16 ? 1 : 2);
        hipLaunchKernelGGL(knl, grid, block, lds, (hipStream_t)stream, a);
        return DLWP_OK;
    };
    if (io_bf16) {
        if (d <= 16) rc = nc <= 4 ? go(winattn_small_fwd_kernel<4, 1, true, true, true, true>) : go(winattn_small_fwd_kernel<8, 1, true, true, true, true>);
        else rc = nc <= 4 ? go(winattn_small_fwd_kernel<4, 2, true, true, true, true>) : go(winattn_small_fwd_kernel<8, 2, true, true, true, true>);
    } else {
        if (d <= 16) rc = nc <= 4 ? go(winattn_small_fwd_kernel<4, 1, true, true, true>) : go(winattn_small_fwd_kernel<8, 1, true, true, true>);
        else rc = nc <= 4 ? go(winattn_small_fwd_kernel<4, 2, true, true, true>) : go(winattn_small_fwd_kernel<8, 2, true, true, true>);
    }
    if (rc) return rc;
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}
extern "C" int dlwp_window_attn_bwd_tokens(const float* qkv, const float* fill, const float* bias_table, const float* packed_table,
                                           const int* ia, const int* ib, const int* labels, const float* out, const float* lse,
                                           const float* gout_tokens, const int* dst_map, const int* src_map, float* gqkv_tokens,
                                           float* gfill, float* gbias_table, int B_, int nW, int N, int Ltok, int TB, int ntypes,
                                           int heads, int d, float scale, int q_lo, int q_hi, int io_bf16, void* stream) {
    DLWP_REQUIRE(qkv && bias_table && ia && ib && out && lse && gout_tokens && dst_map && src_map && gqkv_tokens && gfill && gbias_table,
                 DLWP_E_INVALID, "window_attn_bwd_tokens: NULL argument");
    DLWP_REQUIRE(!io_bf16 || fill, DLWP_E_INVALID, "window_attn_bwd_tokens: bf16 tensors need the token-layout operands (fill != NULL)");
    const int rc = tokens_check("window_attn_bwd_tokens", B_, nW, N, Ltok, TB, ntypes, heads, d, q_lo, q_hi);
    if (rc) return rc;
    DLWP_REQUIRE(dlwp_window_attn_bwd_tokens_supported(N, d, TB), DLWP_E_UNSUPPORTED,
                 "window_attn_bwd_tokens: needs the bf16 matrix mode, N <= 128, head_dim <= 32 and %% 4 == 0 (N %d, d %d, table %d)", N, d, TB);
    WsDev a{};
    ws_setup(a, B_, nW, N, TB, ntypes, heads, d, scale, q_lo, q_hi);
    a.qkv = qkv; a.fill = fill; a.table = bias_table; a.table_t = packed_table; a.ia = ia; a.ib = ib; a.labels = labels; a.o = out;
    a.lse_in = lse; a.gout = gout_tokens; a.gqkv = gqkv_tokens; a.gtable = gbias_table;
    a.src_map = src_map; a.dst_map = dst_map; a.gfill = gfill; a.Ltok = Ltok; a.io_bf16 = io_bf16 != 0;
    return one_pass_launch(a, stream);
}

#ifdef DLWP_STAMPS
extern "C" int dlwp_debug_stamps_winattn_small(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dlwp_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : 1;
}
#endif

// Pointwise two-layer channel MLP on channels-first fields (the FNO lifting / projection
// networks: 1x1 conv -> GELU -> 1x1 conv).  Reference call sites: neuralop.FNO.lifting /
// .projection constructed at src/nsbench/models/fno/fno.py:19-27,205-215 and
// src/dlwpbench/models/fno/fno.py:38-47 (third-party arithmetic).
//
// MI355X design: one workgroup = 64 consecutive pixels of one sample x all channels.
// Both GEMMs run on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32); the hidden activations
// (Ch x 64) never leave registers: the accumulator of GEMM1 (rows = hidden channel 4g+j,
// col = pixel) is directly the B operand of GEMM2 when the A operand (W2) is read with the
// matching permuted-k order (common.hip.h: mfma16_chunk).  Backward recomputes the hidden
// layer, keeps weight-gradient partials in registers over the 64 pixels and leaves them in a
// per-workgroup slab in accumulator-tile order (plain 16-byte stores; one slab-parallel launch
// folds every slab of a training step, fold_slabs_kernel below); without a slab they go out as
// hardware float atomics (API path).  The backward kernel's waves own their hidden blocks alone, so they read their weight
// fragments straight from L2 into registers (no LDS images; measured equal to the staged form, 22.1 / 20.6 us, and it frees
// 55 KB of LDS); the forward kernel's four pixel-block waves share every fragment, there the LDS images stay (the register
// form re-read each fragment four times through L1: 14.9 / 12.4 us against 11.4 / 10.1).  The forward kernel can append the W-axis DFT of its output
// rows (lifting -> first spectral block), the backward kernel the adjoint DFT of its input
// gradient rows (projection backward -> last spectral block's backward).
#include "common.hip.h"
#include "dlwpmi_internal.h"
#include "fno_rows.hip.h"

namespace {

constexpr int PT = 64;        // pixels per workgroup
constexpr int FWD_WAVES = 16; // 4 waves per SIMD: the forward kernel needs 54 VGPRs and its main loop is bound by the dependent
                              // z -> GELU -> second-GEMM chain of a wave, not by the matrix pipe (8 waves: 1993, 16 waves: 2007 samples/s)
constexpr int BWD_WAVES = 8;   // (4 waves with two hidden blocks in flight each, 352 VGPRs, was measured: 1884 samples/s against 2020;
                               // plain 4 waves: 1858)
constexpr int LDP = PT + 4;   // LDS row stride of pixel tiles (4*LDP % 32 == 16: conflict-free B reads)

__device__ __forceinline__ const float* chan_ptr(const dlwp_chan_src& s, int b, int c) {
    if (s.tab) return s.tab[c] ? s.tab[c] + (long long)b * s.tab_bstride[c] : nullptr;
    return s.base + (long long)b * s.bstride + (long long)c * s.cstride;
}
__device__ __forceinline__ float* chan_ptr(const dlwp_chan_dst& s, int b, int c) {
    if (s.tab) return s.tab[c] ? s.tab[c] + (long long)b * s.tab_bstride[c] : nullptr;
    return s.base ? s.base + (long long)b * s.bstride + (long long)c * s.cstride : nullptr;
}


// MFMA fragments of a dense row-major [R][K] matrix straight from global memory (L2 / L1 resident weights), zero outside the
// matrix.  Loads are unconditional from clamped addresses with the select afterwards, so that a group of them is in flight
// together.  frag_row: element s = M[row][k + s] (vec: K % 4 == 0 and a 16-byte aligned base, k a multiple of 4);
// frag_col: element s = M[row + s][col].
__device__ __forceinline__ f32x4 frag_row(const float* __restrict__ M, int R, int K, int row, int k, bool vec) {
    f32x4 v;
    if (vec) {
        const bool ok = row < R && k < K;
        v = *reinterpret_cast<const f32x4*>(M + (ok ? (long long)row * K + k : 0));
        if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool ok = row < R && k + s < K;
            const float x = M[ok ? (long long)row * K + k + s : 0];
            v[s] = ok ? x : 0.f;
        }
    }
    return v;
}
__device__ __forceinline__ f32x4 frag_col(const float* __restrict__ M, int R, int K, int row, int col) {
    f32x4 v;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bool ok = row + s < R && col < K;
        const float x = M[ok ? (long long)(row + s) * K + col : 0];
        v[s] = ok ? x : 0.f;
    }
    return v;
}

// pixel tile [C_pad][PT] of a channel view -> LDS (row stride LDP); 16-byte loads along pixels
// (channels in `skip` are left alone: a fused producer in the same workgroup writes them)
template <typename SRC>
__device__ __forceinline__ void stage_pixels(float* dst, const SRC& s, int b, int C, int C_pad, int p0, int P,
                                             bool vec_ok, unsigned long long skip = 0ull) {
    if (vec_ok) {
#pragma unroll 2
        for (int u = threadIdx.x; u < C_pad * (PT / 4); u += blockDim.x) {
            const int c = u / (PT / 4), q = u % (PT / 4);
            if (c < 64 && ((skip >> c) & 1ull)) continue;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < C && p0 + 4 * q < P) {
                const float* src = chan_ptr(s, b, c);
                if (src) v = *reinterpret_cast<const float4*>(src + p0 + 4 * q);
            }
            *reinterpret_cast<float4*>(&dst[c * LDP + 4 * q]) = v;
        }
    } else {
        for (int idx = threadIdx.x; idx < C_pad * PT; idx += blockDim.x) {
            const int c = idx / PT, p = idx % PT;
            if (c < 64 && ((skip >> c) & 1ull)) continue;
            float v = 0.f;
            if (c < C && p0 + p < P) {
                const float* src = chan_ptr(s, b, c);
                if (src) v = src[p0 + p];
            }
            dst[c * LDP + p] = v;
        }
    }
}

struct FwdArgs {
    dlwp_chan_src x;
    const float *w1, *b1, *w2, *b2;
    dlwp_chan_dst y;
    dlwp_chan_src res;  // optional residual added to y (base==nullptr && tab==nullptr: none)
    int B, Cin, Ch, Cout, P, tiles_per_sample;
    int Cin_pad, Ch_pad, Cout_pad;
    int vec_w, vec_x;   // 16-byte loads allowed for the weights / the pixel planes
    FastDiv dCin, dCh;
    // optional fused W-axis DFT of the output row (lifting MLP -> first spectral block): needs PT == W, NP == 16
    float2* x1_out;
    const float* FT;
    int rows_H, m2c;
};

// ---- forward MLP of one 64-pixel tile, in pieces that a kernel strings together (the single-MLP kernel below; the chained
// projection -> next lifting kernel further down): issue (global loads into registers) -> commit (LDS images) -> barrier ->
// compute (both GEMMs) -> barrier -> finish (reduce the hidden groups' partial tiles, bias, residual, store, optional rows DFT).
struct FwdLds {
    float *xs, *w1s, *w2s, *b1s, *b2s, *outs;
};
// LDS floats of one MLP: input tile, the two weight images, biases (the partial output tiles `outs` are counted separately)
__host__ __device__ __forceinline__ size_t fwd_lds_floats(int Cin_pad, int Ch_pad, int Cout_pad) {
    return (size_t)Cin_pad * LDP + (size_t)Ch_pad * (Cin_pad + 4) + (size_t)Cout_pad * (Ch_pad + 4) + Ch_pad + Cout_pad;
}
__device__ __forceinline__ FwdLds fwd_carve(float* base, const FwdArgs& a, float* outs) {
    FwdLds L;
    L.xs = base;
    L.w1s = L.xs + a.Cin_pad * LDP;
    L.w2s = L.w1s + a.Ch_pad * (a.Cin_pad + 4);
    L.b1s = L.w2s + a.Cout_pad * (a.Ch_pad + 4);
    L.b2s = L.b1s + a.Ch_pad;
    L.outs = outs ? outs : L.b2s + a.Cout_pad;      // [HQ][Cout_pad][LDP] per-hidden-group partial output tiles
    return L;
}
struct FwdLoads {
    MatLoad<4> l2;
    MatLoad<2> l1;
    float4 ftv;          // DFT table [16][PT] of the fused rows step
    int n1, n2;
};
__device__ __forceinline__ void fwd_issue(const FwdArgs& a, FwdLoads& q) {
    q.n2 = matload_units(a.Cout, a.Ch, a.vec_w != 0, 4);
    q.n1 = matload_units(a.Ch, a.Cin, a.vec_w != 0, 2);
    q.l2.issue(a.w2, q.n2);
    q.l1.issue(a.w1, q.n1);
    q.ftv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.x1_out && threadIdx.x < 16 * (PT / 4)) q.ftv = reinterpret_cast<const float4*>(a.FT)[threadIdx.x];
}
__device__ __forceinline__ void fwd_commit(const FwdArgs& a, const FwdLds& L, const FwdLoads& q) {
    const int LD1 = a.Cin_pad + 4, LD2 = a.Ch_pad + 4, tid = threadIdx.x, NT = blockDim.x;
    q.l2.commit<false>(L.w2s, LD2, a.Ch, a.dCh, q.n2);
    q.l1.commit<false>(L.w1s, LD1, a.Cin, a.dCin, q.n1);
    stage_matrix_tail<false>(L.w2s, LD2, a.w2, a.Cout, a.Ch, a.dCh, q.n2);
    stage_matrix_tail<false>(L.w1s, LD1, a.w1, a.Ch, a.Cin, a.dCin, q.n1);
    zero_padding(L.w1s, LD1, a.Ch, a.Cin, a.Ch_pad, a.Cin_pad);
    zero_padding(L.w2s, LD2, a.Cout, a.Ch, a.Cout_pad, a.Ch_pad);
    for (int idx = tid; idx < a.Ch_pad; idx += NT) L.b1s[idx] = idx < a.Ch ? a.b1[idx] : 0.f;
    for (int idx = tid; idx < a.Cout_pad; idx += NT) L.b2s[idx] = idx < a.Cout ? a.b2[idx] : 0.f;
}
// NW waves: wave w owns pixel block (w & 3) and the hidden blocks hq, hq+HQ, ... (hq = w >> 2,
// HQ = NW/4); two or more waves per SIMD overlap one wave's GELU (VALU) with another's MFMAs.
template <int NOB, int NW>
__device__ __forceinline__ void fwd_compute(const FwdArgs& a, const FwdLds& L) {
    constexpr int HQ = NW / 4;
    const int LD1 = a.Cin_pad + 4, LD2 = a.Ch_pad + 4;
    const int lane = lane_id(), w = wave_id(), r = lane & 15, g = lane >> 4;
    const int pw0 = (w & 3) * 16, hq = w >> 2;
    const int nkc = a.Cin_pad / 16, nhb = a.Ch_pad / 16;
    f32x4 oacc[NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) oacc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int hb = hq; hb < nhb; hb += HQ) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < nkc; ++kc) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(&L.w1s[(hb * 16 + r) * LD1 + kc * 16 + 4 * g]);
            f32x4 b4;
#pragma unroll
            for (int s = 0; s < 4; ++s) b4[s] = L.xs[(kc * 16 + 4 * g + s) * LDP + pw0 + r];
            z = mfma16_chunk(a4, b4, z);
        }
        {
            const f32x4 zb = z + *reinterpret_cast<const f32x4*>(&L.b1s[hb * 16 + 4 * g]);
            f32x4 dz;
            gelu_both4(zb, z, dz);                      // packed fp32 polynomial (the derivative half is dead code here)
        }
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(&L.w2s[(ob * 16 + r) * LD2 + hb * 16 + 4 * g]);
            oacc[ob] = mfma16_chunk(w4, z, oacc[ob]);
        }
    }
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            L.outs[(hq * a.Cout_pad + ob * 16 + 4 * g + j) * LDP + pw0 + r] = oacc[ob][j];
        }
}
// next_xs / next_ch: a chained MLP's input tile in LDS and, per output channel, the input channel of that tile the finished
// values are copied to (-1: none) -- the consumer then never reads them back from global memory
template <int NOB, int NW>
__device__ __forceinline__ void fwd_finish(const FwdArgs& a, const FwdLds& L, const float4 ftv, int b, int p0, int tile,
                                           float* next_xs = nullptr, const signed char* next_ch = nullptr) {
    constexpr int NT = NW * 64, HQ = NW / 4;
    const int tid = threadIdx.x, w = wave_id();
    float* outs = L.outs;
    const bool has_res = a.res.base || a.res.tab;
    if (a.vec_x) {
        for (int u = tid; u < a.Cout * (PT / 4); u += NT) {
            const int o = u / (PT / 4), q = u % (PT / 4), p = p0 + 4 * q;
            if (p < a.P) {
                float4 v = *reinterpret_cast<const float4*>(&outs[o * LDP + 4 * q]);
#pragma unroll
                for (int h2 = 1; h2 < HQ; ++h2) {
                    const float4 pv = *reinterpret_cast<const float4*>(&outs[(h2 * a.Cout_pad + o) * LDP + 4 * q]);
                    v.x += pv.x; v.y += pv.y; v.z += pv.z; v.w += pv.w;
                }
                const float bb = L.b2s[o];
                v.x += bb; v.y += bb; v.z += bb; v.w += bb;
                if (has_res) {
                    const float* rp = chan_ptr(a.res, b, o);
                    if (rp) {
                        const float4 rv = *reinterpret_cast<const float4*>(rp + p);
                        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                    }
                }
                float* dst = chan_ptr(a.y, b, o);
                if (dst) *reinterpret_cast<float4*>(dst + p) = v;
                if (a.x1_out) *reinterpret_cast<float4*>(&outs[o * LDP + 4 * q]) = v;   // finished row tile, in place
                if (next_xs && o < 8 && next_ch[o] >= 0) *reinterpret_cast<float4*>(&next_xs[next_ch[o] * LDP + 4 * q]) = v;
            }
        }
        if (a.x1_out) {
            // W-axis pruned DFT of the finished row, so that the first spectral block needs no separate rows kernel.
            // The W2 image is dead: it hosts the table and the per-wave partial spectra.
            float* ft = L.w2s;                       // [16][LDP]
            float* x1s = L.w2s + 16 * LDP;           // [4 waves][Cout_pad][16]
            if (tid < 16 * (PT / 4)) *reinterpret_cast<float4*>(&ft[(tid / (PT / 4)) * LDP + 4 * (tid % (PT / 4))]) = ftv;
            for (int idx = a.Cout * LDP + tid; idx < a.Cout_pad * LDP; idx += NT) outs[idx] = 0.f;
            __syncthreads();
            if (w < 4) tile_rows_dft<NOB, 1>(outs, ft, x1s, LDP, 16, PT / 16, false);
            __syncthreads();
            store_x1(x1s, a.x1_out, b, tile, a.rows_H, a.m2c, a.Cout, a.Cout_pad, 16);
        }
    } else {
        for (int idx = tid; idx < a.Cout * PT; idx += NT) {
            const int o = idx / PT, p = p0 + idx % PT;
            if (p < a.P) {
                float v = L.b2s[o];
#pragma unroll
                for (int h2 = 0; h2 < HQ; ++h2) v += outs[(h2 * a.Cout_pad + o) * LDP + idx % PT];
                if (has_res) {
                    const float* rp = chan_ptr(a.res, b, o);
                    if (rp) v += rp[p];
                }
                float* dst = chan_ptr(a.y, b, o);
                if (dst) dst[p] = v;
            }
        }
    }
}

template <int NOB, int NW>
__global__ __launch_bounds__(NW * 64) void pwmlp_fwd_kernel(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const FwdLds L = fwd_carve(smem, a, nullptr);
    const int b = blockIdx.x / a.tiles_per_sample, tile = blockIdx.x % a.tiles_per_sample;
    const int p0 = tile * PT;
    DLWP_STAMP(0);
    // issue every independent global load first, then fill the LDS images
    FwdLoads q;
    fwd_issue(a, q);
    DLWP_STAMP(1);
    stage_pixels(L.xs, a.x, b, a.Cin, a.Cin_pad, p0, a.P, a.vec_x != 0);
    DLWP_STAMP(2);
    fwd_commit(a, L, q);
    DLWP_STAMP(4);
    __syncthreads();
    DLWP_STAMP(5);
    fwd_compute<NOB, NW>(a, L);
    DLWP_STAMP(6);
    __syncthreads();
    DLWP_STAMP(7);
    fwd_finish<NOB, NW>(a, L, q.ftv, b, p0, tile);
    DLWP_STAMP(8);
}

// Projection of net call k chained with the lifting of net call k + 1 (both pointwise, the same 64-pixel tile): the rollout's
// closed loop feeds the frame the projection has just produced straight into the next lifting layer's input tile, in LDS;
// one launch, one prologue (both weight sets are requested together) instead of two, and the frame is not read back.
struct ChainArgs {
    FwdArgs p, l;
    signed char next_ch[8];        // output channel o of the first MLP -> input channel of the second (-1: none)
    unsigned long long skip;       // input channels of the second MLP that come from the first
};
template <int NOB1, int NOB2, int NW>
__global__ __launch_bounds__(NW * 64) void pwmlp_fwd_chain_kernel(ChainArgs c) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int HQ = NW / 4;
    // first MLP with its partial tiles, then the second MLP whose partial tiles alias the first one's W1 image (dead by then)
    const FwdLds P = fwd_carve(smem, c.p, nullptr);
    float* base2 = P.outs + HQ * c.p.Cout_pad * LDP;
    const FwdLds Lq = fwd_carve(base2, c.l, P.w1s);
    const int b = blockIdx.x / c.p.tiles_per_sample, tile = blockIdx.x % c.p.tiles_per_sample;
    const int p0 = tile * PT;
    FwdLoads qp, ql;
    fwd_issue(c.p, qp);
    fwd_issue(c.l, ql);
    stage_pixels(P.xs, c.p.x, b, c.p.Cin, c.p.Cin_pad, p0, c.p.P, c.p.vec_x != 0);
    stage_pixels(Lq.xs, c.l.x, b, c.l.Cin, c.l.Cin_pad, p0, c.l.P, c.l.vec_x != 0, c.skip);
    fwd_commit(c.p, P, qp);
    fwd_commit(c.l, Lq, ql);
    __syncthreads();
    fwd_compute<NOB1, NW>(c.p, P);
    __syncthreads();
    fwd_finish<NOB1, NW>(c.p, P, qp.ftv, b, p0, tile, Lq.xs, c.next_ch);
    __syncthreads();
    fwd_compute<NOB2, NW>(c.l, Lq);
    __syncthreads();
    fwd_finish<NOB2, NW>(c.l, Lq, ql.ftv, b, p0, tile);
}

struct BwdArgs {
    dlwp_chan_src x;
    const float *w1, *b1, *w2;
    dlwp_chan_src gy;            // upstream gradient (may be absent)
    dlwp_chan_src pred, target;  // optional MSE term: gy_eff += mse_scale * (pred - target)
    float mse_scale;
    dlwp_chan_dst gx;            // nullable; per-channel nullable in table mode
    int gx_accumulate;
    dlwp_chan_dst gres;          // optional: gres += effective upstream gradient (identity path of a residual)
    float *gw1, *gb1, *gw2, *gb2;  // accumulated with float atomics (slab == nullptr)
    // optional per-workgroup partial slab [grid][slab_stride] laid out {gw1,gb1,gw2,gb2}: plain
    // stores (slab_accumulate == 0) or read-modify-write by the owning workgroup; deterministic and
    // not bound by the chip-wide float-atomic rate.  dlwp_pwmlp_slab_reduce folds it into the grads.
    float* slab;
    long long slab_stride;
    int slab_accumulate;
    int B, Cin, Ch, Cout, P, tiles_per_sample;
    int Cin_pad, Ch_pad, Cout_pad;
    int vec_w, vec_x;
    FastDiv dCin, dCh;
    // optional fused W-axis DFT (adjoint table) of the finished gx row -- projection backward -> last spectral block's
    // backward without a separate rows launch; needs PT == W, NP == 16, dense gx, gx_accumulate == 0
    float2* x1_out;
    const float* FT;
    int rows_H, m2c;
};

__device__ __forceinline__ void grad_flush(float* slab_ptr, float* grad_ptr, bool accumulate, float v) {
    if (slab_ptr) *slab_ptr = accumulate ? *slab_ptr + v : v;
    else atomic_add_f32(grad_ptr, v);
}

template <int NIB, int NOB, int NW>
__global__ __launch_bounds__(NW * 64) void pwmlp_bwd_kernel(BwdArgs a) {
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                        // [Cin_pad][LDP]
    float* gys = xs + a.Cin_pad * LDP;       // [Cout_pad][LDP]
    float* red = gys + a.Cout_pad * LDP;     // [NW][Cin_pad][LDP] gx partial tiles; later the rows-DFT table / partial spectra

    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int r = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.tiles_per_sample;
    const int p0 = (blockIdx.x % a.tiles_per_sample) * PT;

    const int nhb = a.Ch_pad / 16;
    const bool vec1 = a.vec_w != 0 && (a.Cin & 3) == 0;
    DLWP_STAMP(9);
    // weight fragments of one hidden block, straight from L2: W1 rows (B operand of z^T), b1, W2 columns of the block (B operand
    // of g_a^T), W1 columns (A operand of dX).  A wave owns its hidden blocks alone, so every fragment is read once per workgroup.
    struct HbW {
        f32x4 w1b[NIB], w2c[NOB], w1c[NIB];
        float b1v;
    };
    auto load_hb = [&](int hb, HbW& h) {
#pragma unroll
        for (int kc = 0; kc < NIB; ++kc) h.w1b[kc] = frag_row(a.w1, a.Ch, a.Cin, hb * 16 + r, kc * 16 + 4 * g, vec1);
#pragma unroll
        for (int oc = 0; oc < NOB; ++oc) h.w2c[oc] = frag_col(a.w2, a.Cout, a.Ch, oc * 16 + 4 * g, hb * 16 + r);
#pragma unroll
        for (int ib = 0; ib < NIB; ++ib) h.w1c[ib] = frag_col(a.w1, a.Ch, a.Cin, hb * 16 + 4 * g, ib * 16 + r);
        const bool okh = hb * 16 + r < a.Ch;
        const float bv = a.b1[okh ? hb * 16 + r : 0];
        h.b1v = okh ? bv : 0.f;
    };
    HbW wcur;
    load_hb(w < nhb ? w : 0, wcur);                        // in flight while the pixel tiles are staged
    float4 ftv = make_float4(0.f, 0.f, 0.f, 0.f);          // DFT table [16][PT] of the fused rows step
    if (a.x1_out && tid < 16 * (PT / 4)) ftv = reinterpret_cast<const float4*>(a.FT)[tid];
    stage_pixels(xs, a.x, b, a.Cin, a.Cin_pad, p0, a.P, a.vec_x != 0);
    const bool has_gy = a.gy.base || a.gy.tab;
    const bool has_mse = a.pred.base || a.pred.tab;
    if (has_gy && !has_mse) {
        stage_pixels(gys, a.gy, b, a.Cout, a.Cout_pad, p0, a.P, a.vec_x != 0);
    } else {
        for (int idx = tid; idx < a.Cout_pad * PT; idx += NT) {
            const int c = idx / PT, p = idx % PT;
            float v = 0.f;
            if (c < a.Cout && p0 + p < a.P) {
                if (has_gy) {
                    const float* src = chan_ptr(a.gy, b, c);
                    if (src) v = src[p0 + p];
                }
                if (has_mse) {
                    const float* pp = chan_ptr(a.pred, b, c);
                    const float* tp = chan_ptr(a.target, b, c);
                    v += a.mse_scale * (pp[p0 + p] - tp[p0 + p]);
                }
            }
            gys[c * LDP + p] = v;
        }
    }
    __syncthreads();
    if (a.gres.base || a.gres.tab) {
        for (int idx = tid; idx < a.Cout * PT; idx += NT) {
            const int o = idx / PT, p = idx % PT;
            float* dst = chan_ptr(a.gres, b, o);
            if (dst && p0 + p < a.P) dst[p0 + p] += gys[o * LDP + p];
        }
    }

    DLWP_STAMP(10);

    f32x4 gxacc[4][NIB];
#pragma unroll
    for (int pb = 0; pb < 4; ++pb)
#pragma unroll
        for (int ib = 0; ib < NIB; ++ib) gxacc[pb][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
    // identity as a permuted-k B operand: I[k = 4g+s][n = r]
    f32x4 ident;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) ident[s2] = (4 * g + s2 == r) ? 1.f : 0.f;

    float* sl = a.slab ? a.slab + (long long)blockIdx.x * a.slab_stride : nullptr;
    // slab layout per workgroup: [nhb][NOB][64][4] dW2 tiles | [nhb][NIB][64][4] dW1 tiles | db1[Ch_pad] | db2[Cout_pad]
    const long long o_t1 = (long long)(a.Ch_pad / 16) * NOB * 256, o_gb1 = o_t1 + (long long)(a.Ch_pad / 16) * NIB * 256;
    const long long o_gb2 = o_gb1 + a.Ch_pad;
    const bool accum = sl && a.slab_accumulate != 0;
    DLWP_STAMP(11);
    // Everything below works in the TRANSPOSED orientation (rows = pixels 4g+j, column = hidden channel r):
    // the accumulator of z^T / gz^T is then directly the B operand of the pixel contractions (dW2) and, read as an
    // A operand, the hidden-major matrix for dW1; the one product that contracts over the hidden index (dX) gets
    // its operand through a 4-MFMA multiplication with the identity instead of an LDS round trip.
    // One hidden block's state while its four pixel blocks are processed.
    struct HbState {
        f32x4 pgw2[NOB], pgw1[NIB];   // running slab partials (read-modify-write accumulation across net calls)
        float pgb1;
        f32x4 gw2acc[NOB], gw1acc[NIB];
        float gb1acc;
        float *t2, *t1;
        int hb;
    };
    auto hb_begin = [&](HbState& st, int hb) {
        st.hb = hb;
        // The slab is private scratch, so it is laid out in ACCUMULATOR-TILE order: tile (hb, ob) holds the 4
        // registers of every lane contiguously -> one 16-byte access per lane, 1 KiB per wave-instruction
        // (narrow 64-byte-segment stores were issue-bound: ~350 cycles each).  The old values are fetched here.
        st.pgb1 = 0.f;
        st.t2 = sl ? sl + ((long long)hb * NOB * 64 + lane) * 4 : nullptr;
        st.t1 = sl ? sl + o_t1 + ((long long)hb * NIB * 64 + lane) * 4 : nullptr;
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
            st.pgw2[ob] = accum ? *reinterpret_cast<const f32x4*>(st.t2 + ob * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ib = 0; ib < NIB; ++ib)
            st.pgw1[ib] = accum ? *reinterpret_cast<const f32x4*>(st.t1 + ib * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (accum && g == 0) st.pgb1 = sl[o_gb1 + hb * 16 + r];
        st.gb1acc = 0.f;
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) st.gw2acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ib = 0; ib < NIB; ++ib) st.gw1acc[ib] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // One pixel block in two stages, software-pipelined by the caller: stage A (z^T and g_a^T, MFMA only) of block pb + 1 is
    // issued in front of stage B (GELU on the VALU, then the products that need it) of block pb, and the k-steps of
    // independent accumulators are interleaved (a 16x16x4 MFMA that accumulates into its predecessor's result issues every 40
    // cycles instead of 32).  Measured: neither changes the kernel (22.4 / 20.4 us before and after) -- on one SIMD the exact-f32
    // MFMA and ordinary VALU instructions do not co-execute at all (tools/micro/mfma_valu_coexec.hip: 257 k + 177 k cycles
    // apart, 433 k together), so the main loop costs MFMA cycles PLUS GELU cycles however they are ordered.
    auto hb_stage_a = [&](const HbW& hw, int pb, f32x4& zt, f32x4& gat) {
        // z^T[p][h] = sum_i x[i][p] W1[h][i];  g_a^T[p][h] = sum_o gy[o][p] W2[o][h]
        f32x4 az[NIB], ag[NOB];
#pragma unroll
        for (int kc = 0; kc < NIB; ++kc)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) az[kc][s2] = xs[(kc * 16 + 4 * g + s2) * LDP + pb * 16 + r];
#pragma unroll
        for (int oc = 0; oc < NOB; ++oc)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) ag[oc][s2] = gys[(oc * 16 + 4 * g + s2) * LDP + pb * 16 + r];
        zt = f32x4{0.f, 0.f, 0.f, 0.f};
        gat = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int NMAX = NIB > NOB ? NIB : NOB;
#pragma unroll
        for (int c = 0; c < NMAX; ++c)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                if (c < NIB) zt = mfma16(az[c][s2], hw.w1b[c][s2], zt);
                if (c < NOB) gat = mfma16(ag[c][s2], hw.w2c[c][s2], gat);
            }
    };
    auto hb_stage_b = [&](HbState& st, const HbW& hw, int pb, const f32x4 zt, const f32x4 gat) {
        f32x4 actt, gzt, dvt;
        gelu_both4(zt + f32x4{hw.b1v, hw.b1v, hw.b1v, hw.b1v}, actt, dvt);      // packed fp32 polynomial
        gzt = gat * dvt;
        st.gb1acc += (gzt[0] + gzt[1]) + (gzt[2] + gzt[3]);
        // dW2[o][h] += sum_p gy[o][p] act^T[p][h];  dW1[h][i] += sum_p gz[h][p] x[i][p]  (gz^T registers read as an A operand
        // are gz);  gz[h][p] in accumulator layout = gz^T (as A operand) x identity
        f32x4 a2[NOB], b1f[NIB];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) a2[ob] = *reinterpret_cast<const f32x4*>(&gys[(ob * 16 + r) * LDP + pb * 16 + 4 * g]);
#pragma unroll
        for (int ib = 0; ib < NIB; ++ib) b1f[ib] = *reinterpret_cast<const f32x4*>(&xs[(ib * 16 + r) * LDP + pb * 16 + 4 * g]);
        f32x4 gz = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            gz = mfma16(gzt[s2], ident[s2], gz);
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) st.gw2acc[ob] = mfma16(a2[ob][s2], actt[s2], st.gw2acc[ob]);
#pragma unroll
            for (int ib = 0; ib < NIB; ++ib) st.gw1acc[ib] = mfma16(gzt[s2], b1f[ib][s2], st.gw1acc[ib]);
        }
        // dX[i][p] += sum_h W1[h][i] gz[h][p]
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int ib = 0; ib < NIB; ++ib) gxacc[pb][ib] = mfma16(hw.w1c[ib][s2], gz[s2], gxacc[pb][ib]);
    };
    auto hb_flush = [&](HbState& st) {
        const int hb = st.hb;
        if (sl) {
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) {
                f32x4 v = st.gw2acc[ob];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += st.pgw2[ob][j];
                *reinterpret_cast<f32x4*>(st.t2 + ob * 256) = v;
            }
#pragma unroll
            for (int ib = 0; ib < NIB; ++ib) {
                f32x4 v = st.gw1acc[ib];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += st.pgw1[ib][j];
                *reinterpret_cast<f32x4*>(st.t1 + ib * 256) = v;
            }
        } else {
            const int hcol = hb * 16 + r;
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int o = ob * 16 + 4 * g + j;
                    if (o < a.Cout && hcol < a.Ch) atomic_add_f32(a.gw2 + o * a.Ch + hcol, st.gw2acc[ob][j]);
                }
#pragma unroll
            for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int h = hb * 16 + 4 * g + j, i = ib * 16 + r;
                    if (h < a.Ch && i < a.Cin) atomic_add_f32(a.gw1 + h * a.Cin + i, st.gw1acc[ib][j]);
                }
        }
        float v = st.gb1acc;                       // bias gradient of hidden channel hb*16 + r: sum the 4 lane groups
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        const int h = hb * 16 + r;
        if (g == 0) {
            if (sl) sl[o_gb1 + h] = st.pgb1 + v;
            else if (h < a.Ch) atomic_add_f32(a.gb1 + h, v);
        }
    };
    {
        // Two waves share each SIMD's matrix pipe; in-kernel stamps showed waves 0-3 leaving this loop after 20 k cycles and
        // waves 4-7 after 29 k, with the barrier behind it waiting for the slowest.  The younger half runs the FIRST half of
        // its hidden blocks at raised priority and the second half at the default (moves the skew by < 2 k cycles).
        const int hb_half = w + NW * (((nhb - w + NW - 1) / NW) / 2);     // first hidden block of this wave's second half
        if (w >= NW / 2) __builtin_amdgcn_s_setprio(1);
        for (int hb = w; hb < nhb; hb += NW) {
            if (w >= NW / 2 && hb == hb_half) __builtin_amdgcn_s_setprio(0);
            HbState st;
            hb_begin(st, hb);
            if (hb == w) DLWP_STAMP(12);
            f32x4 zt, gat;
            hb_stage_a(wcur, 0, zt, gat);
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                f32x4 ztn = zt, gatn = gat;
                if (pb < 3) hb_stage_a(wcur, pb + 1, ztn, gatn);
                hb_stage_b(st, wcur, pb, zt, gat);
                zt = ztn; gat = gatn;
            }
            if (hb == w) DLWP_STAMP(18);
            // the next block's fragments are requested here, behind the last use of the current ones: their latency overlaps the
            // slab stores and the next hb_begin, and they add no live registers to the pixel-block loop
            if (hb + NW < nhb) load_hb(hb + NW, wcur);
            hb_flush(st);
            if (hb == w) DLWP_STAMP(19);
        }
    }
    __builtin_amdgcn_s_setprio(0);
    DLWP_STAMP(20);
    DLWP_STAMP_WAVE(24);      // end of every wave's main loop (slots 24..31)
    // cross-wave reduction of gx: every wave parks its [Cin_pad][64] partial tile in LDS (the weight
    // images are dead by now and are reused), then all threads sum the NW tiles.  (LDS float atomics
    // run at <1 lane-op per clock per CU on gfx950 and were the slowest phase of this kernel.)
    const bool want_gx = a.gx.base || a.gx.tab;
    DLWP_STAMP(20);
    __syncthreads();
    if (want_gx) {
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
            for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    red[(w * a.Cin_pad + ib * 16 + 4 * g + j) * LDP + pb * 16 + r] = gxacc[pb][ib][j];
    }
    DLWP_STAMP(21);
    __syncthreads();
    DLWP_STAMP(22);
    if (want_gx) {
        for (int u = tid; u < a.Cin * (PT / 4); u += NT) {
            const int c = u / (PT / 4), q = u % (PT / 4), p = p0 + 4 * q;
            float* dst = chan_ptr(a.gx, b, c);
            if (dst && p < a.P) {
                float4 v = *reinterpret_cast<const float4*>(&red[c * LDP + 4 * q]);
#pragma unroll
                for (int w2 = 1; w2 < NW; ++w2) {
                    const float4 pv = *reinterpret_cast<const float4*>(&red[(w2 * a.Cin_pad + c) * LDP + 4 * q]);
                    v.x += pv.x; v.y += pv.y; v.z += pv.z; v.w += pv.w;
                }
                if (a.vec_x) {
                    float4* d4 = reinterpret_cast<float4*>(dst + p);
                    if (a.gx_accumulate) { const float4 o = *d4; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                    *d4 = v;
                    if (a.x1_out) *reinterpret_cast<float4*>(&xs[c * LDP + 4 * q]) = v;   // finished row tile (x is dead)
                } else {
                    const float vv[4] = {v.x, v.y, v.z, v.w};
                    for (int k = 0; k < 4; ++k)
                        if (p + k < a.P) dst[p + k] = a.gx_accumulate ? dst[p + k] + vv[k] : vv[k];
                }
            }
        }
    }
    if (tid < a.Cout) {
        float s = 0.f;
        for (int p = 0; p < PT; ++p) s += gys[tid * LDP + p];
        grad_flush(sl ? sl + o_gb2 + tid : nullptr, a.gb2 + tid, accum, s);
    }
    if (a.x1_out) {
        // W-axis pruned DFT of the finished gradient row.  The partial-tile region is dead after the barrier: it hosts the
        // table and the per-wave partial spectra.
        __syncthreads();
        float* x1s = red;                          // [4 waves][Cin_pad][16]
        float* ft = red + 4 * a.Cin_pad * 16;      // [16][LDP]
        if (tid < 16 * (PT / 4)) *reinterpret_cast<float4*>(&ft[(tid / (PT / 4)) * LDP + 4 * (tid % (PT / 4))]) = ftv;
        for (int idx = a.Cin * LDP + tid; idx < a.Cin_pad * LDP; idx += NT) xs[idx] = 0.f;
        __syncthreads();
        if (w < 4) tile_rows_dft<NIB, 1>(xs, ft, x1s, LDP, 16, PT / 16, false);
        __syncthreads();
        store_x1(x1s, a.x1_out, b, (int)(blockIdx.x % a.tiles_per_sample), a.rows_H, a.m2c, a.Cin, a.Cin_pad, 16);
    }
    DLWP_STAMP(23);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
template <typename V>
bool view_vec_ok(const V& v) {
    if (v.tab) return true;  // table views are built by the trainer from 16-byte aligned planes
    if (!v.base) return true;
    return aligned16(v.base) && v.bstride % 4 == 0 && v.cstride % 4 == 0;
}

// ---- one launch that folds EVERY partial slab of a training step into the gradient buffer.
// (The first version folded each slab set with its own launch, every thread walking the slab dimension serially: 256
// dependent-free but serial loads per thread, 12-17 us per launch, six launches per step.)  A workgroup owns 64 consecutive slab offsets (coalesced 256-byte rows) and its four
// waves each sum a quarter of the slabs, 16 loads in flight per lane, combined through LDS; the destination of an offset
// is found by inverting the slab layout (tile order of pwmlp_bwd, plain order of the skip-weight slabs).
constexpr int FOLD_MAX_JOBS = 8;
struct FoldJobDev {
    const float* slab;
    long long stride;           // floats between consecutive slabs
    int nslab, kind;            // kind 0: plain {d1[n1] | d2[n2]}; 1: pwmlp accumulator-tile layout
    int blk0, nblk;             // block range of this job inside the launch
    float *d1, *d2, *d3, *d4;   // kind 0: d1, d2;  kind 1: gw1, gb1, gw2, gb2
    int n1, n2;                 // kind 0 segment lengths
    int Cin, Ch, Cout, nob, nib, Ch_pad, Cout_pad;
};
struct FoldArgs {
    FoldJobDev j[FOLD_MAX_JOBS];
    int njobs;
};

__device__ __forceinline__ float* fold_dst(const FoldJobDev& J, long long off) {
    if (J.kind == 0) {
        if (off < J.n1) return J.d1 + off;
        if (off < (long long)J.n1 + J.n2) return J.d2 + (off - J.n1);
        return nullptr;
    }
    const long long o_t1 = (long long)(J.Ch_pad / 16) * J.nob * 256, o_gb1 = o_t1 + (long long)(J.Ch_pad / 16) * J.nib * 256;
    const long long o_gb2 = o_gb1 + J.Ch_pad;
    if (off < o_t1) {            // dW2 tile (hb, ob): lane = (o%16/4)*16 + h%16, register j = o%4
        const int j = (int)(off & 3), lane = (int)((off >> 2) & 63), t = (int)(off >> 8);
        const int hb = t / J.nob, ob = t - hb * J.nob;
        const int o = ob * 16 + (lane >> 4) * 4 + j, h = hb * 16 + (lane & 15);
        return (o < J.Cout && h < J.Ch) ? J.d3 + (long long)o * J.Ch + h : nullptr;
    }
    if (off < o_gb1) {           // dW1 tile (hb, ib): lane = (h%16/4)*16 + i%16, register j = h%4
        const long long e = off - o_t1;
        const int j = (int)(e & 3), lane = (int)((e >> 2) & 63), t = (int)(e >> 8);
        const int hb = t / J.nib, ib = t - hb * J.nib;
        const int h = hb * 16 + (lane >> 4) * 4 + j, i = ib * 16 + (lane & 15);
        return (h < J.Ch && i < J.Cin) ? J.d1 + (long long)h * J.Cin + i : nullptr;
    }
    if (off < o_gb2) {
        const int h = (int)(off - o_gb1);
        return h < J.Ch ? J.d2 + h : nullptr;
    }
    const int o = (int)(off - o_gb2);
    return o < J.Cout ? J.d4 + o : nullptr;
}

__global__ __launch_bounds__(256) void fold_slabs_kernel(FoldArgs a) {
    __shared__ float part[4][64];
    int ji = 0;
#pragma unroll
    for (int k = 1; k < FOLD_MAX_JOBS; ++k)
        if (k < a.njobs && (int)blockIdx.x >= a.j[k].blk0) ji = k;
    const FoldJobDev& J = a.j[ji];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long off = (long long)((int)blockIdx.x - J.blk0) * 64 + lane;
    float* dst = off < J.stride ? fold_dst(J, off) : nullptr;
    float acc = 0.f;
    if (dst) {
        const int per = (J.nslab + 3) / 4, s0 = w * per, s1 = min(J.nslab, s0 + per);
        const float* p = J.slab + off;
        int s = s0;
        for (; s + 16 <= s1; s += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = p[(long long)(s + u) * J.stride];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += v[u];
        }
        for (; s < s1; ++s) acc += p[(long long)s * J.stride];
    }
    part[w][lane] = acc;
    __syncthreads();
    if (w == 0 && dst) *dst += (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
    return dlwp_ensure_lds(reinterpret_cast<const void*>(kernel), bytes, "pwmlp");
}

}  // namespace

int dlwp_pwmlp_fwd_ex(const dlwp_chan_src* x, const float* w1, const float* b1, const float* w2,
                      const float* b2, const dlwp_chan_dst* y, const dlwp_chan_src* res, int B,
                      int Cin, int Ch, int Cout, int P, hipStream_t stream) {
    return dlwp_pwmlp_fwd_rows_ex(x, w1, b1, w2, b2, y, res, B, Cin, Ch, Cout, P, nullptr, nullptr, stream);
}

bool dlwp_pwmlp_rows_fusable(const dlwp_fno_plan* p, int Cout, int P) {
    return p && p->W == PT && p->NP == 16 && p->C == Cout && P == p->H * p->W && p->C_pad <= 64;
}

// fills and validates the launch arguments of one forward MLP; rows_after: the rows DFT was asked for but the epilogue
// cannot host it (the caller runs the separate rows kernel)
static int fwd_make_args(FwdArgs& a, const dlwp_chan_src* x, const float* w1, const float* b1, const float* w2,
                         const float* b2, const dlwp_chan_dst* y, const dlwp_chan_src* res, int B, int Cin, int Ch, int Cout,
                         int P, const dlwp_fno_plan* rows_plan, float2* x1_out, bool& rows_after) {
    DLWP_REQUIRE(B > 0 && Cin > 0 && Ch > 0 && Cout > 0 && P > 0, DLWP_E_INVALID,
                 "pwmlp_fwd: non-positive dimension");
    a = FwdArgs{};
    a.x = *x; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.y = *y;
    if (res) a.res = *res;
    a.B = B; a.Cin = Cin; a.Ch = Ch; a.Cout = Cout; a.P = P;
    a.tiles_per_sample = ceil_div(P, PT);
    a.Cin_pad = round_up(Cin, 16); a.Ch_pad = round_up(Ch, 16); a.Cout_pad = round_up(Cout, 16);
    a.vec_w = aligned16(w1) && aligned16(w2);
    a.dCin = make_fastdiv(Cin); a.dCh = make_fastdiv(Ch);
    a.vec_x = P % 4 == 0 && view_vec_ok(a.x);
    const int nob = a.Cout_pad / 16;
    DLWP_REQUIRE(nob <= 4 && a.Cin_pad <= 64, DLWP_E_UNSUPPORTED,
                 "pwmlp_fwd: Cin<=64 and Cout<=64 supported (got %d, %d)", Cin, Cout);
    a.vec_x = a.vec_x && view_vec_ok(a.y) && view_vec_ok(a.res);
    rows_after = false;
    if (x1_out) {
        DLWP_REQUIRE(rows_plan && y->base, DLWP_E_INVALID, "pwmlp_fwd: the rows DFT needs a plan and a dense output");
        const bool fuse = dlwp_pwmlp_rows_fusable(rows_plan, Cout, P) && a.vec_x &&
                          (size_t)a.Cout_pad * (a.Ch_pad + 4) >= (size_t)16 * LDP + (size_t)4 * a.Cout_pad * 16;
        if (fuse) { a.x1_out = x1_out; a.FT = rows_plan->FT_fwd; a.rows_H = rows_plan->H; a.m2c = rows_plan->m2c; }
        else rows_after = true;       // shapes the epilogue cannot host: separate rows kernel, same result
    }
    return DLWP_OK;
}

int dlwp_pwmlp_fwd_rows_ex(const dlwp_chan_src* x, const float* w1, const float* b1, const float* w2,
                           const float* b2, const dlwp_chan_dst* y, const dlwp_chan_src* res, int B,
                           int Cin, int Ch, int Cout, int P, const dlwp_fno_plan* rows_plan, float2* x1_out,
                           hipStream_t stream) {
    FwdArgs a;
    bool rows_after;
    int rc = fwd_make_args(a, x, w1, b1, w2, b2, y, res, B, Cin, Ch, Cout, P, rows_plan, x1_out, rows_after);
    if (rc) return rc;
    const int nob = a.Cout_pad / 16;
    const size_t lds = sizeof(float) * (fwd_lds_floats(a.Cin_pad, a.Ch_pad, a.Cout_pad) + (size_t)(FWD_WAVES / 4) * a.Cout_pad * LDP);
    constexpr int NW = FWD_WAVES;
    const dim3 grid(B * a.tiles_per_sample), block(NW * 64);
#define LAUNCH(N)                                                             \
    if ((rc = set_lds(pwmlp_fwd_kernel<N, NW>, lds)) != DLWP_OK) return rc;   \
    hipLaunchKernelGGL((pwmlp_fwd_kernel<N, NW>), grid, block, lds, stream, a);
    switch (nob) {
        case 1: LAUNCH(1) break;
        case 2: LAUNCH(2) break;
        case 3: LAUNCH(3) break;
        default: LAUNCH(4) break;
    }
#undef LAUNCH
    DLWP_LAUNCH_CHECK();
    if (rows_after) return dlwp_fno_rows_dft(rows_plan, y->base, 0, 0, x1_out, B, stream);
    return DLWP_OK;
}

// Two forward MLPs on the same pixel tiles in ONE launch (projection of a net call -> lifting of the next one): the second
// MLP's input channels that ARE output channels of the first (matched by plane pointer and batch stride in the caller's
// channel table) are handed over in LDS.  Returns DLWP_E_UNSUPPORTED without touching the error string when the shapes do not
// allow the chain (the caller then issues the two launches separately).
int dlwp_pwmlp_fwd_chain_ex(const dlwp_chan_src* x1v, const float* w11, const float* b11, const float* w12, const float* b12,
                            const dlwp_chan_dst* y1, const dlwp_chan_src* res1, int Cin1, int Ch1, int Cout1,
                            const dlwp_chan_src* x2v, const float* w21, const float* b21, const float* w22, const float* b22,
                            const dlwp_chan_dst* y2, int Cin2, int Ch2, int Cout2, const signed char* next_ch, int B, int P,
                            const dlwp_fno_plan* rows_plan, float2* x1_out, hipStream_t stream) {
    ChainArgs c{};
    bool ra1, ra2;
    int rc = fwd_make_args(c.p, x1v, w11, b11, w12, b12, y1, res1, B, Cin1, Ch1, Cout1, P, nullptr, nullptr, ra1);
    if (rc) return rc;
    if ((rc = fwd_make_args(c.l, x2v, w21, b21, w22, b22, y2, nullptr, B, Cin2, Ch2, Cout2, P, rows_plan, x1_out, ra2))) return rc;
    constexpr int NW = FWD_WAVES, HQ = NW / 4;
    const size_t lds = sizeof(float) * (fwd_lds_floats(c.p.Cin_pad, c.p.Ch_pad, c.p.Cout_pad) + (size_t)HQ * c.p.Cout_pad * LDP +
                                        fwd_lds_floats(c.l.Cin_pad, c.l.Ch_pad, c.l.Cout_pad));
    const bool alias_ok = (size_t)HQ * c.l.Cout_pad * LDP <= (size_t)c.p.Ch_pad * (c.p.Cin_pad + 4);
    const int nob1 = c.p.Cout_pad / 16, nob2 = c.l.Cout_pad / 16;
    if (ra2 || !c.p.vec_x || !c.l.vec_x || Cout1 > 8 || !alias_ok || lds > 160 * 1024 || nob1 != 1 || nob2 > 2) return DLWP_E_UNSUPPORTED;
    c.skip = 0;
    for (int o = 0; o < 8; ++o) {
        c.next_ch[o] = o < Cout1 ? next_ch[o] : (signed char)-1;
        if (c.next_ch[o] >= 0) {
            if (c.next_ch[o] >= Cin2) return DLWP_E_UNSUPPORTED;
            c.skip |= 1ull << c.next_ch[o];
        }
    }
    const dim3 grid(B * c.p.tiles_per_sample), block(NW * 64);
    if (nob2 == 1) {
        if ((rc = set_lds(pwmlp_fwd_chain_kernel<1, 1, NW>, lds)) != DLWP_OK) return rc;
        hipLaunchKernelGGL((pwmlp_fwd_chain_kernel<1, 1, NW>), grid, block, lds, stream, c);
    } else {
        if ((rc = set_lds(pwmlp_fwd_chain_kernel<1, 2, NW>, lds)) != DLWP_OK) return rc;
        hipLaunchKernelGGL((pwmlp_fwd_chain_kernel<1, 2, NW>), grid, block, lds, stream, c);
    }
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

int dlwp_pwmlp_bwd_ex(const dlwp_chan_src* x, const float* w1, const float* b1, const float* w2,
                      const dlwp_chan_src* gy, const dlwp_chan_src* pred, const dlwp_chan_src* target,
                      float mse_scale, const dlwp_chan_dst* gx, int gx_accumulate, const dlwp_chan_dst* gres,
                      float* gw1, float* gb1, float* gw2, float* gb2, float* slab, int slab_accumulate, int B,
                      int Cin, int Ch, int Cout, int P, hipStream_t stream) {
    return dlwp_pwmlp_bwd_rows_ex(x, w1, b1, w2, gy, pred, target, mse_scale, gx, gx_accumulate, gres, gw1, gb1, gw2, gb2,
                                  slab, slab_accumulate, B, Cin, Ch, Cout, P, nullptr, nullptr, stream);
}

int dlwp_pwmlp_bwd_rows_ex(const dlwp_chan_src* x, const float* w1, const float* b1, const float* w2,
                           const dlwp_chan_src* gy, const dlwp_chan_src* pred, const dlwp_chan_src* target,
                           float mse_scale, const dlwp_chan_dst* gx, int gx_accumulate, const dlwp_chan_dst* gres,
                           float* gw1, float* gb1, float* gw2, float* gb2, float* slab, int slab_accumulate, int B,
                           int Cin, int Ch, int Cout, int P, const dlwp_fno_plan* rows_plan, float2* x1_out,
                           hipStream_t stream) {
    DLWP_REQUIRE(B > 0 && Cin > 0 && Ch > 0 && Cout > 0 && P > 0, DLWP_E_INVALID,
                 "pwmlp_bwd: non-positive dimension");
    BwdArgs a{};
    a.x = *x; a.w1 = w1; a.b1 = b1; a.w2 = w2;
    if (gy) a.gy = *gy;
    if (pred) { a.pred = *pred; a.target = *target; }
    a.mse_scale = mse_scale;
    if (gx) a.gx = *gx;
    a.gx_accumulate = gx_accumulate;
    if (gres) a.gres = *gres;
    a.gw1 = gw1; a.gb1 = gb1; a.gw2 = gw2; a.gb2 = gb2;
    a.B = B; a.Cin = Cin; a.Ch = Ch; a.Cout = Cout; a.P = P;
    a.tiles_per_sample = ceil_div(P, PT);
    a.Cin_pad = round_up(Cin, 16); a.Ch_pad = round_up(Ch, 16); a.Cout_pad = round_up(Cout, 16);
    a.slab = slab; a.slab_stride = dlwp_pwmlp_slab_stride(Cin, Ch, Cout); a.slab_accumulate = slab_accumulate;
    a.vec_w = aligned16(w1);
    a.dCin = make_fastdiv(Cin); a.dCh = make_fastdiv(Ch);
    a.vec_x = P % 4 == 0 && view_vec_ok(a.x) && view_vec_ok(a.gy);
    const int nib = a.Cin_pad / 16, nob = a.Cout_pad / 16;
    DLWP_REQUIRE(nib <= 4 && nob <= 4, DLWP_E_UNSUPPORTED,
                 "pwmlp_bwd: Cin<=64 and Cout<=64 supported (got %d, %d)", Cin, Cout);
    bool rows_after = false;
    if (x1_out) {
        DLWP_REQUIRE(rows_plan && gx && gx->base && !gx_accumulate, DLWP_E_INVALID,
                     "pwmlp_bwd: the rows DFT needs a plan and a dense, overwritten gx");
        const bool fuse = dlwp_pwmlp_rows_fusable(rows_plan, Cin, P) && a.vec_x && view_vec_ok(a.gx) &&
                          gx->bstride == (long long)Cin * P && gx->cstride == P &&
                          (size_t)BWD_WAVES * a.Cin_pad * LDP >= (size_t)4 * a.Cin_pad * 16 + (size_t)16 * LDP;
        if (fuse) { a.x1_out = x1_out; a.FT = rows_plan->FT_adj; a.rows_H = rows_plan->H; a.m2c = rows_plan->m2c; }
        else rows_after = true;       // shapes the epilogue cannot host: separate rows kernel, same result
    }
    const size_t red = (size_t)BWD_WAVES * a.Cin_pad * LDP;  // gx partial tiles (host the rows-DFT scratch afterwards)
    const size_t lds = sizeof(float) * ((size_t)a.Cin_pad * LDP + (size_t)a.Cout_pad * LDP + red);
    constexpr int NW = BWD_WAVES;
    const dim3 grid(B * a.tiles_per_sample), block(NW * 64);
    int rc;
#define LAUNCH(I, O)                                                           \
    if ((rc = set_lds(pwmlp_bwd_kernel<I, O, NW>, lds)) != DLWP_OK) return rc; \
    hipLaunchKernelGGL((pwmlp_bwd_kernel<I, O, NW>), grid, block, lds, stream, a);
#define ROW(I)                       \
    switch (nob) {                   \
        case 1: LAUNCH(I, 1) break;  \
        case 2: LAUNCH(I, 2) break;  \
        case 3: LAUNCH(I, 3) break;  \
        default: LAUNCH(I, 4) break; \
    }
    switch (nib) {
        case 1: ROW(1) break;
        case 2: ROW(2) break;
        case 3: ROW(3) break;
        default: ROW(4) break;
    }
#undef ROW
#undef LAUNCH
    DLWP_LAUNCH_CHECK();
    if (rows_after) return dlwp_fno_rows_dft(rows_plan, gx->base, 0, 1, x1_out, B, stream);
    return DLWP_OK;
}

long long dlwp_pwmlp_slab_stride(int Cin, int Ch, int Cout) {
    const long long nhb = round_up(Ch, 16) / 16, nib = round_up(Cin, 16) / 16, nob = round_up(Cout, 16) / 16;
    return nhb * (nob + nib) * 256 + round_up(Ch, 16) + round_up(Cout, 16);
}

int dlwp_pwmlp_slab_count(int B, int P) { return B * ceil_div(P, PT); }

int dlwp_fold_slabs(const dlwp_fold_job* jobs, int njobs, hipStream_t stream) {
    DLWP_REQUIRE(jobs && njobs >= 1 && njobs <= FOLD_MAX_JOBS, DLWP_E_INVALID, "fold_slabs: 1..%d jobs", FOLD_MAX_JOBS);
    FoldArgs a{};
    a.njobs = njobs;
    int blocks = 0;
    for (int k = 0; k < njobs; ++k) {
        const dlwp_fold_job& q = jobs[k];
        FoldJobDev& J = a.j[k];
        DLWP_REQUIRE(q.slab && q.nslab > 0, DLWP_E_INVALID, "fold_slabs: job %d has no slab", k);
        J.slab = q.slab; J.nslab = q.nslab; J.kind = q.pwmlp ? 1 : 0;
        if (q.pwmlp) {
            J.stride = dlwp_pwmlp_slab_stride(q.Cin, q.Ch, q.Cout);
            J.Cin = q.Cin; J.Ch = q.Ch; J.Cout = q.Cout;
            J.nob = round_up(q.Cout, 16) / 16; J.nib = round_up(q.Cin, 16) / 16;
            J.Ch_pad = round_up(q.Ch, 16); J.Cout_pad = round_up(q.Cout, 16);
            J.d1 = q.d1; J.d2 = q.d2; J.d3 = q.d3; J.d4 = q.d4;
            DLWP_REQUIRE(q.d1 && q.d2 && q.d3 && q.d4, DLWP_E_INVALID, "fold_slabs: job %d needs gw1, gb1, gw2, gb2", k);
        } else {
            J.stride = q.stride; J.d1 = q.d1; J.d2 = q.d2; J.n1 = (int)q.n1; J.n2 = (int)q.n2;
            DLWP_REQUIRE(q.d1 && q.n1 + q.n2 <= q.stride, DLWP_E_INVALID, "fold_slabs: job %d segments exceed the stride", k);
        }
        J.blk0 = blocks;
        J.nblk = (int)((J.stride + 63) / 64);
        blocks += J.nblk;
    }
    hipLaunchKernelGGL(fold_slabs_kernel, dim3(blocks), dim3(256), 0, stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

int dlwp_pwmlp_slab_reduce(const float* slab, int nslab, int Cin, int Ch, int Cout, float* gw1, float* gb1,
                           float* gw2, float* gb2, hipStream_t stream) {
    dlwp_fold_job q{};
    q.slab = slab; q.nslab = nslab; q.pwmlp = 1; q.Cin = Cin; q.Ch = Ch; q.Cout = Cout;
    q.d1 = gw1; q.d2 = gb1; q.d3 = gw2; q.d4 = gb2;
    return dlwp_fold_slabs(&q, 1, stream);
}

#ifdef DLWP_STAMPS
extern "C" int dlwp_debug_stamps_pwmlp(unsigned long long* host_out) {
    DLWP_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dlwp_stamps), sizeof(unsigned long long) * 32));
    return DLWP_OK;
}
#endif

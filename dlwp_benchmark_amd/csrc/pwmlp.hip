// Pointwise two-layer channel MLP on channels-first fields (the FNO lifting / projection
// networks: 1x1 conv -> GELU -> 1x1 conv).  Reference call sites: neuralop.FNO.lifting /
// .projection constructed at src/nsbench/models/fno/fno.py:19-27,205-215 and
// src/dlwpbench/models/fno/fno.py:38-47 (third-party arithmetic).
//
// MI355X design: one workgroup = 64 consecutive pixels of one sample x all channels.
// Both GEMMs run on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32); the hidden activations
// (Ch x 64) never leave registers: the accumulator of GEMM1 (rows = hidden channel 4g+j,
// col = pixel) is directly the B operand of GEMM2 when the A operand (W2) is read with the
// matching permuted-k order (common.cuh: mfma16_chunk).  Backward recomputes the hidden
// layer, keeps weight-gradient partials in registers over the 64 pixels and leaves them in a
// per-workgroup slab in accumulator-tile order (plain 16-byte stores; one slab-parallel launch
// folds every slab of a training step, fold_slabs_kernel below); without a slab they go out as
// hardware float atomics (API path).  The forward kernel can append the W-axis DFT of its output
// rows (lifting -> first spectral block), the backward kernel the adjoint DFT of its input
// gradient rows (projection backward -> last spectral block's backward).
#include "common.cuh"
#include "dlwpmi_internal.h"
#include "fno_rows.cuh"

namespace {

constexpr int PT = 64;        // pixels per workgroup
constexpr int FWD_WAVES = 16; // 4 waves per SIMD: the forward kernel needs 54 VGPRs and its main loop is bound by the dependent
                              // z -> GELU -> second-GEMM chain of a wave, not by the matrix pipe (8 waves: 1993, 16 waves: 2007 samples/s)
constexpr int BWD_STREAMS = 1; // hidden blocks in flight per wave of the backward kernel.  2 (with 4 waves, one per SIMD, 352 VGPRs)
                               // was measured: 1884 samples/s against 2020 for 8 waves x 1 stream (plain 4 waves: 1858)
constexpr int BWD_WAVES = BWD_STREAMS == 2 ? 4 : 8;
constexpr int LDP = PT + 4;   // LDS row stride of pixel tiles (4*LDP % 32 == 16: conflict-free B reads)

__device__ __forceinline__ const float* chan_ptr(const dlwp_chan_src& s, int b, int c) {
    if (s.tab) return s.tab[c] ? s.tab[c] + (long long)b * s.tab_bstride[c] : nullptr;
    return s.base + (long long)b * s.bstride + (long long)c * s.cstride;
}
__device__ __forceinline__ float* chan_ptr(const dlwp_chan_dst& s, int b, int c) {
    if (s.tab) return s.tab[c] ? s.tab[c] + (long long)b * s.tab_bstride[c] : nullptr;
    return s.base ? s.base + (long long)b * s.bstride + (long long)c * s.cstride : nullptr;
}


// pixel tile [C_pad][PT] of a channel view -> LDS (row stride LDP); 16-byte loads along pixels
template <typename SRC>
__device__ __forceinline__ void stage_pixels(float* dst, const SRC& s, int b, int C, int C_pad, int p0, int P,
                                             bool vec_ok) {
    if (vec_ok) {
#pragma unroll 2
        for (int u = threadIdx.x; u < C_pad * (PT / 4); u += blockDim.x) {
            const int c = u / (PT / 4), q = u % (PT / 4);
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

// NW waves: wave w owns pixel block (w & 3) and the hidden blocks hq, hq+HQ, ... (hq = w >> 2,
// HQ = NW/4); two or more waves per SIMD overlap one wave's GELU (VALU) with another's MFMAs.
template <int NOB, int NW>
__global__ __launch_bounds__(NW * 64) void pwmlp_fwd_kernel(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = NW * 64, HQ = NW / 4;
    const int LD1 = a.Cin_pad + 4, LD2 = a.Ch_pad + 4;
    float* xs = smem;
    float* w1s = xs + a.Cin_pad * LDP;
    float* w2s = w1s + a.Ch_pad * LD1;
    float* b1s = w2s + a.Cout_pad * LD2;
    float* b2s = b1s + a.Ch_pad;
    float* outs = b2s + a.Cout_pad;  // [HQ][Cout_pad][LDP] per-hidden-group partial output tiles

    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int r = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.tiles_per_sample;
    const int p0 = (blockIdx.x % a.tiles_per_sample) * PT;

    DLWP_STAMP(0);
    // issue every independent global load first, then fill the LDS images
    MatLoad<4> l2;
    MatLoad<2> l1;
    const int n2 = matload_units(a.Cout, a.Ch, a.vec_w != 0, 4), n1 = matload_units(a.Ch, a.Cin, a.vec_w != 0, 2);
    l2.issue(a.w2, n2);
    l1.issue(a.w1, n1);
    float4 ftv = make_float4(0.f, 0.f, 0.f, 0.f);          // DFT table [16][PT] of the fused rows step
    if (a.x1_out && tid < 16 * (PT / 4)) ftv = reinterpret_cast<const float4*>(a.FT)[tid];
    DLWP_STAMP(1);
    stage_pixels(xs, a.x, b, a.Cin, a.Cin_pad, p0, a.P, a.vec_x != 0);
    DLWP_STAMP(2);
    l2.commit<false>(w2s, LD2, a.Ch, a.dCh, n2);
    l1.commit<false>(w1s, LD1, a.Cin, a.dCin, n1);
    stage_matrix_tail<false>(w2s, LD2, a.w2, a.Cout, a.Ch, a.dCh, n2);
    stage_matrix_tail<false>(w1s, LD1, a.w1, a.Ch, a.Cin, a.dCin, n1);
    DLWP_STAMP(3);
    zero_padding(w1s, LD1, a.Ch, a.Cin, a.Ch_pad, a.Cin_pad);
    zero_padding(w2s, LD2, a.Cout, a.Ch, a.Cout_pad, a.Ch_pad);
    for (int idx = tid; idx < a.Ch_pad; idx += NT) b1s[idx] = idx < a.Ch ? a.b1[idx] : 0.f;
    for (int idx = tid; idx < a.Cout_pad; idx += NT) b2s[idx] = idx < a.Cout ? a.b2[idx] : 0.f;
    DLWP_STAMP(4);
    __syncthreads();
    DLWP_STAMP(5);

    const int pw0 = (w & 3) * 16, hq = w >> 2;
    const int nkc = a.Cin_pad / 16, nhb = a.Ch_pad / 16;
    f32x4 oacc[NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) oacc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int hb = hq; hb < nhb; hb += HQ) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < nkc; ++kc) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(&w1s[(hb * 16 + r) * LD1 + kc * 16 + 4 * g]);
            f32x4 b4;
#pragma unroll
            for (int s = 0; s < 4; ++s) b4[s] = xs[(kc * 16 + 4 * g + s) * LDP + pw0 + r];
            z = mfma16_chunk(a4, b4, z);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) z[j] = gelu_f(z[j] + b1s[hb * 16 + 4 * g + j]);
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(&w2s[(ob * 16 + r) * LD2 + hb * 16 + 4 * g]);
            oacc[ob] = mfma16_chunk(w4, z, oacc[ob]);
        }
    }
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            outs[(hq * a.Cout_pad + ob * 16 + 4 * g + j) * LDP + pw0 + r] = oacc[ob][j];
        }
    DLWP_STAMP(6);
    __syncthreads();
    DLWP_STAMP(7);

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
                const float bb = b2s[o];
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
            }
        }
        if (a.x1_out) {
            // W-axis pruned DFT of the finished row, so that the first spectral block needs no separate rows kernel.
            // The W2 image is dead: it hosts the table and the per-wave partial spectra.
            float* ft = w2s;                       // [16][LDP]
            float* x1s = w2s + 16 * LDP;           // [4 waves][Cout_pad][16]
            if (tid < 16 * (PT / 4)) *reinterpret_cast<float4*>(&ft[(tid / (PT / 4)) * LDP + 4 * (tid % (PT / 4))]) = ftv;
            for (int idx = a.Cout * LDP + tid; idx < a.Cout_pad * LDP; idx += NT) outs[idx] = 0.f;
            __syncthreads();
            if (w < 4) tile_rows_dft<NOB, 1>(outs, ft, x1s, LDP, 16, PT / 16, false);
            __syncthreads();
            store_x1(x1s, a.x1_out, b, (int)(blockIdx.x % a.tiles_per_sample), a.rows_H, a.m2c, a.Cout, a.Cout_pad, 16);
        }
    } else {
        for (int idx = tid; idx < a.Cout * PT; idx += NT) {
            const int o = idx / PT, p = p0 + idx % PT;
            if (p < a.P) {
                float v = b2s[o];
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
    DLWP_STAMP(8);
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
    const int LD1 = a.Cin_pad + 4, LD2 = a.Ch_pad + 4;
    float* xs = smem;                        // [Cin_pad][LDP]
    float* gys = xs + a.Cin_pad * LDP;       // [Cout_pad][LDP]
    float* w1s = gys + a.Cout_pad * LDP;     // [Ch_pad][LD1]
    float* w2s = w1s + a.Ch_pad * LD1;       // [Cout_pad][LD2]
    float* b1s = w2s + a.Cout_pad * LD2;     // [Ch_pad]

    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int r = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.tiles_per_sample;
    const int p0 = (blockIdx.x % a.tiles_per_sample) * PT;

    MatLoad<4> l2;
    MatLoad<4> l1;
    const int n2 = matload_units(a.Cout, a.Ch, a.vec_w != 0, 4), n1 = matload_units(a.Ch, a.Cin, a.vec_w != 0, 4);
    l2.issue(a.w2, n2);
    l1.issue(a.w1, n1);
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
    l2.commit<false>(w2s, LD2, a.Ch, a.dCh, n2);
    l1.commit<false>(w1s, LD1, a.Cin, a.dCin, n1);
    stage_matrix_tail<false>(w2s, LD2, a.w2, a.Cout, a.Ch, a.dCh, n2);
    stage_matrix_tail<false>(w1s, LD1, a.w1, a.Ch, a.Cin, a.dCin, n1);
    zero_padding(w1s, LD1, a.Ch, a.Cin, a.Ch_pad, a.Cin_pad);
    zero_padding(w2s, LD2, a.Cout, a.Ch, a.Cout_pad, a.Ch_pad);
    for (int idx = tid; idx < a.Ch_pad; idx += NT) b1s[idx] = idx < a.Ch ? a.b1[idx] : 0.f;
    __syncthreads();
    if (a.gres.base || a.gres.tab) {
        for (int idx = tid; idx < a.Cout * PT; idx += NT) {
            const int o = idx / PT, p = idx % PT;
            float* dst = chan_ptr(a.gres, b, o);
            if (dst && p0 + p < a.P) dst[p0 + p] += gys[o * LDP + p];
        }
    }

    DLWP_STAMP(10);
    const int nhb = a.Ch_pad / 16;

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
        f32x4 w1b[NIB];        // B operand of z^T: W1[h = r][i = 4g+s]
        float b1v;
        f32x4 pgw2[NOB], pgw1[NIB];   // running slab partials (read-modify-write accumulation across net calls)
        float pgb1;
        f32x4 gw2acc[NOB], gw1acc[NIB];
        float gb1acc;
        float *t2, *t1;
        int hb;
    };
    auto hb_begin = [&](HbState& st, int hb) {
        st.hb = hb;
#pragma unroll
        for (int kc = 0; kc < NIB; ++kc)
            st.w1b[kc] = *reinterpret_cast<const f32x4*>(&w1s[(hb * 16 + r) * LD1 + kc * 16 + 4 * g]);
        st.b1v = b1s[hb * 16 + r];
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
    auto hb_step = [&](HbState& st, int pb) {
        const int hb = st.hb;
        // z^T[p][h] = sum_i x[i][p] W1[h][i]
        f32x4 zt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < NIB; ++kc) {
            f32x4 a4;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) a4[s2] = xs[(kc * 16 + 4 * g + s2) * LDP + pb * 16 + r];
            zt = mfma16_chunk(a4, st.w1b[kc], zt);
        }
        // g_a^T[p][h] = sum_o gy[o][p] W2[o][h]
        f32x4 gat = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int oc = 0; oc < NOB; ++oc) {
            f32x4 a4, b4;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                a4[s2] = gys[(oc * 16 + 4 * g + s2) * LDP + pb * 16 + r];
                b4[s2] = w2s[(oc * 16 + 4 * g + s2) * LD2 + hb * 16 + r];
            }
            gat = mfma16_chunk(a4, b4, gat);
        }
        f32x4 actt, gzt;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float av, dv;
            gelu_both(zt[j] + st.b1v, av, dv);
            actt[j] = av;
            gzt[j] = gat[j] * dv;
            st.gb1acc += gzt[j];
        }
        // dW2[o][h] += sum_p gy[o][p] act^T[p][h]
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(&gys[(ob * 16 + r) * LDP + pb * 16 + 4 * g]);
            st.gw2acc[ob] = mfma16_chunk(a4, actt, st.gw2acc[ob]);
        }
        // dW1[h][i] += sum_p gz[h][p] x[i][p]   (gz^T registers read as an A operand are gz)
#pragma unroll
        for (int ib = 0; ib < NIB; ++ib) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(&xs[(ib * 16 + r) * LDP + pb * 16 + 4 * g]);
            st.gw1acc[ib] = mfma16_chunk(gzt, b4, st.gw1acc[ib]);
        }
        // gz[h][p] in accumulator layout = gz^T (as A operand) x identity
        const f32x4 gz = mfma16_chunk(gzt, ident, f32x4{0.f, 0.f, 0.f, 0.f});
        // dX[i][p] += sum_h W1[h][i] gz[h][p]
#pragma unroll
        for (int ib = 0; ib < NIB; ++ib) {
            f32x4 a4;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) a4[s2] = w1s[(hb * 16 + 4 * g + s2) * LD1 + ib * 16 + r];
            gxacc[pb][ib] = mfma16_chunk(a4, gz, gxacc[pb][ib]);
        }
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
    if constexpr (BWD_STREAMS == 2) {
        // one wave per SIMD, two hidden blocks in flight per wave: the pixel blocks of both are interleaved in program order,
        // so that one block's GELU / dependent MFMA chains fill the other's gaps (no reliance on SIMD arbitration)
        for (int hb = w; hb < nhb; hb += 2 * NW) {
            HbState sa, sb;
            const bool two = hb + NW < nhb;
            hb_begin(sa, hb);
            if (two) hb_begin(sb, hb + NW);
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                hb_step(sa, pb);
                if (two) hb_step(sb, pb);
            }
            hb_flush(sa);
            if (two) hb_flush(sb);
        }
    } else {
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
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) hb_step(st, pb);
            if (hb == w) DLWP_STAMP(18);
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
    float* red = w1s;  // [NW][Cin_pad][LDP]
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

int dlwp_pwmlp_fwd_rows_ex(const dlwp_chan_src* x, const float* w1, const float* b1, const float* w2,
                           const float* b2, const dlwp_chan_dst* y, const dlwp_chan_src* res, int B,
                           int Cin, int Ch, int Cout, int P, const dlwp_fno_plan* rows_plan, float2* x1_out,
                           hipStream_t stream) {
    DLWP_REQUIRE(B > 0 && Cin > 0 && Ch > 0 && Cout > 0 && P > 0, DLWP_E_INVALID,
                 "pwmlp_fwd: non-positive dimension");
    FwdArgs a{};
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
    bool rows_after = false;
    if (x1_out) {
        DLWP_REQUIRE(rows_plan && y->base, DLWP_E_INVALID, "pwmlp_fwd: the rows DFT needs a plan and a dense output");
        const bool fuse = dlwp_pwmlp_rows_fusable(rows_plan, Cout, P) && a.vec_x &&
                          (size_t)a.Cout_pad * (a.Ch_pad + 4) >= (size_t)16 * LDP + (size_t)4 * a.Cout_pad * 16;
        if (fuse) { a.x1_out = x1_out; a.FT = rows_plan->FT_fwd; a.rows_H = rows_plan->H; a.m2c = rows_plan->m2c; }
        else rows_after = true;       // shapes the epilogue cannot host: separate rows kernel, same result
    }
    const size_t lds = sizeof(float) * ((size_t)a.Cin_pad * LDP + (size_t)a.Ch_pad * (a.Cin_pad + 4) +
                                        (size_t)a.Cout_pad * (a.Ch_pad + 4) + a.Ch_pad + a.Cout_pad +
                                        (size_t)(FWD_WAVES / 4) * a.Cout_pad * LDP);
    constexpr int NW = FWD_WAVES;
    const dim3 grid(B * a.tiles_per_sample), block(NW * 64);
    int rc;
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
    a.vec_w = aligned16(w1) && aligned16(w2);
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
    size_t wimg = (size_t)a.Ch_pad * (a.Cin_pad + 4) + (size_t)a.Cout_pad * (a.Ch_pad + 4) + a.Ch_pad;
    const size_t red = (size_t)BWD_WAVES * a.Cin_pad * LDP;  // gx partial tiles alias the weight images
    if (wimg < red) wimg = red;
    const size_t lds = sizeof(float) * ((size_t)a.Cin_pad * LDP + (size_t)a.Cout_pad * LDP + wimg);
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

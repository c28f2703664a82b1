// FNO block = SpectralConv (mode-truncated rfft2 -> per-mode complex channel mixing -> irfft2)
// + linear (1x1) skip + bias, forward and backward.  Arithmetic restated from neuralop
// SpectralConv/FNOBlocks (third-party); reference call sites
// `self.fno(x_t)` src/nsbench/models/fno/fno.py:38,96,246, src/dlwpbench/models/fno/fno.py:103.
//
// MI355X design.  Only m1 x m2c modes of the H x (W/2+1) spectrum are ever used (12 x 7 of
// 64 x 33 at the headline config), so no FFT is run: the transform is a pruned DFT factored
// into a W-axis step (a [C x W] x [W x 2*m2c] GEMM per image row) and an H-axis step
// (m1 complex dot products of length H per (channel, column)).  Three kernels per block:
//   rows   : one workgroup per image row (b,h), all channels: W-axis DFT on MFMA -> x1[b][h][kx][c]
//   mix    : one workgroup per kept mode (j,kx): H-axis step, complex channel contraction with
//            the mode's [C x C] weight slice (mode-major layout => contiguous 8*C*C bytes)
//   spatial: one workgroup per image row: inverse H-step (tiny), then ONE concatenated-K MFMA
//            GEMM  pre[o][w] = [Wskip | Y1(o,:)] . [x ; G]  (skip conv + inverse W-axis DFT),
//            bias, and optionally the next block's `rows` stage fused on the result.
// The backward pass reuses the same three kernels with adjoint tables.
#include "common.hip.h"
#include "dlwpmi_internal.h"
#include "fno_rows.hip.h"
#include <cmath>
#include <vector>

namespace {

struct SpatialDev {
    const float* tin; const float2* spec; const float* wskip; const float* bias;
    const float* pprev; float* out; float2* x1_out; float* g_wskip; float* g_bias;
    float* gslab; int gslab_accumulate;   // per-workgroup {g_wskip[C*C], g_bias[C]} partials (stride C*C+C padded to 4)
    const float2* twH; const float* G; const float* FT;
    int act_tin, transpose_w, act_prev, x1_act, is_bwd, vec_w;
    int C, H, W, m1, m2c, C_pad, NP;
    FastDiv dC;
};

struct RowsDev {
    const float* x; float2* x1; const float* FT;
    int act, C, H, W, m2c, C_pad, NP;
};

template <int NCB, int NBN>
__global__ __launch_bounds__(256) void fno_rows_kernel(RowsDev a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LDP = a.W + 4;
    float* tile = smem;                    // [C_pad][LDP]
    float* ft = tile + a.C_pad * LDP;      // [NP][LDP]
    float* x1s = ft + a.NP * LDP;          // [4 waves][C_pad][NP]
    const int tid = threadIdx.x;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int W4 = a.W / 4;
    for (int idx = tid; idx < a.C_pad * W4; idx += 256) {
        const int c = idx / W4, x4 = idx % W4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < a.C) v = *reinterpret_cast<const float4*>(&a.x[(((long long)b * a.C + c) * a.H + h) * a.W + 4 * x4]);
        *reinterpret_cast<float4*>(&tile[c * LDP + 4 * x4]) = v;
    }
    for (int idx = tid; idx < a.NP * W4; idx += 256) {
        const int n = idx / W4, x4 = idx % W4;
        *reinterpret_cast<float4*>(&ft[n * LDP + 4 * x4]) = *reinterpret_cast<const float4*>(&a.FT[n * a.W + 4 * x4]);
    }
    __syncthreads();
    tile_rows_dft<NCB, NBN>(tile, ft, x1s, LDP, a.NP, a.W / 16, a.act != 0);
    __syncthreads();
    store_x1(x1s, a.x1, b, h, a.H, a.m2c, a.C, a.C_pad, a.NP);
}

// MODE 0: forward (+bias); 1: backward with GELU' of the block input; 2: backward, input was not activated
template <int NCB, int NBN, int MODE>
__global__ __launch_bounds__(256) void fno_spatial_kernel(SpatialDev a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LDP = a.W + 4, LDK = a.C_pad + 4, LDS1 = a.NP + 4;
    float* tin_s = smem;                          // [C_pad][LDP]
    float* tout_s = tin_s + a.C_pad * LDP;        // [C_pad][LDP]
    float* pprev_s = tout_s + a.C_pad * LDP;      // [C_pad][LDP]    (bwd)
    float* ft = pprev_s + a.C_pad * LDP;          // [NP][LDP]
    float* x1s = ft + a.NP * LDP;                 // [4 waves][C_pad][NP]
    float* bias_s = x1s + 4 * a.C_pad * a.NP;     // [C_pad]
    float* gs = bias_s + a.C_pad;                 // [NP][LDP]       dead after the main GEMM
    float* ks = gs + a.NP * LDP;                  // [C_pad][LDK]    dead after the main GEMM
    float* s1 = ks + a.C_pad * LDK;               // [C_pad][LDS1]   dead after the main GEMM
    float* gks = gs;                              // [4 waves][C_pad][C_pad] (bwd) aliases gs/ks/s1

    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int r = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int W4 = a.W / 4, nwb = a.W / 16;
    const bool need_prev = a.is_bwd && (a.act_prev || a.g_wskip || a.gslab);
    DLWP_SPAN_BEGIN();
    DLWP_STAMP(0);
    // ---- issue phase: the first chunk of every independent input goes into registers before any LDS
    // write, so the workgroup pays one global latency for all of them (larger shapes loop afterwards)
    const int tile_units = a.C_pad * W4, tab_units = a.NP * W4;
    float4 tv[2], pv[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = tid + 256 * k, c = u / W4, x4 = u - c * W4;
        tv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        pv[k] = tv[k];
        if (u < tile_units && c < a.C) {
            const long long off = (((long long)b * a.C + c) * a.H + h) * a.W + 4 * x4;
            tv[k] = *reinterpret_cast<const float4*>(&a.tin[off]);
            if (need_prev) pv[k] = *reinterpret_cast<const float4*>(&a.pprev[off]);
        }
    }
    float4 gv = make_float4(0.f, 0.f, 0.f, 0.f), fv = gv;
    if (tid < tab_units) {
        gv = reinterpret_cast<const float4*>(a.G)[tid];
        if (a.x1_out) fv = reinterpret_cast<const float4*>(a.FT)[tid];
    }
    MatLoad<1> lk;
    const int nk = matload_units(a.C, a.C, a.vec_w != 0, 1);
    lk.issue(a.wskip, nk);
    // inverse H-axis step operands for this row: spec[b][j][kx][o], j < min(m1, 16)
    constexpr int MJ = 16;
    float2 sv[MJ];
    const int d_kx = tid / a.C_pad, d_o = tid - d_kx * a.C_pad;
    const bool d_valid = tid < a.C_pad * (a.NP / 2) && d_kx < a.m2c && d_o < a.C;
    // unconditional loads from clamped (always valid) addresses: a select right behind each load would
    // make the compiler wait for every load separately; invalid lanes/steps are ignored at use
    const float2* sp = a.spec + (((long long)b * a.m1) * a.m2c + (d_valid ? d_kx : 0)) * a.C + (d_valid ? d_o : 0);
    const long long jstride = (long long)a.m2c * a.C;
#pragma unroll
    for (int j = 0; j < MJ; ++j) sv[j] = sp[(j < a.m1 ? j : a.m1 - 1) * jstride];
    // twiddles of this row: lane j (mod 16) holds tw[j]; a uniform index would become a chain of scalar loads
    // with a wait each.  Broadcast with shuffles at use.
    const int twj = (lane & 15) < a.m1 ? (lane & 15) : a.m1 - 1;
    const float2 twl = a.twH[twj * a.H + h];
    const float bias_raw = a.bias ? a.bias[tid < a.C ? tid : 0] : 0.f;
    const float bias_v = tid < a.C ? bias_raw : 0.f;
    // running partials of this workgroup's gradient slab (read-modify-write across net calls): fetched here, with every
    // other input, instead of in front of the final store (a dependent global round trip at the very end of the kernel)
    constexpr int SLQ = 4;
    float slab_old[SLQ], slab_b_old = 0.f;
#pragma unroll
    for (int k = 0; k < SLQ; ++k) slab_old[k] = 0.f;
    const long long slab_off = (long long)blockIdx.x * (((long long)a.C * a.C + a.C + 3) & ~3LL);
    if (a.is_bwd && a.gslab && a.gslab_accumulate) {
#pragma unroll
        for (int k = 0; k < SLQ; ++k) slab_old[k] = a.gslab[slab_off + min(tid + 256 * k, a.C * a.C - 1)];
        slab_b_old = a.gslab[slab_off + a.C * a.C + min(tid, a.C - 1)];
    }
    DLWP_STAMP(1);

    // ---- commit phase
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = tid + 256 * k, c = u / W4, x4 = u - c * W4;
        if (u < tile_units) {
            float4 v = tv[k];
            if (a.act_tin) { const f32x4 ga = gelu4(f32x4{v.x, v.y, v.z, v.w}); v = make_float4(ga[0], ga[1], ga[2], ga[3]); }
            *reinterpret_cast<float4*>(&tin_s[c * LDP + 4 * x4]) = v;
            if (need_prev) *reinterpret_cast<float4*>(&pprev_s[c * LDP + 4 * x4]) = pv[k];
        }
    }
    for (int idx = tid + 512; idx < tile_units; idx += 256) {
        const int c = idx / W4, x4 = idx % W4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f), pvv = v;
        if (c < a.C) {
            const long long off = (((long long)b * a.C + c) * a.H + h) * a.W + 4 * x4;
            v = *reinterpret_cast<const float4*>(&a.tin[off]);
            if (a.act_tin) { const f32x4 ga = gelu4(f32x4{v.x, v.y, v.z, v.w}); v = make_float4(ga[0], ga[1], ga[2], ga[3]); }
            if (need_prev) pvv = *reinterpret_cast<const float4*>(&a.pprev[off]);
        }
        *reinterpret_cast<float4*>(&tin_s[c * LDP + 4 * x4]) = v;
        if (need_prev) *reinterpret_cast<float4*>(&pprev_s[c * LDP + 4 * x4]) = pvv;
    }
    DLWP_STAMP(2);
    if (tid < tab_units) {
        const int n = tid / W4, x4 = tid - n * W4;
        *reinterpret_cast<float4*>(&gs[n * LDP + 4 * x4]) = gv;
        if (a.x1_out) *reinterpret_cast<float4*>(&ft[n * LDP + 4 * x4]) = fv;
    }
    for (int idx = tid + 256; idx < tab_units; idx += 256) {
        const int n = idx / W4, x4 = idx % W4;
        *reinterpret_cast<float4*>(&gs[n * LDP + 4 * x4]) = *reinterpret_cast<const float4*>(&a.G[n * a.W + 4 * x4]);
        if (a.x1_out)
            *reinterpret_cast<float4*>(&ft[n * LDP + 4 * x4]) = *reinterpret_cast<const float4*>(&a.FT[n * a.W + 4 * x4]);
    }
    if (a.transpose_w) {
        lk.commit<true>(ks, LDK, a.C, a.dC, nk);
        stage_matrix_tail<true>(ks, LDK, a.wskip, a.C, a.C, a.dC, nk);
    } else {
        lk.commit<false>(ks, LDK, a.C, a.dC, nk);
        stage_matrix_tail<false>(ks, LDK, a.wskip, a.C, a.C, a.dC, nk);
    }
    zero_padding(ks, LDK, a.C, a.C, a.C_pad, a.C_pad);
    if (tid < a.C_pad) bias_s[tid] = bias_v;
    for (int idx = tid + 256; idx < a.C_pad; idx += 256) bias_s[idx] = (a.bias && idx < a.C) ? a.bias[idx] : 0.f;
    DLWP_STAMP(3);
    // inverse H-axis step for this row: s1[o][2kx(+1)] = sum_j spec[b][j][kx][o] * conj(twH[j][h])
    float twx[MJ], twy[MJ];   // broadcast the lane-held twiddles (every lane is active here)
#pragma unroll
    for (int j = 0; j < MJ; ++j) { twx[j] = __shfl(twl.x, j, 16); twy[j] = __shfl(twl.y, j, 16); }
    for (int idx = tid; idx < a.C_pad * (a.NP / 2); idx += 256) {
        const int kx = idx / a.C_pad, o = idx - kx * a.C_pad;
        float re = 0.f, im = 0.f;
        if (idx == tid) {   // first chunk: operands already in registers
#pragma unroll
            for (int j = 0; j < MJ; ++j) {
                if (j < a.m1) {
                    re += sv[j].x * twx[j] + sv[j].y * twy[j];   // v * conj(t)
                    im += sv[j].y * twx[j] - sv[j].x * twy[j];
                }
            }
        }
        if (kx < a.m2c && o < a.C) {
            const float2* sp2 = a.spec + (((long long)b * a.m1) * a.m2c + kx) * a.C + o;
            for (int j = (idx == tid ? MJ : 0); j < a.m1; ++j) {
                const float2 v = sp2[j * jstride];
                const float2 t = a.twH[j * a.H + h];
                re += v.x * t.x + v.y * t.y;
                im += v.y * t.x - v.x * t.y;
            }
        }
        const bool ok = kx < a.m2c && o < a.C;
        s1[o * LDS1 + 2 * kx] = ok ? re : 0.f;
        s1[o * LDS1 + 2 * kx + 1] = ok ? im : 0.f;
    }
    DLWP_STAMP(4);
    __syncthreads();
    DLWP_STAMP(5);

    // main concatenated-K GEMM: acc[o][w] = sum_i ks[o][i] tin[i][w] + sum_n s1[o][n] gs[n][w]
    for (int wb = w; wb < nwb; wb += 4) {
        f32x4 acc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < NCB; ++kc) {
            f32x4 b4;
#pragma unroll
            for (int s = 0; s < 4; ++s) b4[s] = tin_s[(kc * 16 + 4 * g + s) * LDP + wb * 16 + r];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(&ks[(cb * 16 + r) * LDK + kc * 16 + 4 * g]);
                acc[cb] = mfma16_chunk(a4, b4, acc[cb]);
            }
        }
#pragma unroll
        for (int nc = 0; nc < NBN; ++nc) {
            f32x4 b4;
#pragma unroll
            for (int s = 0; s < 4; ++s) b4[s] = gs[(nc * 16 + 4 * g + s) * LDP + wb * 16 + r];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(&s1[(cb * 16 + r) * LDS1 + nc * 16 + 4 * g]);
                acc[cb] = mfma16_chunk(a4, b4, acc[cb]);
            }
        }
        // epilogue: operands for all elements are fetched first, then applied (no per-element waits / branches)
        float ep[NCB][4];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = cb * 16 + 4 * g + j, x = wb * 16 + r;
                ep[cb][j] = MODE == 0 ? bias_s[o] : (MODE == 1 ? pprev_s[o * LDP + x] : 0.f);
            }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            f32x4 v4 = acc[cb];
            const f32x4 e4 = f32x4{ep[cb][0], ep[cb][1], ep[cb][2], ep[cb][3]};
            if (MODE == 0) v4 += e4;
            else if (MODE == 1) v4 *= gelu_grad4(e4);              // packed fp32 polynomial
#pragma unroll
            for (int j = 0; j < 4; ++j) tout_s[(cb * 16 + 4 * g + j) * LDP + wb * 16 + r] = v4[j];
        }
    }
    DLWP_STAMP(6);
    __syncthreads();
    DLWP_STAMP(7);

    for (int idx = tid; idx < a.C * W4; idx += 256) {
        const int c = idx / W4, x4 = idx % W4;
        *reinterpret_cast<float4*>(&a.out[(((long long)b * a.C + c) * a.H + h) * a.W + 4 * x4]) =
            *reinterpret_cast<const float4*>(&tout_s[c * LDP + 4 * x4]);
    }
    DLWP_STAMP(8);
    if (a.x1_out) tile_rows_dft<NCB, NBN>(tout_s, ft, x1s, LDP, a.NP, nwb, a.x1_act != 0);
    DLWP_STAMP(9);

    if (a.is_bwd && (a.g_wskip || a.gslab)) {
        // gK[o][i] += sum_w g_pre[o][w] * act(x)[i][w]
        f32x4 kacc[NCB][NCB];
#pragma unroll
        for (int ob = 0; ob < NCB; ++ob)
#pragma unroll
            for (int ib = 0; ib < NCB; ++ib) kacc[ob][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kc = w; kc < nwb; kc += 4) {
#pragma unroll
            for (int ib = 0; ib < NCB; ++ib) {
                f32x4 b4 = *reinterpret_cast<const f32x4*>(&pprev_s[(ib * 16 + r) * LDP + kc * 16 + 4 * g]);
                if (a.act_prev) {
                    b4 = gelu4(b4);
                }
#pragma unroll
                for (int ob = 0; ob < NCB; ++ob) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(&tin_s[(ob * 16 + r) * LDP + kc * 16 + 4 * g]);
                    kacc[ob][ib] = mfma16_chunk(a4, b4, kacc[ob][ib]);
                }
            }
        }
#pragma unroll
        for (int ob = 0; ob < NCB; ++ob)
#pragma unroll
            for (int ib = 0; ib < NCB; ++ib)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    gks[((w * NCB + ob) * 16 + 4 * g + j) * a.C_pad + ib * 16 + r] = kacc[ob][ib][j];
    }
    DLWP_STAMP(10);
    __syncthreads();
    DLWP_STAMP(11);
    if (a.x1_out) store_x1(x1s, a.x1_out, b, h, a.H, a.m2c, a.C, a.C_pad, a.NP);
    if (a.is_bwd && (a.g_wskip || a.gslab)) {
        int kq = 0;
        for (int idx = tid; idx < a.C * a.C; idx += 256, ++kq) {
            const int o = fastdiv(idx, a.dC), i = idx - o * a.C;
            float v = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) v += gks[(w2 * a.C_pad + o) * a.C_pad + i];
            if (a.gslab) {
                float* sl = a.gslab + slab_off + idx;
                float old = 0.f;
                if (a.gslab_accumulate) {
                    old = kq == 0 ? slab_old[0] : kq == 1 ? slab_old[1] : kq == 2 ? slab_old[2] : kq == 3 ? slab_old[3] : *sl;
                }
                *sl = old + v;
            } else {
                atomic_add_f32(&a.g_wskip[idx], v);   // all workgroups hit the same C*C words: slow, API path only
            }
        }
    }
    if (a.is_bwd && (a.g_bias || a.gslab) && tid < a.C) {
        float s = 0.f;
        for (int x = 0; x < a.W; ++x) s += tin_s[tid * LDP + x];
        if (a.gslab) {
            a.gslab[slab_off + a.C * a.C + tid] = (a.gslab_accumulate ? slab_b_old : 0.f) + s;
        } else {
            atomic_add_f32(&a.g_bias[tid], s);
        }
    }
    DLWP_STAMP(12);
    DLWP_SPAN_END();
}

// ------------------------------------------------------------------------------------------
// Per-mode stage.  One 512-thread workgroup per kept mode (j,kx).  Everything it reads is issued
// as batched independent loads up front (the mode's [C][C] weight slice as 16-byte loads straight
// into an LDS image with a +1 complex row pad, the twiddle row, then the x1 column in unrolled
// groups of 8), because at these sizes the kernel is pure load latency.
constexpr int MIXT = 512;

struct MixDev {
    const float2* x1; const float2* wspec; const float2* xhat_in; float2* xhat; float2* y;
    float2* g_wspec; const float2* twH;
    int B, C, H, m1, m2c;
    FastDiv dC, dBC, dBC2;   // exact division by C, B*C, B*C/2 without the ~40-instruction integer divide
};

// H-axis step, fast path (C even): every thread owns TWO adjacent channels (one 16-byte load per image row) and
// the rows h = hs, hs+HS, ...; loads are issued in batches of 8 BEFORE the twiddles are needed, so the weight
// slice, the twiddle row and the first x1 batch share one global-memory latency.
struct HstepLoads {
    float4 v[8];
    int bc2, hs, b, c, nh;
};
__device__ __forceinline__ void mix_hstep_issue(const MixDev& a, int kx, int HS, HstepLoads& L, int h0) {
    const int BC2 = a.B * a.C / 2;
    const int u = threadIdx.x;
    L.hs = fastdiv(u, a.dBC2); L.bc2 = u - L.hs * BC2;
    L.b = fastdiv(2 * L.bc2, a.dC); L.c = 2 * L.bc2 - L.b * a.C;
    const bool act = u < BC2 * HS;
    const float2* src = a.x1 + (((long long)L.b * a.H) * a.m2c + kx) * a.C + L.c;
    const long long hstride = (long long)a.m2c * a.C;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int h = h0 + L.hs + q * HS;
        const int hc = (act && h < a.H) ? h : 0;                       // clamped: always a valid address
        L.v[q] = *reinterpret_cast<const float4*>(src + hc * hstride);
    }
}
__device__ __forceinline__ void mix_hstep_accum(const MixDev& a, int HS, const HstepLoads& L, const float2* tws, int h0,
                                                float2& s0, float2& s1) {
    float2 tq[8];   // all LDS reads first: at this occupancy a read-then-use loop pays the LDS latency per iteration
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int h = h0 + L.hs + q * HS;
        tq[q] = tws[h < a.H ? h : 0];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int h = h0 + L.hs + q * HS;
        if (h < a.H) {
            const float2 t = tq[q];
            s0.x += L.v[q].x * t.x - L.v[q].y * t.y;
            s0.y += L.v[q].x * t.y + L.v[q].y * t.x;
            s1.x += L.v[q].z * t.x - L.v[q].w * t.y;
            s1.y += L.v[q].z * t.y + L.v[q].w * t.x;
        }
    }
}
// whole fast H-step after the first batch has been issued; returns the number of partial slots
__device__ __forceinline__ int mix_hstep_fast(const MixDev& a, int kx, int HS, HstepLoads& L, const float2* tws, float2* part) {
    const int BC = a.B * a.C, BC2 = BC / 2;
    float2 s0 = make_float2(0.f, 0.f), s1 = s0;
    mix_hstep_accum(a, HS, L, tws, 0, s0, s1);
    for (int h0 = 8 * HS; h0 < a.H; h0 += 8 * HS) {
        mix_hstep_issue(a, kx, HS, L, h0);
        mix_hstep_accum(a, HS, L, tws, h0, s0, s1);
    }
    if ((int)threadIdx.x < BC2 * HS) {
        part[L.hs * BC + 2 * L.bc2] = s0;
        part[L.hs * BC + 2 * L.bc2 + 1] = s1;
    }
    return HS;
}

// H-axis step into LDS partials (generic path): part[hs][b*C+c] = sum_{h = hs mod HS} x1[b][h][kx][c] * tw[h]; returns HS
__device__ __forceinline__ int mix_hstep(const MixDev& a, int kx, const float2* tws, float2* part, int HS) {
    const int BC = a.B * a.C;
    for (int u = threadIdx.x; u < BC * HS; u += MIXT) {
        const int hs = fastdiv(u, a.dBC), bc = u - hs * BC, b = fastdiv(bc, a.dC), c = bc - b * a.C;
        const float2* src = a.x1 + (((long long)b * a.H) * a.m2c + kx) * a.C + c;
        const long long hstride = (long long)a.m2c * a.C;
        float re = 0.f, im = 0.f;
#pragma unroll 8
        for (int h = hs; h < a.H; h += HS) {
            const float2 v = src[h * hstride];
            const float2 t = tws[h];
            re += v.x * t.x - v.y * t.y;
            im += v.x * t.y + v.y * t.x;
        }
        part[hs * BC + bc] = make_float2(re, im);
    }
    return HS;
}

// out[i] = sum_s part[s][i]
__device__ __forceinline__ void mix_fold(float2* out, const float2* part, int n, int ns) {
    for (int i = threadIdx.x; i < n; i += MIXT) {
        float2 v = make_float2(0.f, 0.f);
        for (int s0 = 0; s0 < ns; s0 += 8) {
            float2 p[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) p[q] = part[(s0 + q < ns ? s0 + q : 0) * n + i];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (s0 + q < ns) { v.x += p[q].x; v.y += p[q].y; }
        }
        out[i] = v;
    }
}

// partial slots of the fast H-step: threads / (B*C/2), clamped to [1, H]
__device__ __host__ __forceinline__ int mix_slots2(int B, int C, int H) {
    const int bc2 = B * C / 2;
    int s = bc2 > 0 ? MIXT / bc2 : 1;
    if (s < 1) s = 1;
    if (s > H) s = H;
    return s;
}
// number of partial slots a phase splits its reduction over (threads / outputs, clamped)
__device__ __host__ __forceinline__ int mix_split(int outputs, int limit) {
    int s = MIXT / outputs;
    if (s < 1) s = 1;
    if (s > limit) s = limit;
    return s;
}

// ---- the per-mode complex channel contraction on the matrix cores (exact-f32 v_mfma_f32_16x16x4_f32).
// A complex product y[b][o] = sum_i x[b][i] w[i][o] is ONE real GEMM with the real and imaginary planes of the SAMPLE side
// stacked along M and the (re, im) pair of the weight interleaved along K:
//     rows (b, re) = [ xr_i, -xi_i ]_i      rows (b, im) = [ xi_i, xr_i ]_i      B[k = 2i + ri][n = o] = w[i][o][ri]
// => 8 B C^2 flops, every weight element used exactly once, M = 2 B rows (8 of the 16 tile rows at the headline batch of 4;
// full at B >= 8).  The same scheme gives the backward products: gx = ghat conj(w)^T (B[k = 2o + ri][n = i] = w[i][o][ri])
// and the weight gradient gw[i][o] += sum_b conj(xhat[b][i]) ghat[b][o] (K = 2 B: one 16x16x4 instruction per pair of samples).
// Weight operand: narrow layers (C <= 64) stage the mode's [C][C] slice in LDS with ONE 16-byte load per thread, in flight
// together with the x1 loads of the H-axis step (rows padded by one complex number: conflict-free 8-byte fragment reads along
// rows and columns); wide layers read every fragment once, straight from global memory / L2 into registers (two 8-byte loads
// per 16-deep K chunk, 128 contiguous bytes per 16 lanes) -- a 217-channel slice is 377 KB and is used once per workgroup.

// weight slice [C][C] complex -> LDS rows of (C+1) complex
__device__ __forceinline__ void mix_stage_w(const float2* wm, float2* ws, int C, FastDiv dC) {
    const int n2 = C * C / 2;  // pairs of complex numbers (C*C is even whenever C is even)
    if ((C & 1) == 0) {
#pragma unroll 4
        for (int u = threadIdx.x; u < n2; u += MIXT) {
            const float4 v = reinterpret_cast<const float4*>(wm)[u];
            const int e = 2 * u, i = fastdiv(e, dC), o = e - i * C;
            ws[i * (C + 1) + o] = make_float2(v.x, v.y);
            ws[i * (C + 1) + o + 1] = make_float2(v.z, v.w);
        }
    } else {
#pragma unroll 4
        for (int e = threadIdx.x; e < C * C; e += MIXT) ws[(e / C) * (C + 1) + e % C] = wm[e];
    }
}

// A fragment of chunk kc for tile row m = 16 mt + r: element s <-> k = 16 kc + 4 g + s, complex index i = 8 kc + 2 g + s / 2.
// CONJ = 0: rows (b, re) = (x, -y), (b, im) = (y, x)      [x * w]
// CONJ = 1: rows (b, re) = (x, y),  (b, im) = (y, -x)     [x * conj(w)]
template <int CONJ>
__device__ __forceinline__ f32x4 mix_afrag(const float2* xh, int B, int C, int mt, int kc, int r, int g) {
    const int m = 16 * mt + r, b = m >> 1, part = m & 1, i0 = 8 * kc + 2 * g;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (b < B) {
        const float2 v0 = i0 < C ? xh[b * C + i0] : make_float2(0.f, 0.f);
        const float2 v1 = i0 + 1 < C ? xh[b * C + i0 + 1] : make_float2(0.f, 0.f);
        if (CONJ == 0) a = part ? f32x4{v0.y, v0.x, v1.y, v1.x} : f32x4{v0.x, -v0.y, v1.x, -v1.y};
        else a = part ? f32x4{v0.y, -v0.x, v1.y, -v1.x} : f32x4{v0.x, v0.y, v1.x, v1.y};
    }
    return a;
}
// B fragment of chunk kc for tile column n = 16 nt + r from a [C][pitch] complex image (LDS: pitch C + 1; global: pitch C).
// TRW = 0: w[i = 8 kc + 2 g (+1)][o = n]; TRW = 1: w[i = n][o = 8 kc + 2 g (+1)].  Loads are unconditional from clamped
// addresses (selects afterwards), so that a group of them is in flight together.
template <int TRW>
__device__ __forceinline__ f32x4 mix_bfrag(const float2* wm, int pitch, int C, int nt, int kc, int r, int g) {
    const int n = 16 * nt + r, k0 = 8 * kc + 2 * g;
    const int nc = n < C ? n : C - 1, k0c = k0 < C ? k0 : C - 1, k1c = k0 + 1 < C ? k0 + 1 : C - 1;
    const float2 w0 = TRW ? wm[(long long)nc * pitch + k0c] : wm[(long long)k0c * pitch + nc];
    const float2 w1 = TRW ? wm[(long long)nc * pitch + k1c] : wm[(long long)k1c * pitch + nc];
    const bool ok0 = n < C && k0 < C, ok1 = n < C && k0 + 1 < C;
    return f32x4{ok0 ? w0.x : 0.f, ok0 ? w0.y : 0.f, ok1 ? w1.x : 0.f, ok1 ? w1.y : 0.f};
}

// store one accumulator tile: rows 4g .. 4g+3 = samples 2g, 2g+1 (+ 8 mt), (re, im) each; column n = 16 nt + r
__device__ __forceinline__ void mix_store_tile(float2* out, long long ostride_b, const f32x4 acc, int B, int C, int nt, int mt,
                                               int r, int g) {
    const int n = 16 * nt + r, b0 = 8 * mt + 2 * g;
    if (n < C) {
        if (b0 < B) out[(long long)b0 * ostride_b + n] = make_float2(acc[0], acc[1]);
        if (b0 + 1 < B) out[(long long)(b0 + 1) * ostride_b + n] = make_float2(acc[2], acc[3]);
    }
}

// out[b][n] (complex, global, index base + n) = sum over the K chunks of A(xh) . B(w); work items = (column tile, pair of row
// tiles), dealt to the waves round-robin
template <int CONJ, int TRW>
__device__ __forceinline__ void mix_contract(const float2* xh, const float2* wm, int pitch, float2* out, long long ostride_b,
                                             int B, int C, int nt_begin = 0, int nt_end = 1 << 30) {
    const int lane = lane_id(), w = wave_id(), r = lane & 15, g = lane >> 4;
    constexpr int NW = MIXT / 64;
    const int nkc = (2 * C + 15) / 16, nmt = (2 * B + 15) / 16, nmp = (nmt + 1) / 2;
    const int ntn = min((C + 15) / 16, nt_end) - nt_begin;        // column tiles [nt_begin, nt_end) of this workgroup
    for (int item = w; item < ntn * nmp; item += NW) {
        const int nt = nt_begin + item % ntn, mt0 = 2 * (item / ntn);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;     // two row tiles share every weight fragment (B > 8)
        for (int kc0 = 0; kc0 < nkc; kc0 += 4) {
            f32x4 bf[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bf[q] = mix_bfrag<TRW>(wm, pitch, C, nt, min(kc0 + q, nkc - 1), r, g);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (kc0 + q < nkc) {
                    acc0 = mfma16_chunk(mix_afrag<CONJ>(xh, B, C, mt0, kc0 + q, r, g), bf[q], acc0);
                    if (mt0 + 1 < nmt) acc1 = mfma16_chunk(mix_afrag<CONJ>(xh, B, C, mt0 + 1, kc0 + q, r, g), bf[q], acc1);
                }
        }
        mix_store_tile(out, ostride_b, acc0, B, C, nt, mt0, r, g);
        if (mt0 + 1 < nmt) mix_store_tile(out, ostride_b, acc1, B, C, nt, mt0 + 1, r, g);
    }
}

// LDSW: the mode's weight slice is staged in LDS (narrow layers)
template <bool LDSW>
__global__ __launch_bounds__(MIXT) void fno_mix_fwd_kernel(MixDev a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int BC = a.B * a.C, C = a.C;
    const int NS = mix_split(BC, a.H < C ? a.H : C);   // partial slots of the generic H-step
    const int NS2 = mix_slots2(a.B, C, a.H);           // partial slots of the two-channels-per-thread H-step
    float2* xh = reinterpret_cast<float2*>(smem);  // [BC]
    float2* part = xh + BC;                        // [max(NS, NS2)][BC]
    float2* tws = part + (NS > NS2 ? NS : NS2) * BC;   // [H]
    float2* ws = tws + a.H;                        // [C][C+1]   (LDSW)
    // XCD-aware mode order: workgroups go round-robin to the 8 XCDs; give every XCD a contiguous range of (kx, j) so that
    // the m1 row frequencies of one column kx -- which read the same x1 slice -- meet in one L2
    int mode = blockIdx.x;
    {
        const int nm = gridDim.x, full = (nm / 8) * 8;
        if (mode < full) mode = (mode % 8) * (nm / 8) + mode / 8;
    }
    const int kx = mode / a.m1, j = mode - kx * a.m1;
    const bool fast = (C % 2 == 0) && (BC / 2) * NS2 <= MIXT && NS2 >= 1;
    const float2* wm = a.wspec + ((long long)(j * a.m2c + kx) * C) * C;
    DLWP_STAMP(16);
    HstepLoads hl;
    if (fast) mix_hstep_issue(a, kx, NS2, hl, 0);            // x1 loads in flight before anything is waited for
    if (LDSW) mix_stage_w(wm, ws, C, a.dC);
    for (int i = threadIdx.x; i < a.H; i += MIXT) tws[i] = a.twH[j * a.H + i];
    DLWP_STAMP(17);
    __syncthreads();
    DLWP_STAMP(18);
    const int hs = fast ? mix_hstep_fast(a, kx, NS2, hl, tws, part) : mix_hstep(a, kx, tws, part, NS);
    DLWP_STAMP(19);
    __syncthreads();
    DLWP_STAMP(20);
    mix_fold(xh, part, BC, hs);
    __syncthreads();
    DLWP_STAMP(21);
    const long long mofs = ((long long)j * a.m2c + kx) * C, bstr = (long long)a.m1 * a.m2c * C;
    for (int bc = threadIdx.x; bc < BC; bc += MIXT) {
        const int b = fastdiv(bc, a.dC), c = bc - b * C;
        a.xhat[b * bstr + mofs + c] = xh[bc];
    }
    if (LDSW) mix_contract<0, 0>(xh, ws, C + 1, a.y + mofs, bstr, a.B, C);
    else mix_contract<0, 0>(xh, wm, C, a.y + mofs, bstr, a.B, C);
    DLWP_STAMP(24);
}

template <bool LDSW>
__global__ __launch_bounds__(MIXT) void fno_mix_bwd_kernel(MixDev a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int BC = a.B * a.C, C = a.C;
    const int NS = mix_split(BC, a.H < C ? a.H : C);
    const int NS2 = mix_slots2(a.B, C, a.H);
    float2* gh = reinterpret_cast<float2*>(smem);  // [BC]  ghat
    float2* xsv = gh + BC;                         // [BC]  saved xhat
    float2* part = xsv + BC;                       // [max(NS, NS2)][BC]
    float2* tws = part + (NS > NS2 ? NS : NS2) * BC;   // [H]
    float2* ws = tws + a.H;                        // [C][C+1]   (LDSW)
    int mode = blockIdx.x;
    {
        const int nm = gridDim.x, full = (nm / 8) * 8;
        if (mode < full) mode = (mode % 8) * (nm / 8) + mode / 8;
    }
    const int kx = mode / a.m1, j = mode - kx * a.m1;
    const long long wofs = ((long long)(j * a.m2c + kx) * C) * C;
    const bool fast = (C % 2 == 0) && (BC / 2) * NS2 <= MIXT && NS2 >= 1;
    const float2* wm = a.wspec + wofs;
    HstepLoads hl;
    if (fast) mix_hstep_issue(a, kx, NS2, hl, 0);
    if (LDSW) mix_stage_w(wm, ws, C, a.dC);
    for (int i = threadIdx.x; i < a.H; i += MIXT) tws[i] = a.twH[j * a.H + i];
    const long long mofs = ((long long)j * a.m2c + kx) * C, bstr = (long long)a.m1 * a.m2c * C;
    for (int bc = threadIdx.x; bc < BC; bc += MIXT) {
        const int b = fastdiv(bc, a.dC), c = bc - b * C;
        xsv[bc] = a.xhat_in[b * bstr + mofs + c];
    }
    float2* gw = a.g_wspec + wofs;  // this workgroup owns the mode's weight-gradient slice
    __syncthreads();
    const int hs = fast ? mix_hstep_fast(a, kx, NS2, hl, tws, part) : mix_hstep(a, kx, tws, part, NS);
    __syncthreads();
    mix_fold(gh, part, BC, hs);
    __syncthreads();
    // gw[i][o] += sum_b conj(xhat[b][i]) ghat[b][o]: tile [16 i x 16 o], K = 2 B, k = 2 b' + ri on the lane group g.
    // Tiles are dealt from the LAST wave downwards and the gx column tiles below from the first wave upwards, so that at
    // narrow widths (4 + 2 work items for 8 waves) no wave gets both.
    {
        const int lane = lane_id(), w = wave_id(), r = lane & 15, g = lane >> 4;
        constexpr int NW = MIXT / 64;
        const int nt = (C + 15) / 16, ri = g & 1;
        for (int t = NW - 1 - w; t < nt * nt; t += NW) {
            const int it = t / nt, ot = t - it * nt;
            const int i = 16 * it + r, o = 16 * ot + r;          // A row (i) / B column (o) of this lane
            float2 old[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int io = min(16 * it + 4 * g + q, C - 1);
                old[q] = gw[(long long)io * C + min(o, C - 1)];
            }
            f32x4 are = {0.f, 0.f, 0.f, 0.f}, aim = {0.f, 0.f, 0.f, 0.f};
            for (int b0 = 0; b0 < a.B; b0 += 2) {
                const int b = b0 + (g >> 1);
                const bool okb = b < a.B;
                const float2 xv = (okb && i < C) ? xsv[b * C + i] : make_float2(0.f, 0.f);
                const float2 gv = (okb && o < C) ? gh[b * C + o] : make_float2(0.f, 0.f);
                const float av = ri ? xv.y : xv.x;
                are = mfma16(av, ri ? gv.y : gv.x, are);         // xr gr + xi gi
                aim = mfma16(av, ri ? -gv.x : gv.y, aim);        // xr gi - xi gr
            }
            if (o < C) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int io = 16 * it + 4 * g + q;
                    if (io < C) gw[(long long)io * C + o] = make_float2(old[q].x + are[q], old[q].y + aim[q]);
                }
            }
        }
    }
    // gx[b][i] = sum_o ghat[b][o] conj(w[i][o])
    if (LDSW) mix_contract<1, 1>(gh, ws, C + 1, a.y + mofs, bstr, a.B, C);
    else mix_contract<1, 1>(gh, wm, C, a.y + mofs, bstr, a.B, C);
}

// ---- wide layers (and 3-D plans): the per-mode stage as TWO launches, so that the weight stream can be split over the chip.
// The fused kernel above has one workgroup per mode: at 217 channels each streams a 377 KB weight slice alone (84 workgroups on
// 256 CUs, 0.9 TB/s), and splitting it would make every part re-read the mode's x1 slice for the H-axis step.
//   hstep : xhat[b][j][kx][c] = sum_h twH[j][h] x1[b][h][kx][c] on the matrix cores -- per kept column kx it is the complex GEMM
//           [m1 x H] . [H x C] with the row-frequency twiddles as the A operand (rows (j, re) = [tr, -ti], (j, im) = [ti, tr],
//           K = (h, re|im) interleaved), x1 read ONCE; one wave = (sample, column, 16 channels, pair of 16-row tiles)
//   cmix  : the complex channel contraction of mix_contract, grid = modes x column-tile groups; backward: gx and the weight
//           gradient for the group's rows of the weight slice (contiguous rows: read once, updated in place)
struct HstepDev {
    const float2* x1; const float2* twH; float2* out;
    int B, C, H, m1, m2c, ntc, nmp;       // ntc = channel tiles, nmp = pairs of 16-row (j, re|im) tiles
};

__global__ __launch_bounds__(256) void fno_hstep_kernel(HstepDev a) {
    const int lane = lane_id(), r = lane & 15, g = lane >> 4;
    long long item = (long long)blockIdx.x * 4 + wave_id();
    const long long nitems = (long long)a.B * a.m2c * a.ntc * a.nmp;
    if (item >= nitems) return;
    const int mp = (int)(item % a.nmp); item /= a.nmp;
    const int ct = (int)(item % a.ntc); item /= a.ntc;
    const int kx = (int)(item % a.m2c), b = (int)(item / a.m2c);
    const int c = 16 * ct + r, cc = c < a.C ? c : a.C - 1;
    const float2* xb = a.x1 + (((long long)b * a.H) * a.m2c + kx) * a.C + cc;
    const long long hstride = (long long)a.m2c * a.C;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const int nkc = (2 * a.H + 15) / 16;
    for (int kc0 = 0; kc0 < nkc; kc0 += 4) {
        f32x4 bf[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {          // B fragment: x1 rows h0, h0 + 1 (clamped loads, masked afterwards)
            const int h0 = 8 * (kc0 + q) + 2 * g;
            const float2 v0 = xb[(long long)min(h0, a.H - 1) * hstride], v1 = xb[(long long)min(h0 + 1, a.H - 1) * hstride];
            const bool ok0 = h0 < a.H && c < a.C, ok1 = h0 + 1 < a.H && c < a.C;
            bf[q] = f32x4{ok0 ? v0.x : 0.f, ok0 ? v0.y : 0.f, ok1 ? v1.x : 0.f, ok1 ? v1.y : 0.f};
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int h0 = 8 * (kc0 + q) + 2 * g;
#pragma unroll
            for (int t = 0; t < 2; ++t) {      // A fragment of row m = (j, part): twiddles of rows h0, h0 + 1
                const int m = 16 * (2 * mp + t) + r, j = m >> 1, part = m & 1;
                f32x4 af = {0.f, 0.f, 0.f, 0.f};
                if (j < a.m1 && h0 < a.H) {
                    const float2 t0 = a.twH[(long long)j * a.H + h0];
                    const float2 t1 = h0 + 1 < a.H ? a.twH[(long long)j * a.H + h0 + 1] : make_float2(0.f, 0.f);
                    af = part ? f32x4{t0.y, t0.x, t1.y, t1.x} : f32x4{t0.x, -t0.y, t1.x, -t1.y};
                }
                acc[t] = mfma16_chunk(af, bf[q], acc[t]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {              // rows 4g .. 4g+3 = frequencies j0, j0 + 1, (re, im) each
        const int j0 = 8 * (2 * mp + t) + 2 * g;
        if (c < a.C) {
            if (j0 < a.m1) a.out[(((long long)b * a.m1 + j0) * a.m2c + kx) * a.C + c] = make_float2(acc[t][0], acc[t][1]);
            if (j0 + 1 < a.m1) a.out[(((long long)b * a.m1 + j0 + 1) * a.m2c + kx) * a.C + c] = make_float2(acc[t][2], acc[t][3]);
        }
    }
}

struct CmixDev {
    const float2* xhat; const float2* ghat; const float2* wspec; float2* y; float2* g_wspec;
    int B, C, m1, m2c, tiles_per_wg;
    FastDiv dC;
};

// forward: y[b][mode][o] = sum_i xhat[b][mode][i] w[mode][i][o] for the workgroup's column tiles
__global__ __launch_bounds__(MIXT) void fno_cmix_fwd_kernel(CmixDev a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float2* xh = reinterpret_cast<float2*>(smem);      // [B][C]
    const int C = a.C, mode = blockIdx.x;
    const long long mofs = (long long)mode * C, bstr = (long long)a.m1 * a.m2c * C;
    for (int bc = threadIdx.x; bc < a.B * C; bc += MIXT) {
        const int b = fastdiv(bc, a.dC), c = bc - b * C;
        xh[bc] = a.xhat[b * bstr + mofs + c];
    }
    __syncthreads();
    const int nt0 = blockIdx.y * a.tiles_per_wg;
    mix_contract<0, 0>(xh, a.wspec + mofs * C, C, a.y + mofs, bstr, a.B, C, nt0, nt0 + a.tiles_per_wg);
}

// backward: gx[b][mode][i] = sum_o ghat[b][o] conj(w[i][o]) and gw[i][o] += sum_b conj(xhat[b][i]) ghat[b][o] for the workgroup's rows i
__global__ __launch_bounds__(MIXT) void fno_cmix_bwd_kernel(CmixDev a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float2* gh = reinterpret_cast<float2*>(smem);      // [B][C]
    float2* xsv = gh + a.B * a.C;                      // [B][C]
    const int C = a.C, mode = blockIdx.x;
    const long long mofs = (long long)mode * C, bstr = (long long)a.m1 * a.m2c * C;
    for (int bc = threadIdx.x; bc < a.B * C; bc += MIXT) {
        const int b = fastdiv(bc, a.dC), c = bc - b * C;
        gh[bc] = a.ghat[b * bstr + mofs + c];
        xsv[bc] = a.xhat[b * bstr + mofs + c];
    }
    __syncthreads();
    const int nt = (C + 15) / 16, it0 = blockIdx.y * a.tiles_per_wg, it1 = min(nt, it0 + a.tiles_per_wg);
    const float2* wm = a.wspec + mofs * C;
    float2* gw = a.g_wspec + mofs * C;
    mix_contract<1, 1>(gh, wm, C, a.y + mofs, bstr, a.B, C, it0, it1);
    {
        const int lane = lane_id(), w = wave_id(), r = lane & 15, g = lane >> 4;
        constexpr int NW = MIXT / 64;
        const int ri = g & 1;
        for (int t = NW - 1 - w; t < (it1 - it0) * nt; t += NW) {
            const int it = it0 + t / nt, ot = t % nt;
            const int i = 16 * it + r, o = 16 * ot + r;
            float2 old[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) old[q] = gw[(long long)min(16 * it + 4 * g + q, C - 1) * C + min(o, C - 1)];
            f32x4 are = {0.f, 0.f, 0.f, 0.f}, aim = {0.f, 0.f, 0.f, 0.f};
            for (int b0 = 0; b0 < a.B; b0 += 2) {
                const int b = b0 + (g >> 1);
                const bool okb = b < a.B;
                const float2 xv = (okb && i < C) ? xsv[b * C + i] : make_float2(0.f, 0.f);
                const float2 gv = (okb && o < C) ? gh[b * C + o] : make_float2(0.f, 0.f);
                const float av = ri ? xv.y : xv.x;
                are = mfma16(av, ri ? gv.y : gv.x, are);
                aim = mfma16(av, ri ? -gv.x : gv.y, aim);
            }
            if (o < C) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int io = 16 * it + 4 * g + q;
                    if (io < C) gw[(long long)io * C + o] = make_float2(old[q].x + are[q], old[q].y + aim[q]);
                }
            }
        }
    }
}

template <typename K>
int set_lds(K kernel, size_t bytes, const char* what) {
    return dlwp_ensure_lds(reinterpret_cast<const void*>(kernel), bytes, what);
}

template <typename T>
int upload(T** dst, const std::vector<T>& host) {
    DLWP_HIP(hipMalloc(reinterpret_cast<void**>(dst), host.size() * sizeof(T)));
    DLWP_HIP(hipMemcpy(*dst, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return DLWP_OK;
}

}  // namespace

// kept signed frequencies of an fftshift-ed axis of length n (neuralop SpectralConv slicing, App. A-1)
static std::vector<int> kept_freqs(int n, int m) {
    const int start = n - m, lo = start / 2;
    std::vector<int> k(m);
    for (int j = 0; j < m; ++j) k[j] = lo + j - n / 2;
    return k;
}

// T == 1: the 2-D plan.  T > 1: the (T*H) x W image of a [T][H][W] volume with m0*m1 separable (time, row) frequencies.
static int plan_create_impl(int C, int T, int H, int W, int m0, int m1, int m2c, dlwp_fno_plan** out) {
    DLWP_REQUIRE(out, DLWP_E_INVALID, "fno_plan_create: out is NULL");
    DLWP_REQUIRE(C > 0 && T > 0 && H > 0 && W > 0 && m0 > 0 && m1 > 0 && m2c > 0, DLWP_E_INVALID, "fno_plan_create: bad dims");
    DLWP_REQUIRE(W % 16 == 0, DLWP_E_UNSUPPORTED, "fno_plan_create: W must be a multiple of 16 (got %d)", W);
    DLWP_REQUIRE(H % 2 == 0, DLWP_E_UNSUPPORTED, "fno_plan_create: H must be even (got %d)", H);
    DLWP_REQUIRE(T == 1 || T % 2 == 0, DLWP_E_UNSUPPORTED, "fno_plan_create3d: the time axis must be even (got %d)", T);
    DLWP_REQUIRE(m1 <= H && m0 <= T && m2c <= W / 2 + 1, DLWP_E_INVALID, "fno_plan_create: more modes than the grid has");
    DLWP_REQUIRE(C <= 1024, DLWP_E_UNSUPPORTED, "fno_plan_create: hidden_channels <= 1024 supported (got %d)", C);
    DLWP_REQUIRE(2 * m2c <= 32, DLWP_E_UNSUPPORTED, "fno_plan_create: n_modes[-1]/2+1 <= 16 supported (got %d)", m2c);
    dlwp_fno_plan* p = new dlwp_fno_plan();
    const int HH = T * H, MM = m0 * m1;
    p->C = C; p->H = HH; p->W = W; p->m1 = MM; p->m2c = m2c;
    p->C_pad = round_up(C, 16);
    p->NP = round_up(2 * m2c, 16);
    p->force_wide = T > 1;
    const double PI = 3.14159265358979323846;
    const std::vector<int> kt = kept_freqs(T, m0), kh = kept_freqs(H, m1);
    std::vector<float2> tw((size_t)MM * HH);
    for (int a = 0; a < m0; ++a)
        for (int j = 0; j < m1; ++j)
            for (int t = 0; t < T; ++t)
                for (int h = 0; h < H; ++h) {
                    const long long at = (((long long)kt[a] * t) % T + T) % T, ah = (((long long)kh[j] * h) % H + H) % H;
                    const double ang = -2.0 * PI * ((double)at / T + (double)ah / H);
                    tw[((size_t)a * m1 + j) * HH + (size_t)t * H + h] = make_float2((float)cos(ang), (float)sin(ang));
                }
    std::vector<float> ft_fwd((size_t)p->NP * W, 0.f), ft_adj(ft_fwd), g_inv(ft_fwd), g_adj(ft_fwd);
    const double inv_hw = 1.0 / ((double)HH * W);
    for (int kx = 0; kx < m2c; ++kx) {
        const double ck = (kx == 0 || (W % 2 == 0 && kx == W / 2)) ? 1.0 : 2.0;
        for (int w = 0; w < W; ++w) {
            const long long kw = ((long long)kx * w) % W;
            const double ang = 2.0 * PI * (double)kw / W;
            const double c = cos(ang), s = sin(ang);
            ft_fwd[(size_t)(2 * kx) * W + w] = (float)(c * inv_hw);
            ft_fwd[(size_t)(2 * kx + 1) * W + w] = (float)(-s * inv_hw);
            ft_adj[(size_t)(2 * kx) * W + w] = (float)(c * ck);
            ft_adj[(size_t)(2 * kx + 1) * W + w] = (float)(-s * ck);
            g_inv[(size_t)(2 * kx) * W + w] = (float)(c * ck);
            g_inv[(size_t)(2 * kx + 1) * W + w] = (float)(-s * ck);
            g_adj[(size_t)(2 * kx) * W + w] = (float)(c * inv_hw);
            g_adj[(size_t)(2 * kx + 1) * W + w] = (float)(-s * inv_hw);
        }
    }
    int rc;
    if ((rc = upload(&p->twH, tw)) || (rc = upload(&p->FT_fwd, ft_fwd)) || (rc = upload(&p->FT_adj, ft_adj)) ||
        (rc = upload(&p->G_inv, g_inv)) || (rc = upload(&p->G_adj, g_adj))) {
        delete p;
        return rc;
    }
    *out = p;
    return DLWP_OK;
}

extern "C" int dlwp_fno_plan_create(int C, int H, int W, int m1, int m2c, dlwp_fno_plan** out) {
    return plan_create_impl(C, 1, H, W, 1, m1, m2c, out);
}

extern "C" int dlwp_fno_plan_create3d(int C, int T, int H, int W, int m0, int m1, int m2c, dlwp_fno_plan** out) {
    DLWP_REQUIRE(T > 1, DLWP_E_INVALID, "fno_plan_create3d: T must be > 1 (use dlwp_fno_plan_create for images)");
    return plan_create_impl(C, T, H, W, m0, m1, m2c, out);
}

extern "C" void dlwp_fno_plan_destroy(dlwp_fno_plan* p) {
    if (!p) return;
    (void)hipFree(p->twH); (void)hipFree(p->FT_fwd); (void)hipFree(p->FT_adj);
    (void)hipFree(p->G_inv); (void)hipFree(p->G_adj);
    delete p;
}

static size_t x1_elems(const dlwp_fno_plan* p, int B) { return (size_t)B * p->H * p->m2c * p->C; }
static size_t spec_elems(const dlwp_fno_plan* p, int B) { return (size_t)B * p->m1 * p->m2c * p->C; }
static size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

extern "C" size_t dlwp_fno_block_workspace_bytes(const dlwp_fno_plan* p, int B) {
    return align256(x1_elems(p, B) * sizeof(float2)) + align256(spec_elems(p, B) * sizeof(float2));
}

#define DISPATCH_NCB_NBN(NCBV, NBNV, MACRO)                      \
    switch ((NCBV) * 10 + (NBNV)) {                              \
        case 11: MACRO(1, 1) break; case 12: MACRO(1, 2) break;  \
        case 21: MACRO(2, 1) break; case 22: MACRO(2, 2) break;  \
        case 31: MACRO(3, 1) break; case 32: MACRO(3, 2) break;  \
        case 41: MACRO(4, 1) break; case 42: MACRO(4, 2) break;  \
        default:                                                 \
            dlwp_set_error("fno: unsupported tile shape");       \
            return DLWP_E_UNSUPPORTED;                           \
    }

int dlwp_fno_rows_dft(const dlwp_fno_plan* p, const float* x, int act_in, int adjoint, float2* x1, int B,
                      hipStream_t stream) {
    if (dlwp_fno_is_wide(p)) return dlwp_fno_rows_dft_wide(p, x, act_in, adjoint, x1, B, stream);
    RowsDev a{x, x1, adjoint ? p->FT_adj : p->FT_fwd, act_in, p->C, p->H, p->W, p->m2c, p->C_pad, p->NP};
    const int LDP = p->W + 4;
    const size_t lds = sizeof(float) * ((size_t)p->C_pad * LDP + (size_t)p->NP * LDP + (size_t)4 * p->C_pad * p->NP);
    const dim3 grid(B * p->H), block(256);
    int rc;
#define LAUNCH(N, M)                                                                   \
    if ((rc = set_lds(fno_rows_kernel<N, M>, lds, "fno_rows")) != DLWP_OK) return rc; \
    hipLaunchKernelGGL((fno_rows_kernel<N, M>), grid, block, lds, stream, a);
    DISPATCH_NCB_NBN(p->C_pad / 16, p->NP / 16, LAUNCH)
#undef LAUNCH
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

static int mix_launch(const dlwp_fno_plan* p, bool bwd, MixDev& a, hipStream_t stream) {
    const int BC = a.B * a.C;
    int ns = MIXT / BC;
    if (ns < 1) ns = 1;
    const int lim = a.H < a.C ? a.H : a.C;
    if (ns > lim) ns = lim;
    const int ns2 = mix_slots2(a.B, a.C, a.H);
    if (ns2 > ns) ns = ns2;
    const bool ldsw = a.C <= 64;                    // the [C][C+1] complex weight image fits beside the H-step partials
    const size_t lds = sizeof(float2) * ((size_t)BC * ((bwd ? 2 : 1) + ns) + a.H + (ldsw ? (size_t)a.C * (a.C + 1) : 0));
    const dim3 grid(p->m1 * p->m2c), block(MIXT);
    int rc;
#define MIX_LAUNCH(KERNEL, NAME)                                                    \
    do {                                                                            \
        if ((rc = set_lds(KERNEL, lds, NAME)) != DLWP_OK) return rc;                \
        hipLaunchKernelGGL(KERNEL, grid, block, lds, stream, a);                    \
    } while (0)
    if (bwd) {
        if (ldsw) MIX_LAUNCH(fno_mix_bwd_kernel<true>, "fno_mix_bwd"); else MIX_LAUNCH(fno_mix_bwd_kernel<false>, "fno_mix_bwd");
    } else {
        if (ldsw) MIX_LAUNCH(fno_mix_fwd_kernel<true>, "fno_mix_fwd"); else MIX_LAUNCH(fno_mix_fwd_kernel<false>, "fno_mix_fwd");
    }
#undef MIX_LAUNCH
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

static int hstep_launch(const dlwp_fno_plan* p, const float2* x1, float2* out, int B, hipStream_t stream) {
    HstepDev a{x1, p->twH, out, B, p->C, p->H, p->m1, p->m2c, ceil_div(p->C, 16), ceil_div(ceil_div(2 * p->m1, 16), 2)};
    const long long items = (long long)B * p->m2c * a.ntc * a.nmp;
    hipLaunchKernelGGL(fno_hstep_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

// column-tile groups per mode: enough workgroups to cover the chip about twice (every group streams its own part of the slice)
static int cmix_tiles_per_wg(const dlwp_fno_plan* p) {
    const int nt = ceil_div(p->C, 16), modes = p->m1 * p->m2c;
    int groups = ceil_div(512, modes);
    if (groups > nt) groups = nt;
    if (groups < 1) groups = 1;
    return ceil_div(nt, groups);
}

int dlwp_fno_mix_fwd(const dlwp_fno_plan* p, const float2* x1, const float2* wspec, float2* xhat, float2* y,
                     int B, hipStream_t stream) {
    if (dlwp_fno_is_wide(p)) {
        int rc = hstep_launch(p, x1, xhat, B, stream);
        if (rc) return rc;
        CmixDev a{xhat, nullptr, wspec, y, nullptr, B, p->C, p->m1, p->m2c, cmix_tiles_per_wg(p), make_fastdiv(p->C)};
        const size_t lds = sizeof(float2) * (size_t)B * p->C;
        if ((rc = set_lds(fno_cmix_fwd_kernel, lds, "fno_cmix_fwd")) != DLWP_OK) return rc;
        hipLaunchKernelGGL(fno_cmix_fwd_kernel, dim3(p->m1 * p->m2c, ceil_div(ceil_div(p->C, 16), a.tiles_per_wg)), dim3(MIXT), lds,
                           stream, a);
        DLWP_LAUNCH_CHECK();
        return DLWP_OK;
    }
    MixDev a{x1, wspec, nullptr, xhat, y, nullptr, p->twH, B, p->C, p->H, p->m1, p->m2c,
             make_fastdiv(p->C), make_fastdiv(B * p->C), make_fastdiv(B * p->C / 2 > 0 ? B * p->C / 2 : 1)};
    return mix_launch(p, false, a, stream);
}

// (wide layers: the H-step result ghat [B][m1][m2c][C] is produced in gxhat and moved into the storage of g1 -- dead once the
// H-step has read it, and at least as large since m1 <= H -- because the contraction reads ghat of ALL channels while it
// writes gx into gxhat; the caller's g1 buffer is therefore overwritten)
int dlwp_fno_mix_bwd(const dlwp_fno_plan* p, const float2* g1, const float2* wspec, const float2* xhat,
                     float2* gxhat, float2* g_wspec, int B, hipStream_t stream) {
    if (dlwp_fno_is_wide(p)) {
        DLWP_REQUIRE(p->m1 <= p->H, DLWP_E_INVALID, "fno_mix_bwd: more row frequencies than rows");
        int rc = hstep_launch(p, g1, gxhat, B, stream);
        if (rc) return rc;
        const size_t n = (size_t)B * p->m1 * p->m2c * p->C;
        float2* ghat = const_cast<float2*>(g1);
        DLWP_HIP(hipMemcpyAsync(ghat, gxhat, n * sizeof(float2), hipMemcpyDeviceToDevice, stream));
        CmixDev a{xhat, ghat, wspec, gxhat, g_wspec, B, p->C, p->m1, p->m2c, cmix_tiles_per_wg(p), make_fastdiv(p->C)};
        const size_t lds = sizeof(float2) * (size_t)2 * B * p->C;
        if ((rc = set_lds(fno_cmix_bwd_kernel, lds, "fno_cmix_bwd")) != DLWP_OK) return rc;
        hipLaunchKernelGGL(fno_cmix_bwd_kernel, dim3(p->m1 * p->m2c, ceil_div(ceil_div(p->C, 16), a.tiles_per_wg)), dim3(MIXT), lds,
                           stream, a);
        DLWP_LAUNCH_CHECK();
        return DLWP_OK;
    }
    MixDev a{g1, wspec, xhat, nullptr, gxhat, g_wspec, p->twH, B, p->C, p->H, p->m1, p->m2c,
             make_fastdiv(p->C), make_fastdiv(B * p->C), make_fastdiv(B * p->C / 2 > 0 ? B * p->C / 2 : 1)};
    return mix_launch(p, true, a, stream);
}

int dlwp_fno_spatial(const dlwp_fno_plan* p, const dlwp_fno_spatial_args* s, hipStream_t stream) {
    if (dlwp_fno_is_wide(p)) {
        // channel-blocked form: the fused by-products (next block's row DFT, skip-weight / bias gradients) are separate launches
        int rc = dlwp_fno_spatial_wide(p, s, stream);
        if (rc) return rc;
        if (s->inverse_adjoint && (s->g_wskip || s->g_bias)) {
            DLWP_REQUIRE(s->g_wskip && s->g_bias && !s->gslab, DLWP_E_INVALID, "fno_spatial (wide): needs g_wskip and g_bias, no slab");
            if ((rc = dlwp_fno_skip_wgrad(p, s->tin, s->pprev, s->act_prev, s->g_wskip, s->g_bias, s->B, stream))) return rc;
        }
        if (s->x1_out) return dlwp_fno_rows_dft_wide(p, s->out, s->x1_act, s->x1_adjoint, s->x1_out, s->B, stream);
        return DLWP_OK;
    }
    SpatialDev a{};
    a.tin = s->tin; a.spec = s->spec; a.wskip = s->wskip; a.bias = s->bias; a.pprev = s->pprev;
    a.out = s->out; a.x1_out = s->x1_out; a.g_wskip = s->g_wskip; a.g_bias = s->g_bias;
    a.gslab = s->gslab; a.gslab_accumulate = s->gslab_accumulate;
    a.twH = p->twH;
    a.G = s->inverse_adjoint ? p->G_adj : p->G_inv;
    a.FT = s->x1_adjoint ? p->FT_adj : p->FT_fwd;
    a.act_tin = s->act_tin; a.transpose_w = s->transpose_w; a.act_prev = s->act_prev;
    a.x1_act = s->x1_act; a.is_bwd = s->inverse_adjoint;
    a.vec_w = (reinterpret_cast<uintptr_t>(s->wskip) & 15) == 0;
    a.dC = make_fastdiv(p->C);
    a.C = p->C; a.H = p->H; a.W = p->W; a.m1 = p->m1; a.m2c = p->m2c; a.C_pad = p->C_pad; a.NP = p->NP;
    const int LDP = p->W + 4;
    size_t dead = (size_t)p->NP * LDP + (size_t)p->C_pad * (p->C_pad + 4) + (size_t)p->C_pad * (p->NP + 4);
    if (dead < (size_t)4 * p->C_pad * p->C_pad) dead = (size_t)4 * p->C_pad * p->C_pad;
    const size_t lds = sizeof(float) * ((size_t)3 * p->C_pad * LDP + (size_t)p->NP * LDP +
                                        (size_t)4 * p->C_pad * p->NP + p->C_pad + dead);
    const dim3 grid(s->B * p->H), block(256);
    int rc;
    const int mode = !a.is_bwd ? 0 : (a.act_prev ? 1 : 2);
#define LAUNCH_MODE(N, M, MD)                                                                     \
    if ((rc = set_lds(fno_spatial_kernel<N, M, MD>, lds, "fno_spatial")) != DLWP_OK) return rc; \
    hipLaunchKernelGGL((fno_spatial_kernel<N, M, MD>), grid, block, lds, stream, a);
#define LAUNCH(N, M)                                   \
    if (mode == 0) { LAUNCH_MODE(N, M, 0) }            \
    else if (mode == 1) { LAUNCH_MODE(N, M, 1) }       \
    else { LAUNCH_MODE(N, M, 2) }
    DISPATCH_NCB_NBN(p->C_pad / 16, p->NP / 16, LAUNCH)
#undef LAUNCH
#undef LAUNCH_MODE
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_fno_block_fwd(const dlwp_fno_plan* p, const float* x, int act_in, const float* wspec,
                                  const float* wskip, const float* bias, float* pre, float* xhat, int B,
                                  void* workspace, void* stream_) {
    DLWP_REQUIRE(p && x && wspec && wskip && pre && xhat && workspace && B > 0, DLWP_E_INVALID,
                 "fno_block_fwd: NULL argument or B<=0");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    float2* x1 = static_cast<float2*>(workspace);
    float2* y = reinterpret_cast<float2*>(static_cast<char*>(workspace) + align256(x1_elems(p, B) * sizeof(float2)));
    int rc;
    if ((rc = dlwp_fno_rows_dft(p, x, act_in, 0, x1, B, stream))) return rc;
    if ((rc = dlwp_fno_mix_fwd(p, x1, reinterpret_cast<const float2*>(wspec), reinterpret_cast<float2*>(xhat), y, B, stream))) return rc;
    dlwp_fno_spatial_args s{};
    s.tin = x; s.act_tin = act_in; s.spec = y; s.wskip = wskip; s.bias = bias; s.out = pre; s.B = B;
    return dlwp_fno_spatial(p, &s, stream);
}

extern "C" int dlwp_fno_block_bwd(const dlwp_fno_plan* p, const float* x, int act_in, const float* wspec,
                                  const float* wskip, const float* g_pre, const float* xhat, float* g_x,
                                  float* g_wspec, float* g_wskip, float* g_bias, int B, void* workspace,
                                  void* stream_) {
    DLWP_REQUIRE(p && x && wspec && wskip && g_pre && xhat && g_x && g_wspec && workspace && B > 0,
                 DLWP_E_INVALID, "fno_block_bwd: NULL argument or B<=0");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    float2* g1 = static_cast<float2*>(workspace);
    float2* gxhat = reinterpret_cast<float2*>(static_cast<char*>(workspace) + align256(x1_elems(p, B) * sizeof(float2)));
    int rc;
    if ((rc = dlwp_fno_rows_dft(p, g_pre, 0, 1, g1, B, stream))) return rc;
    if ((rc = dlwp_fno_mix_bwd(p, g1, reinterpret_cast<const float2*>(wspec), reinterpret_cast<const float2*>(xhat),
                               gxhat, reinterpret_cast<float2*>(g_wspec), B, stream))) return rc;
    dlwp_fno_spatial_args s{};
    s.tin = g_pre; s.spec = gxhat; s.wskip = wskip; s.transpose_w = 1; s.pprev = x; s.act_prev = act_in;
    s.out = g_x; s.g_wskip = g_wskip; s.g_bias = g_bias; s.inverse_adjoint = 1; s.B = B;
    return dlwp_fno_spatial(p, &s, stream);
}

#ifdef DLWP_STAMPS
extern "C" int dlwp_debug_span_fno(unsigned long long* host_out, int reset) {
    DLWP_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dlwp_span), sizeof(unsigned long long) * 2));
    if (reset) {
        const unsigned long long init[2] = {~0ull, 0ull};
        DLWP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_dlwp_span), init, sizeof(init)));
    }
    return DLWP_OK;
}
extern "C" int dlwp_debug_stamps_fno(unsigned long long* host_out) {
    DLWP_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dlwp_stamps), sizeof(unsigned long long) * 32));
    return DLWP_OK;
}
#endif

long long dlwp_fno_gslab_stride(int C) { return ((long long)C * C + C + 3) & ~3LL; }

// one forward `spatial` launch exactly as the rollout issues it for an inner block (GELU on load,
// fused rows-DFT of gelu(result)); exposed so bench.py can time the dominant kernel with HIP events.
extern "C" int dlwp_fno_spatial_fwd_probe(const dlwp_fno_plan* p, const float* x, const float* spec, const float* wskip,
                                          const float* bias, float* pre, float* x1_out, int B, void* stream) {
    DLWP_REQUIRE(p && x && spec && wskip && pre && x1_out && B > 0, DLWP_E_INVALID, "spatial_fwd_probe: NULL argument");
    dlwp_fno_spatial_args s{};
    s.tin = x; s.act_tin = 1; s.spec = reinterpret_cast<const float2*>(spec); s.wskip = wskip; s.bias = bias;
    s.out = pre; s.x1_out = reinterpret_cast<float2*>(x1_out); s.x1_act = 1; s.B = B;
    return dlwp_fno_spatial(p, &s, static_cast<hipStream_t>(stream));
}

// one forward per-mode launch (H-axis step + complex channel contraction on MFMA) exactly as the rollout issues it; exposed
// so bench.py can time the spectral contraction with HIP events (roofline_mix)
extern "C" int dlwp_fno_mix_fwd_probe(const dlwp_fno_plan* p, const float* x1, const float* wspec, float* xhat, float* y, int B,
                                      void* stream) {
    DLWP_REQUIRE(p && x1 && wspec && xhat && y && B > 0, DLWP_E_INVALID, "mix_fwd_probe: NULL argument");
    return dlwp_fno_mix_fwd(p, reinterpret_cast<const float2*>(x1), reinterpret_cast<const float2*>(wspec),
                            reinterpret_cast<float2*>(xhat), reinterpret_cast<float2*>(y), B, static_cast<hipStream_t>(stream));
}

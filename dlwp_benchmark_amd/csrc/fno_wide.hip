// Wide FNO layers (hidden_channels > 64): the channel-blocked forms of the row / spatial stages and the lifting /
// projection MLPs as channels-first GEMMs.  The reference's published TFNO2D sweep runs hidden_channels up to 217
// (src/nsbench/scripts/train_commands.txt:83-91 <-> the 5k..32M budgets of src/nsbench/scripts/plot_results.py:58); the
// fused kernels of fno_block.hip / pwmlp.hip keep every channel of a row in LDS and stop at 64.
//
//   rows    : one workgroup = one image row x 64 channels: W-axis pruned DFT on MFMA (as fno_rows_kernel)
//   mix     : fno_block.hip's per-mode kernel (any width: the weight slice streams from L2 into MFMA fragments)
//   spatial : one workgroup = 64 pixels of one image row x 64 OUTPUT channels; K loop over 64-channel chunks of the input
//             (skip 1x1 convolution) followed by the 2*m2c-deep inverse W-axis step -- one concatenated-K MFMA GEMM, with
//             the inverse H-axis step of the workgroup's 64 channels computed on the fly, bias / GELU' epilogue
//   skip weight gradient, lifting and projection MLPs: dlwp_gemm_run (token_ops.hip) on [C, H*W] fields per sample: row
//             bias, GELU on the activated operand, bias gradients as row sums, batches combined with float atomics
// Hidden activations of the MLPs ([B, 256, H*W]) are stored for the backward pass here (HBM is 288 GB; the fused narrow
// kernels recompute them instead).
#include "common.hip.h"
#include "dlwpmi_internal.h"
#include "fno_rows.hip.h"

namespace {

constexpr int WSEG = 64;      // pixels per workgroup (segment of one image row)
constexpr int OBLK = 64;      // output channels per workgroup
constexpr int KBLK = 64;      // input channels per K chunk
constexpr int LDPW = WSEG + 4;
constexpr int LDKW = KBLK + 4;

struct WideRowsDev {
    const float* x; float2* x1; const float* FT;
    int act, C, H, W, m2c, NP;
};

// x1[b][h][kx][c] = sum_w act(x[b][c][h][w]) FT[2kx(+1)][w] for the 64 channels c0 .. c0+63 of one image row
template <int NBN>
__global__ __launch_bounds__(256) void fno_rows_wide_kernel(WideRowsDev a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LDP = a.W + 4;
    float* tile = smem;                    // [64][LDP]
    float* ft = tile + 64 * LDP;           // [NP][LDP]
    float* x1s = ft + a.NP * LDP;          // [4 waves][64][NP]
    const int tid = threadIdx.x;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H, c0 = blockIdx.y * 64;
    const int W4 = a.W / 4;
    for (int idx = tid; idx < 64 * W4; idx += 256) {
        const int c = idx / W4, x4 = idx % W4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 + c < a.C) v = *reinterpret_cast<const float4*>(&a.x[(((long long)b * a.C + c0 + c) * a.H + h) * a.W + 4 * x4]);
        *reinterpret_cast<float4*>(&tile[c * LDP + 4 * x4]) = v;
    }
    for (int idx = tid; idx < a.NP * W4; idx += 256) {
        const int n = idx / W4, x4 = idx % W4;
        *reinterpret_cast<float4*>(&ft[n * LDP + 4 * x4]) = *reinterpret_cast<const float4*>(&a.FT[n * a.W + 4 * x4]);
    }
    __syncthreads();
    tile_rows_dft<4, NBN>(tile, ft, x1s, LDP, a.NP, a.W / 16, a.act != 0);
    __syncthreads();
    float2* dst = a.x1 + ((long long)(b * a.H + h) * a.m2c) * a.C;
    const int cl = min(64, a.C - c0);
    for (int idx = tid; idx < a.m2c * cl; idx += 256) {
        const int kx = idx / cl, c = idx - kx * cl;
        float re = 0.f, im = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            re += x1s[(w * 64 + c) * a.NP + 2 * kx];
            im += x1s[(w * 64 + c) * a.NP + 2 * kx + 1];
        }
        dst[(long long)kx * a.C + c0 + c] = make_float2(re, im);
    }
}

struct WideSpatialDev {
    const float* tin; const float2* spec; const float* wskip; const float* bias; const float* pprev; float* out;
    const float2* twH; const float* G;
    int act_tin, transpose_w;
    int C, H, W, m1, m2c, NP, nseg;
};

// MODE 0: forward (+ bias); 1: backward, multiply by GELU'(pprev); 2: backward, block input was not activated
template <int MODE, int NBN>
__global__ __launch_bounds__(256) void fno_spatial_wide_kernel(WideSpatialDev a) {
    __shared__ __attribute__((aligned(16))) float tin_s[KBLK * LDPW];     // [i][p]
    __shared__ __attribute__((aligned(16))) float ks[OBLK * LDKW];        // [o][i]
    __shared__ __attribute__((aligned(16))) float gs[16 * NBN * LDPW];    // [n][p]
    __shared__ __attribute__((aligned(16))) float s1[OBLK * (16 * NBN + 4)];   // [o][n]
    constexpr int LDS1 = 16 * NBN + 4;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id(), r = lane & 15, g = lane >> 4;
    const int seg = blockIdx.x % a.nseg, bh = blockIdx.x / a.nseg, b = bh / a.H, h = bh - b * a.H;
    const int w0 = seg * WSEG, wv = min(WSEG, a.W - w0);     // valid pixels of this segment (multiple of 16)
    const int ob0 = blockIdx.y * OBLK;
    const long long HW = (long long)a.H * a.W;
    // ---- inverse H-axis step of this row for the workgroup's channels: s1[o][2kx(+1)] = sum_j spec[b][j][kx][o] conj(twH[j][h])
    for (int idx = tid; idx < OBLK * (8 * NBN); idx += 256) {
        const int kx = idx / OBLK, o = idx - kx * OBLK;
        float re = 0.f, im = 0.f;
        if (kx < a.m2c && ob0 + o < a.C) {
            const float2* sp = a.spec + (((long long)b * a.m1) * a.m2c + kx) * a.C + ob0 + o;
            const long long jstride = (long long)a.m2c * a.C;
#pragma unroll 4
            for (int j = 0; j < a.m1; ++j) {
                const float2 v = sp[j * jstride];
                const float2 t = a.twH[j * a.H + h];
                re += v.x * t.x + v.y * t.y;
                im += v.y * t.x - v.x * t.y;
            }
        }
        s1[o * LDS1 + 2 * kx] = re;
        s1[o * LDS1 + 2 * kx + 1] = im;
    }
    for (int idx = tid; idx < 16 * NBN * (WSEG / 4); idx += 256) {
        const int n = idx / (WSEG / 4), p4 = idx - n * (WSEG / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (4 * p4 < wv) v = *reinterpret_cast<const float4*>(&a.G[(long long)n * a.W + w0 + 4 * p4]);
        *reinterpret_cast<float4*>(&gs[n * LDPW + 4 * p4]) = v;
    }
    f32x4 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool active = 16 * w < wv;           // wave w owns pixels 16 w .. 16 w + 15 of the segment
    // K loop over 64-channel chunks, register-prefetched: the global loads of chunk i+1 are in flight while chunk i runs on
    // the matrix cores (one LDS image per operand, two barriers per chunk)
    float4 tv[4];
    float kv[16];
    auto issue = [&](int ic0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u = tid + 256 * q, i = u / (WSEG / 4), p4 = u - i * (WSEG / 4);
            tv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ic0 + i < a.C && 4 * p4 < wv)
                tv[q] = *reinterpret_cast<const float4*>(&a.tin[((long long)b * a.C + ic0 + i) * HW + (long long)h * a.W + w0 + 4 * p4]);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int u = tid + 256 * q, hi = u >> 6, lo = u & 63;       // lo runs along the contiguous index of wskip
            const int o = a.transpose_w ? lo : hi, i = a.transpose_w ? hi : lo;
            const bool ok = ob0 + o < a.C && ic0 + i < a.C;
            const long long src = a.transpose_w ? (long long)(ic0 + i) * a.C + ob0 + o : (long long)(ob0 + o) * a.C + ic0 + i;
            kv[q] = ok ? a.wskip[src] : 0.f;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u = tid + 256 * q, i = u / (WSEG / 4), p4 = u - i * (WSEG / 4);
            float4 v = tv[q];
            if (a.act_tin) { const f32x4 ga = gelu4(f32x4{v.x, v.y, v.z, v.w}); v = make_float4(ga[0], ga[1], ga[2], ga[3]); }
            *reinterpret_cast<float4*>(&tin_s[i * LDPW + 4 * p4]) = v;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int u = tid + 256 * q, hi = u >> 6, lo = u & 63;
            const int o = a.transpose_w ? lo : hi, i = a.transpose_w ? hi : lo;
            ks[o * LDKW + i] = kv[q];
        }
    };
    issue(0);
    commit();
    __syncthreads();                            // s1 / gs / the first chunk are in LDS
    for (int ic0 = 0; ic0 < a.C; ic0 += KBLK) {
        const bool more = ic0 + KBLK < a.C;
        if (more) issue(ic0 + KBLK);
        if (active) {
#pragma unroll
            for (int kc = 0; kc < KBLK / 16; ++kc) {
                f32x4 b4;
#pragma unroll
                for (int s = 0; s < 4; ++s) b4[s] = tin_s[(kc * 16 + 4 * g + s) * LDPW + 16 * w + r];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    acc[cb] = mfma16_chunk(*reinterpret_cast<const f32x4*>(&ks[(cb * 16 + r) * LDKW + kc * 16 + 4 * g]), b4, acc[cb]);
            }
        }
        if (more) {
            __syncthreads();                    // every wave has consumed the current chunk
            commit();
            __syncthreads();
        }
    }
    if (active) {
#pragma unroll
        for (int nc = 0; nc < NBN; ++nc) {
            f32x4 b4;
#pragma unroll
            for (int s = 0; s < 4; ++s) b4[s] = gs[(nc * 16 + 4 * g + s) * LDPW + 16 * w + r];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
                acc[cb] = mfma16_chunk(*reinterpret_cast<const f32x4*>(&s1[(cb * 16 + r) * LDS1 + nc * 16 + 4 * g]), b4, acc[cb]);
        }
        // epilogue: rows o = 16 cb + 4 g + j, column p = 16 w + r; the operands of every element are fetched before use
        float ep[4][4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = min(ob0 + cb * 16 + 4 * g + j, a.C - 1);
                if (MODE == 0) ep[cb][j] = a.bias ? a.bias[o] : 0.f;
                else if (MODE == 1) ep[cb][j] = a.pprev[((long long)b * a.C + o) * HW + (long long)h * a.W + w0 + 16 * w + r];
                else ep[cb][j] = 0.f;
            }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = ob0 + cb * 16 + 4 * g + j;
                float v = acc[cb][j];
                if (MODE == 0) v += ep[cb][j];
                else if (MODE == 1) v *= gelu_grad_f(ep[cb][j]);
                if (o < a.C) a.out[((long long)b * a.C + o) * HW + (long long)h * a.W + w0 + 16 * w + r] = v;
            }
    }
}

// ---- channel gather / scatter between the rollout's pointer tables and dense [B][C][P] fields
struct ChanTabDev {
    const float* const* src; float* const* dst; const long long* bstride;
    float* dense; int C; long long P;
};
__global__ __launch_bounds__(256) void gather_channels_kernel(ChanTabDev a) {
    const int c = blockIdx.y, b = blockIdx.z;
    const float* s = a.src[c] + (long long)b * a.bstride[c];
    float* d = a.dense + ((long long)b * a.C + c) * a.P;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; 4 * i < a.P; i += (long long)gridDim.x * 256)
        reinterpret_cast<float4*>(d)[i] = reinterpret_cast<const float4*>(s)[i];
}
__global__ __launch_bounds__(256) void scatter_add_channels_kernel(ChanTabDev a) {
    const int c = blockIdx.y, b = blockIdx.z;
    float* d = a.dst[c];
    if (!d) return;
    d += (long long)b * a.bstride[c];
    const float* s = a.dense + ((long long)b * a.C + c) * a.P;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; 4 * i < a.P; i += (long long)gridDim.x * 256) {
        float4 v = reinterpret_cast<const float4*>(s)[i];
        const float4 o = reinterpret_cast<float4*>(d)[i];
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        reinterpret_cast<float4*>(d)[i] = v;
    }
}

// gy[b][c][p] = g_out[b][c][p] + mse_scale (pred - target); optional gres[b][c][p] += gy (identity path of the residual)
struct ProjGyDev {
    const float *g_out, *pred, *target; float *gy, *gres;
    long long bs_out, bs_res, CP;     // batch strides of the trajectory buffers, channels * pixels per sample
    float mse_scale;
};
__global__ __launch_bounds__(256) void proj_gy_kernel(ProjGyDev a) {
    const int b = blockIdx.y;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.CP; i += (long long)gridDim.x * 256) {
        float v = a.g_out[(long long)b * a.bs_out + i];
        if (a.pred) v += a.mse_scale * (a.pred[(long long)b * a.bs_out + i] - a.target[(long long)b * a.bs_out + i]);
        a.gy[(long long)b * a.CP + i] = v;
        if (a.gres) a.gres[(long long)b * a.bs_res + i] += v;
    }
}

}  // namespace

int dlwp_fno_rows_dft_wide(const dlwp_fno_plan* p, const float* x, int act_in, int adjoint, float2* x1, int B,
                           hipStream_t stream) {
    WideRowsDev a{x, x1, adjoint ? p->FT_adj : p->FT_fwd, act_in, p->C, p->H, p->W, p->m2c, p->NP};
    const int LDP = p->W + 4;
    const size_t lds = sizeof(float) * ((size_t)64 * LDP + (size_t)p->NP * LDP + (size_t)4 * 64 * p->NP);
    const dim3 grid(B * p->H, ceil_div(p->C, 64)), block(256);
    int rc;
    if (p->NP == 16) {
        if ((rc = dlwp_ensure_lds(reinterpret_cast<const void*>(fno_rows_wide_kernel<1>), lds, "fno_rows_wide"))) return rc;
        hipLaunchKernelGGL(fno_rows_wide_kernel<1>, grid, block, lds, stream, a);
    } else {
        if ((rc = dlwp_ensure_lds(reinterpret_cast<const void*>(fno_rows_wide_kernel<2>), lds, "fno_rows_wide"))) return rc;
        hipLaunchKernelGGL(fno_rows_wide_kernel<2>, grid, block, lds, stream, a);
    }
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

// the spatial stage of a wide block: same argument block as the fused kernel; x1_out / gslab / g_wskip / g_bias are not
// served here (the caller runs dlwp_fno_rows_dft_wide and dlwp_fno_skip_wgrad)
int dlwp_fno_spatial_wide(const dlwp_fno_plan* p, const dlwp_fno_spatial_args* s, hipStream_t stream) {
    WideSpatialDev a{};
    a.tin = s->tin; a.spec = s->spec; a.wskip = s->wskip; a.bias = s->bias; a.pprev = s->pprev; a.out = s->out;
    a.twH = p->twH; a.G = s->inverse_adjoint ? p->G_adj : p->G_inv;
    a.act_tin = s->act_tin; a.transpose_w = s->transpose_w;
    a.C = p->C; a.H = p->H; a.W = p->W; a.m1 = p->m1; a.m2c = p->m2c; a.NP = p->NP;
    a.nseg = ceil_div(p->W, WSEG);
    const dim3 grid(s->B * p->H * a.nseg, ceil_div(p->C, OBLK)), block(256);
    const int mode = !s->inverse_adjoint ? 0 : (s->act_prev ? 1 : 2);
#define WIDE_LAUNCH(MD, NB) hipLaunchKernelGGL((fno_spatial_wide_kernel<MD, NB>), grid, block, 0, stream, a)
    if (p->NP == 16) {
        if (mode == 0) WIDE_LAUNCH(0, 1); else if (mode == 1) WIDE_LAUNCH(1, 1); else WIDE_LAUNCH(2, 1);
    } else {
        if (mode == 0) WIDE_LAUNCH(0, 2); else if (mode == 1) WIDE_LAUNCH(1, 2); else WIDE_LAUNCH(2, 2);
    }
#undef WIDE_LAUNCH
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

// g_wskip[o][i] += sum_{b,p} g_pre[b][o][p] act(x[b][i][p]),  g_bias[o] += sum_{b,p} g_pre[b][o][p]
int dlwp_fno_skip_wgrad(const dlwp_fno_plan* p, const float* g_pre, const float* x, int act_x, float* g_wskip, float* g_bias,
                        int B, hipStream_t stream) {
    const int P = p->H * p->W;
    dlwp_gemm_args g{};
    g.A = g_pre; g.B = x; g.C = g_wskip;
    g.M = p->C; g.N = p->C; g.K = P; g.lda = P; g.ldb = P; g.ldc = p->C; g.transA = 0; g.transB = 1;
    g.nb = B; g.sA = (long long)p->C * P; g.sB = (long long)p->C * P; g.sC = 0;
    g.accumulate = 1; g.act_b = act_x; g.rowsum = g_bias;
    return dlwp_gemm_run(g, stream);
}

int dlwp_gather_channels(const float* const* src_tab, const long long* bstride_tab, float* dense, int B, int C, long long P,
                         hipStream_t stream) {
    DLWP_REQUIRE(P % 4 == 0, DLWP_E_UNSUPPORTED, "gather_channels: H*W must be a multiple of 4");
    ChanTabDev a{src_tab, nullptr, bstride_tab, dense, C, P};
    hipLaunchKernelGGL(gather_channels_kernel, dim3((unsigned)std::min<long long>(16, (P / 4 + 255) / 256), C, B), dim3(256), 0,
                       stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}
int dlwp_scatter_add_channels(float* const* dst_tab, const long long* bstride_tab, const float* dense, int B, int C, long long P,
                              hipStream_t stream) {
    DLWP_REQUIRE(P % 4 == 0, DLWP_E_UNSUPPORTED, "scatter_add_channels: H*W must be a multiple of 4");
    ChanTabDev a{nullptr, dst_tab, bstride_tab, const_cast<float*>(dense), C, P};
    hipLaunchKernelGGL(scatter_add_channels_kernel, dim3((unsigned)std::min<long long>(16, (P / 4 + 255) / 256), C, B), dim3(256),
                       0, stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}
int dlwp_proj_gy(const float* g_out, const float* pred, const float* target, float mse_scale, float* gy, float* gres,
                 long long bs_out, long long bs_res, long long CP, int B, hipStream_t stream) {
    ProjGyDev a{g_out, pred, target, gy, gres, bs_out, bs_res, CP, mse_scale};
    hipLaunchKernelGGL(proj_gy_kernel, dim3((unsigned)std::min<long long>(64, (CP + 255) / 256), B), dim3(256), 0, stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

// ---- y = W2 gelu(W1 x + b1) + b2 (+ res) on [B][C][P] fields as two GEMMs per sample; zpre / act [B][Ch][P] are kept
int dlwp_cfmlp_fwd(const float* x, long long x_bs, const float* w1, const float* b1, const float* w2, const float* b2, float* y,
                   long long y_bs, const float* res, long long res_bs, float* zpre, float* act, int B, int Cin, int Ch, int Cout,
                   int P, hipStream_t stream, long long x_cs) {
    dlwp_gemm_args g{};
    g.A = w1; g.B = x; g.C = act; g.M = Ch; g.N = P; g.K = Cin; g.lda = Cin; g.ldb = x_cs ? (int)x_cs : P; g.ldc = P;
    g.nb = B; g.sA = 0; g.sB = x_bs; g.sC = (long long)Ch * P; g.bias = b1; g.bias_row = 1; g.act = 1; g.preact = zpre;
    int rc = dlwp_gemm_run(g, stream);
    if (rc) return rc;
    dlwp_gemm_args h{};
    h.A = w2; h.B = act; h.C = y; h.M = Cout; h.N = P; h.K = Ch; h.lda = Ch; h.ldb = P; h.ldc = P;
    h.nb = B; h.sA = 0; h.sB = (long long)Ch * P; h.sC = y_bs; h.bias = b2; h.bias_row = 1; h.residual = res; h.sR = res_bs;
    return dlwp_gemm_run(h, stream);
}

// gy [B][Cout][P] (batch stride gy_bs); gx (nullable) [B][Cin][P] (batch stride gx_bs) is overwritten; parameter gradients are
// ACCUMULATED into; gz: scratch [B][Ch][P]
int dlwp_cfmlp_bwd(const float* x, long long x_bs, const float* w1, const float* w2, const float* gy, long long gy_bs,
                   const float* zpre, const float* act, float* gx, long long gx_bs, float* gz, float* gw1, float* gb1, float* gw2,
                   float* gb2, int B, int Cin, int Ch, int Cout, int P, hipStream_t stream, long long x_cs, long long gx_cs) {
    int rc;
    dlwp_gemm_args a{};           // gz = (W2^T gy) * GELU'(zpre)
    a.A = w2; a.B = gy; a.C = gz; a.M = Ch; a.N = P; a.K = Cout; a.lda = Ch; a.ldb = P; a.ldc = P; a.transA = 1;
    a.nb = B; a.sB = gy_bs; a.sC = (long long)Ch * P; a.act = 4; a.residual = zpre; a.sR = (long long)Ch * P;
    if ((rc = dlwp_gemm_run(a, stream))) return rc;
    dlwp_gemm_args w2g{};         // gW2 += gy act^T, gb2 += rowsum(gy)
    w2g.A = gy; w2g.B = act; w2g.C = gw2; w2g.M = Cout; w2g.N = Ch; w2g.K = P; w2g.lda = P; w2g.ldb = P; w2g.ldc = Ch;
    w2g.transB = 1; w2g.nb = B; w2g.sA = gy_bs; w2g.sB = (long long)Ch * P; w2g.sC = 0; w2g.accumulate = 1; w2g.rowsum = gb2;
    if ((rc = dlwp_gemm_run(w2g, stream))) return rc;
    if (gx) {                     // gx = W1^T gz
        dlwp_gemm_args b{};
        b.A = w1; b.B = gz; b.C = gx; b.M = Cin; b.N = P; b.K = Ch; b.lda = Cin; b.ldb = P; b.ldc = gx_cs ? (int)gx_cs : P; b.transA = 1;
        b.nb = B; b.sB = (long long)Ch * P; b.sC = gx_bs;
        if ((rc = dlwp_gemm_run(b, stream))) return rc;
    }
    dlwp_gemm_args w1g{};         // gW1 += gz x^T, gb1 += rowsum(gz)
    w1g.A = gz; w1g.B = x; w1g.C = gw1; w1g.M = Ch; w1g.N = Cin; w1g.K = P; w1g.lda = P; w1g.ldb = x_cs ? (int)x_cs : P; w1g.ldc = Cin;
    w1g.transB = 1; w1g.nb = B; w1g.sA = (long long)Ch * P; w1g.sB = x_bs; w1g.sC = 0; w1g.accumulate = 1; w1g.rowsum = gb1;
    return dlwp_gemm_run(w1g, stream);
}

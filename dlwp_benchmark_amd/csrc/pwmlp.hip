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
// layer, keeps weight-gradient partials in registers over the 64 pixels and flushes them
// with one hardware float atomic per element per workgroup.
#include "common.cuh"
#include "dlwpmi_internal.h"

namespace {

constexpr int PT = 64;        // pixels per workgroup
constexpr int LDP = PT + 4;   // LDS row stride of pixel tiles (4*LDP % 32 == 16: conflict-free B reads)

__device__ __forceinline__ const float* chan_ptr(const dlwp_chan_src& s, int b, int c) {
    if (s.tab) return s.tab[c] ? s.tab[c] + (long long)b * s.tab_bstride[c] : nullptr;
    return s.base + (long long)b * s.bstride + (long long)c * s.cstride;
}
__device__ __forceinline__ float* chan_ptr(const dlwp_chan_dst& s, int b, int c) {
    if (s.tab) return s.tab[c] ? s.tab[c] + (long long)b * s.tab_bstride[c] : nullptr;
    return s.base ? s.base + (long long)b * s.bstride + (long long)c * s.cstride : nullptr;
}

struct FwdArgs {
    dlwp_chan_src x;
    const float *w1, *b1, *w2, *b2;
    dlwp_chan_dst y;
    dlwp_chan_src res;  // optional residual added to y (base==nullptr && tab==nullptr: none)
    int B, Cin, Ch, Cout, P, tiles_per_sample;
    int Cin_pad, Ch_pad, Cout_pad;
};

template <int NOB>
__global__ __launch_bounds__(256) void pwmlp_fwd_kernel(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LD1 = a.Cin_pad + 4, LD2 = a.Ch_pad + 4;
    float* xs = smem;
    float* w1s = xs + a.Cin_pad * LDP;
    float* w2s = w1s + a.Ch_pad * LD1;
    float* b1s = w2s + a.Cout_pad * LD2;
    float* b2s = b1s + a.Ch_pad;

    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int r = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.tiles_per_sample;
    const int p0 = (blockIdx.x % a.tiles_per_sample) * PT;

    for (int idx = tid; idx < a.Cin_pad * PT; idx += 256) {
        const int c = idx / PT, p = idx % PT;
        float v = 0.f;
        if (c < a.Cin && p0 + p < a.P) {
            const float* src = chan_ptr(a.x, b, c);
            if (src) v = src[p0 + p];
        }
        xs[c * LDP + p] = v;
    }
    for (int idx = tid; idx < a.Ch_pad * a.Cin_pad; idx += 256) {
        const int h = idx / a.Cin_pad, i = idx % a.Cin_pad;
        w1s[h * LD1 + i] = (h < a.Ch && i < a.Cin) ? a.w1[h * a.Cin + i] : 0.f;
    }
    for (int idx = tid; idx < a.Cout_pad * a.Ch_pad; idx += 256) {
        const int o = idx / a.Ch_pad, h = idx % a.Ch_pad;
        w2s[o * LD2 + h] = (o < a.Cout && h < a.Ch) ? a.w2[o * a.Ch + h] : 0.f;
    }
    for (int idx = tid; idx < a.Ch_pad; idx += 256) b1s[idx] = idx < a.Ch ? a.b1[idx] : 0.f;
    for (int idx = tid; idx < a.Cout_pad; idx += 256) b2s[idx] = idx < a.Cout ? a.b2[idx] : 0.f;
    __syncthreads();

    const int pw0 = w * 16;
    const int nkc = a.Cin_pad / 16, nhb = a.Ch_pad / 16;
    f32x4 oacc[NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) oacc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int hb = 0; hb < nhb; ++hb) {
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

    const int p = p0 + pw0 + r;
    if (p < a.P) {
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = ob * 16 + 4 * g + j;
                if (o < a.Cout) {
                    float v = oacc[ob][j] + b2s[o];
                    if (a.res.base || a.res.tab) {
                        const float* rp = chan_ptr(a.res, b, o);
                        if (rp) v += rp[p];
                    }
                    float* dst = chan_ptr(a.y, b, o);
                    if (dst) dst[p] = v;
                }
            }
        }
    }
}

struct BwdArgs {
    dlwp_chan_src x;
    const float *w1, *b1, *w2;
    dlwp_chan_src gy;            // upstream gradient (may be absent)
    dlwp_chan_src pred, target;  // optional MSE term: gy_eff += mse_scale * (pred - target)
    float mse_scale;
    dlwp_chan_dst gx;            // nullable; per-channel nullable in table mode
    int gx_accumulate;
    float *gw1, *gb1, *gw2, *gb2;  // accumulated with float atomics
    int B, Cin, Ch, Cout, P, tiles_per_sample;
    int Cin_pad, Ch_pad, Cout_pad;
};

template <int NIB, int NOB>
__global__ __launch_bounds__(256) void pwmlp_bwd_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LD1 = a.Cin_pad + 4, LDT = a.Cout_pad + 4;
    float* xs = smem;                        // [Cin_pad][LDP]
    float* gys = xs + a.Cin_pad * LDP;       // [Cout_pad][LDP]
    float* gxs = gys + a.Cout_pad * LDP;     // [Cin_pad][LDP]  cross-wave reduction of gx
    float* w1s = gxs + a.Cin_pad * LDP;      // [Ch_pad][LD1]
    float* w2ts = w1s + a.Ch_pad * LD1;      // [Ch_pad][LDT]   W2 transposed
    float* b1s = w2ts + a.Ch_pad * LDT;      // [Ch_pad]
    float* tr = b1s + a.Ch_pad;              // [4 waves][2][16*20] transpose scratch

    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int r = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.tiles_per_sample;
    const int p0 = (blockIdx.x % a.tiles_per_sample) * PT;

    for (int idx = tid; idx < a.Cin_pad * PT; idx += 256) {
        const int c = idx / PT, p = idx % PT;
        float v = 0.f;
        if (c < a.Cin && p0 + p < a.P) {
            const float* src = chan_ptr(a.x, b, c);
            if (src) v = src[p0 + p];
        }
        xs[c * LDP + p] = v;
        gxs[c * LDP + p] = 0.f;
    }
    const bool has_gy = a.gy.base || a.gy.tab;
    const bool has_mse = a.pred.base || a.pred.tab;
    for (int idx = tid; idx < a.Cout_pad * PT; idx += 256) {
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
    for (int idx = tid; idx < a.Ch_pad * a.Cin_pad; idx += 256) {
        const int h = idx / a.Cin_pad, i = idx % a.Cin_pad;
        w1s[h * LD1 + i] = (h < a.Ch && i < a.Cin) ? a.w1[h * a.Cin + i] : 0.f;
    }
    for (int idx = tid; idx < a.Ch_pad * a.Cout_pad; idx += 256) {
        const int h = idx / a.Cout_pad, o = idx % a.Cout_pad;
        w2ts[h * LDT + o] = (h < a.Ch && o < a.Cout) ? a.w2[o * a.Ch + h] : 0.f;
    }
    for (int idx = tid; idx < a.Ch_pad; idx += 256) b1s[idx] = idx < a.Ch ? a.b1[idx] : 0.f;
    __syncthreads();

    float* T = tr + (w * 2 + 0) * 320;
    float* T2 = tr + (w * 2 + 1) * 320;
    const int nhb = a.Ch_pad / 16;

    f32x4 gxacc[4][NIB];
#pragma unroll
    for (int pb = 0; pb < 4; ++pb)
#pragma unroll
        for (int ib = 0; ib < NIB; ++ib) gxacc[pb][ib] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int hb = w; hb < nhb; hb += 4) {
        f32x4 a1[NIB], w2t[NOB];
#pragma unroll
        for (int kc = 0; kc < NIB; ++kc)
            a1[kc] = *reinterpret_cast<const f32x4*>(&w1s[(hb * 16 + r) * LD1 + kc * 16 + 4 * g]);
#pragma unroll
        for (int oc = 0; oc < NOB; ++oc)
            w2t[oc] = *reinterpret_cast<const f32x4*>(&w2ts[(hb * 16 + r) * LDT + oc * 16 + 4 * g]);
        float b1v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b1v[j] = b1s[hb * 16 + 4 * g + j];

        f32x4 gw2acc[NOB], gw1acc[NIB];
        f32x4 gb1acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) gw2acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ib = 0; ib < NIB; ++ib) gw1acc[ib] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            // recompute hidden pre-activation z[h][p]
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < NIB; ++kc) {
                f32x4 b4;
#pragma unroll
                for (int s = 0; s < 4; ++s) b4[s] = xs[(kc * 16 + 4 * g + s) * LDP + pb * 16 + r];
                z = mfma16_chunk(a1[kc], b4, z);
            }
            f32x4 act, gz;
            // g_a[h][p] = sum_o W2[o][h] gy[o][p]
            f32x4 ga = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int oc = 0; oc < NOB; ++oc) {
                f32x4 b4;
#pragma unroll
                for (int s = 0; s < 4; ++s) b4[s] = gys[(oc * 16 + 4 * g + s) * LDP + pb * 16 + r];
                ga = mfma16_chunk(w2t[oc], b4, ga);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float zz = z[j] + b1v[j];
                act[j] = gelu_f(zz);
                gz[j] = ga[j] * gelu_grad_f(zz);
                gb1acc[j] += gz[j];
            }
            // gx[i][p] += sum_h W1[h][i] gz[h][p]
#pragma unroll
            for (int ib = 0; ib < NIB; ++ib) {
                f32x4 a4;
#pragma unroll
                for (int s = 0; s < 4; ++s) a4[s] = w1s[(hb * 16 + 4 * g + s) * LD1 + ib * 16 + r];
                gxacc[pb][ib] = mfma16_chunk(a4, gz, gxacc[pb][ib]);
            }
            // wave-private transposes: T[h][p] <- act, T2[h][p] <- gz
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                T[(4 * g + j) * 20 + r] = act[j];
                T2[(4 * g + j) * 20 + r] = gz[j];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const f32x4 aT = *reinterpret_cast<const f32x4*>(&T[r * 20 + 4 * g]);    // act[h=r][p=4g+s]
            const f32x4 gzT = *reinterpret_cast<const f32x4*>(&T2[r * 20 + 4 * g]);  // gz [h=r][p=4g+s]
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // gW2[o][h] += sum_p gy[o][p] act[h][p]
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(&gys[(ob * 16 + r) * LDP + pb * 16 + 4 * g]);
                gw2acc[ob] = mfma16_chunk(a4, aT, gw2acc[ob]);
            }
            // gW1[h][i] += sum_p gz[h][p] x[i][p]
#pragma unroll
            for (int ib = 0; ib < NIB; ++ib) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(&xs[(ib * 16 + r) * LDP + pb * 16 + 4 * g]);
                gw1acc[ib] = mfma16_chunk(gzT, b4, gw1acc[ib]);
            }
        }
        // flush this hidden block's parameter gradients
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = ob * 16 + 4 * g + j, h = hb * 16 + r;
                if (o < a.Cout && h < a.Ch) atomic_add_f32(&a.gw2[o * a.Ch + h], gw2acc[ob][j]);
            }
#pragma unroll
        for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int h = hb * 16 + 4 * g + j, i = ib * 16 + r;
                if (h < a.Ch && i < a.Cin) atomic_add_f32(&a.gw1[h * a.Cin + i], gw1acc[ib][j]);
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = gb1acc[j];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            const int h = hb * 16 + 4 * g + j;
            if (r == 0 && h < a.Ch) atomic_add_f32(&a.gb1[h], v);
        }
    }

    // cross-wave reduction of gx through LDS
    const bool want_gx = a.gx.base || a.gx.tab;
    if (want_gx) {
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
            for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    atomicAdd(&gxs[(ib * 16 + 4 * g + j) * LDP + pb * 16 + r], gxacc[pb][ib][j]);
    }
    __syncthreads();
    if (want_gx) {
        for (int idx = tid; idx < a.Cin * PT; idx += 256) {
            const int c = idx / PT, p = idx % PT;
            if (p0 + p < a.P) {
                float* dst = chan_ptr(a.gx, b, c);
                if (dst) {
                    const float v = gxs[c * LDP + p];
                    dst[p0 + p] = a.gx_accumulate ? dst[p0 + p] + v : v;
                }
            }
        }
    }
    if (tid < a.Cout) {
        float s = 0.f;
        for (int p = 0; p < PT; ++p) s += gys[tid * LDP + p];
        atomic_add_f32(&a.gb2[tid], s);
    }
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
    return dlwp_ensure_lds(reinterpret_cast<const void*>(kernel), bytes, "pwmlp");
}

}  // namespace

int dlwp_pwmlp_fwd_ex(const dlwp_chan_src* x, const float* w1, const float* b1, const float* w2,
                      const float* b2, const dlwp_chan_dst* y, const dlwp_chan_src* res, int B,
                      int Cin, int Ch, int Cout, int P, hipStream_t stream) {
    DLWP_REQUIRE(B > 0 && Cin > 0 && Ch > 0 && Cout > 0 && P > 0, DLWP_E_INVALID,
                 "pwmlp_fwd: non-positive dimension");
    FwdArgs a{};
    a.x = *x; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.y = *y;
    if (res) a.res = *res;
    a.B = B; a.Cin = Cin; a.Ch = Ch; a.Cout = Cout; a.P = P;
    a.tiles_per_sample = ceil_div(P, PT);
    a.Cin_pad = round_up(Cin, 16); a.Ch_pad = round_up(Ch, 16); a.Cout_pad = round_up(Cout, 16);
    const int nob = a.Cout_pad / 16;
    DLWP_REQUIRE(nob <= 4 && a.Cin_pad <= 64, DLWP_E_UNSUPPORTED,
                 "pwmlp_fwd: Cin<=64 and Cout<=64 supported (got %d, %d)", Cin, Cout);
    const size_t lds = sizeof(float) * ((size_t)a.Cin_pad * LDP + (size_t)a.Ch_pad * (a.Cin_pad + 4) +
                                        (size_t)a.Cout_pad * (a.Ch_pad + 4) + a.Ch_pad + a.Cout_pad);
    const dim3 grid(B * a.tiles_per_sample), block(256);
    int rc;
#define LAUNCH(N)                                                         \
    if ((rc = set_lds(pwmlp_fwd_kernel<N>, lds)) != DLWP_OK) return rc;   \
    hipLaunchKernelGGL(pwmlp_fwd_kernel<N>, grid, block, lds, stream, a);
    switch (nob) {
        case 1: LAUNCH(1) break;
        case 2: LAUNCH(2) break;
        case 3: LAUNCH(3) break;
        default: LAUNCH(4) break;
    }
#undef LAUNCH
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

int dlwp_pwmlp_bwd_ex(const dlwp_chan_src* x, const float* w1, const float* b1, const float* w2,
                      const dlwp_chan_src* gy, const dlwp_chan_src* pred, const dlwp_chan_src* target,
                      float mse_scale, const dlwp_chan_dst* gx, int gx_accumulate, float* gw1,
                      float* gb1, float* gw2, float* gb2, int B, int Cin, int Ch, int Cout, int P,
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
    a.gw1 = gw1; a.gb1 = gb1; a.gw2 = gw2; a.gb2 = gb2;
    a.B = B; a.Cin = Cin; a.Ch = Ch; a.Cout = Cout; a.P = P;
    a.tiles_per_sample = ceil_div(P, PT);
    a.Cin_pad = round_up(Cin, 16); a.Ch_pad = round_up(Ch, 16); a.Cout_pad = round_up(Cout, 16);
    const int nib = a.Cin_pad / 16, nob = a.Cout_pad / 16;
    DLWP_REQUIRE(nib <= 4 && nob <= 4, DLWP_E_UNSUPPORTED,
                 "pwmlp_bwd: Cin<=64 and Cout<=64 supported (got %d, %d)", Cin, Cout);
    const size_t lds = sizeof(float) * ((size_t)2 * a.Cin_pad * LDP + (size_t)a.Cout_pad * LDP +
                                        (size_t)a.Ch_pad * (a.Cin_pad + 4) +
                                        (size_t)a.Ch_pad * (a.Cout_pad + 4) + a.Ch_pad + 4 * 2 * 320);
    const dim3 grid(B * a.tiles_per_sample), block(256);
    int rc;
#define LAUNCH(I, O)                                                       \
    if ((rc = set_lds(pwmlp_bwd_kernel<I, O>, lds)) != DLWP_OK) return rc; \
    hipLaunchKernelGGL((pwmlp_bwd_kernel<I, O>), grid, block, lds, stream, a);
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
    return DLWP_OK;
}

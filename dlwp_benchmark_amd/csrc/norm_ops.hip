// Token-level normalisation and elementwise building blocks of the AFNO / Swin / Pangu blocks (split from token_ops.hip in round 4:
// that file keeps the GEMM family): LayerNorm forward / backward (row, vector and wide kernels), activation backward, column sums
// (bias gradients) and the fp32 -> bf16 cast of the bf16-storage mode.
// Reference call sites (file:line under /root/reference/src/nsbench/models):
//   nn.LayerNorm   fourcastnet/fourcastnet.py:213 (eps 1e-6), swintransformer/swin_transformer.py:187,194 (eps 1e-5)
//   nn.GELU        fourcastnet.py:45, swin_transformer.py:37 (backward of the activation where it is not fused into a GEMM epilogue)
//   bias gradients of nn.Linear / position embeddings: column sums over the token dimension
#include <algorithm>
#include <cstdlib>
#include "common.hip.h"
#include "dlwpmi_internal.h"

namespace {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// Second output of the LayerNorm backward kernels (round 6): gx * row_scale[sample] as a bf16 array -- the operand the NEXT backward
// products read (the branch that ends in this residual stream applied a per-sample stochastic-depth scale; its backward wants the scaled
// gradient as bf16 for its two products).  Written from the registers that hold gx: the separate cast pass (19 MB at Pangu's 8192 x 384,
// 5.9 us of launch 28 times per step) disappears.  out == nullptr: nothing.
struct LnLowp {
    __bf16* out;
    const float* scale;      // [samples] or nullptr (1)
    int rows_per_sample;
};
// the row's scale, once per row (a 32-bit division: T < 2^31 rows)
__device__ __forceinline__ float ln_lowp_scale(const LnLowp& lp, long long row) {
    return (lp.out && lp.scale) ? lp.scale[(unsigned)row / (unsigned)lp.rows_per_sample] : 1.f;
}
__device__ __forceinline__ void ln_lowp_store4(const LnLowp& lp, float sc, long long off, const f32x4& v) {
    if (!lp.out) return;
    *reinterpret_cast<bf16x4*>(lp.out + off) = bf16x4{(__bf16)(v[0] * sc), (__bf16)(v[1] * sc), (__bf16)(v[2] * sc), (__bf16)(v[3] * sc)};
}



// ---- LayerNorm over the last dimension: one wave per row
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int T,
                                                            int C, float eps, int y_bf16) {
    const int row = blockIdx.x * 4 + wave_id(), lane = lane_id();
    if (row >= T) return;
    const float* xr = x + (long long)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mu = wave_sum64(s) / C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mu; v += d * d; }
    const float rs = rsqrtf(wave_sum64(v) / C + eps);
    if (y_bf16) {         // the output feeds a GEMM under bf16 storage: rounded once here instead of in every tile load
        __bf16* yh = reinterpret_cast<__bf16*>(y);
        for (int c = lane; c < C; c += 64) yh[(long long)row * C + c] = (__bf16)((xr[c] - mu) * rs * gamma[c] + beta[c]);
    } else {
        for (int c = lane; c < C; c += 64) y[(long long)row * C + c] = (xr[c] - mu) * rs * gamma[c] + beta[c];
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// Wide rows (256 < C <= 1024, C % 4 == 0): the row is read ONCE into NV float4 groups per lane (the scalar kernel above walks it
// three times with 4-byte loads), statistics in registers, 16-byte (bf16: 8-byte) stores.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_wide_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* __restrict__ y,
                                                                 float* __restrict__ mean, float* __restrict__ rstd, int T,
                                                                 int C, float eps, int y_bf16) {
    const int row = blockIdx.x * 4 + wave_id(), lane = lane_id();
    if (row >= T) return;
    const float* xr = x + (long long)row * C;
    f32x4 v[NV];
    bool ok[NV];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int c = 4 * lane + 256 * q;
        ok[q] = c < C;
        v[q] = *reinterpret_cast<const f32x4*>(xr + (ok[q] ? c : 0));
        if (ok[q]) s += (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
    }
    const float mu = wave_sum64(s) / C;
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < NV; ++q)
        if (ok[q]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float d = v[q][k] - mu; var += d * d; }
        }
    const float rs = rsqrtf(wave_sum64(var) / C + eps);
#pragma unroll
    for (int q = 0; q < NV; ++q)
        if (ok[q]) {
            const int c = 4 * lane + 256 * q;
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (v[q][k] - mu) * rs * gm[k] + bt[k];
            if (y_bf16)
                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(y) + (long long)row * C + c) =
                    bf16x4{(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
            else
                *reinterpret_cast<f32x4*>(y + (long long)row * C + c) = o;
        }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// gx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat)) (+ gadd: the gradient that reached x along the residual
// branch of a pre-norm block, so the two paths meet here instead of in a separate add); ggamma/gbeta partials via atomics
template <int NQ>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gy, const float* __restrict__ gadd,
                                                            float* __restrict__ gx, float* ggamma, float* gbeta, int T, int C,
                                                            int rows_per_block, int gy_bf16, LnLowp lp) {
    extern __shared__ float sm[];            // [2][C] per-block partial of ggamma, gbeta
    float* sg = sm;
    float* sb = sm + C;
    for (int c = threadIdx.x; c < 2 * C; c += 256) sm[c] = 0.f;
    __syncthreads();
    const int lane = lane_id(), w = wave_id();
    const int row0 = blockIdx.x * rows_per_block;
    // each wave owns the columns c = lane, lane+64, ... (C <= 64*NQ, checked by the host wrapper) of its rows and keeps the
    // column partials in registers.  The next row's loads are in flight while the current row is reduced (one wave per SIMD
    // would otherwise pay a full memory latency per row); loads use clamped addresses, masks are applied afterwards.
    float pg[NQ], pb[NQ], gam[NQ], xv[NQ], gv[NQ], av[NQ], xn[NQ], gn[NQ], an[NQ];
    bool okc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        pg[q] = 0.f; pb[q] = 0.f;
        okc[q] = lane + 64 * q < C;
        gam[q] = okc[q] ? gamma[min(lane + 64 * q, C - 1)] : 0.f;
    }
    auto load_row = [&](int row, float (&X)[NQ], float (&G)[NQ], float (&A)[NQ], float& mu, float& rs) {
        const int rc = min(row, T - 1);
        const float* xr = x + (long long)rc * C;
        const float* gr = gy + (long long)rc * C;
        const __bf16* gh = reinterpret_cast<const __bf16*>(gy) + (long long)rc * C;      // gy_bf16: the upstream gradient is a bf16 array
        const float* ar = gadd ? gadd + (long long)rc * C : nullptr;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = min(lane + 64 * q, C - 1);
            X[q] = xr[c];
            G[q] = gy_bf16 ? (float)gh[c] : gr[c];
            A[q] = ar ? ar[c] : 0.f;
        }
        mu = mean[rc];
        rs = rstd[rc];
    };
    float mu, rs, mun = 0.f, rsn = 0.f;
    load_row(row0 + w, xv, gv, av, mu, rs);
    for (int rr = w; rr < rows_per_block; rr += 4) {
        const int row = row0 + rr;
        if (row >= T) break;
        const bool more = rr + 4 < rows_per_block && row + 4 < T;
        if (more) load_row(row + 4, xn, gn, an, mun, rsn);
        float s1 = 0.f, s2 = 0.f, xh[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float g0 = okc[q] ? gv[q] : 0.f;
            xh[q] = okc[q] ? (xv[q] - mu) * rs : 0.f;
            const float gg = g0 * gam[q];
            s1 += gg;
            s2 += gg * xh[q];
            pg[q] += g0 * xh[q];
            pb[q] += g0;
        }
        s1 = wave_sum64(s1) / C;
        s2 = wave_sum64(s2) / C;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (okc[q]) {
                const float o1 = rs * (gv[q] * gam[q] - s1 - xh[q] * s2) + av[q];
                gx[(long long)row * C + lane + 64 * q] = o1;
                if (lp.out) lp.out[(long long)row * C + lane + 64 * q] = (__bf16)(o1 * ln_lowp_scale(lp, row));
            }
        if (more) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) { xv[q] = xn[q]; gv[q] = gn[q]; av[q] = an[q]; }
            mu = mun; rs = rsn;
        }
    }
    // combine the four waves' column partials one wave at a time (no LDS float atomics)
    for (int ww = 0; ww < 4; ++ww) {
        if (w == ww) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (okc[q]) { sg[lane + 64 * q] += pg[q]; sb[lane + 64 * q] += pb[q]; }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        atomic_add_f32(&ggamma[c], sg[c]);
        atomic_add_f32(&gbeta[c], sb[c]);
    }
}

// gz = gy * act'(z): act 1 GELU, 2 ReLU, 3 soft-shrink(lam)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ z, const float* __restrict__ gy,
                                                      float* __restrict__ gz, long long n, int act, float lam) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float v = z[i];
        float d = 1.f;
        if (act == 1) d = gelu_grad_f(v);
        else if (act == 2) d = v > 0.f ? 1.f : 0.f;
        else if (act == 3) d = (v > lam || v < -lam) ? 1.f : 0.f;
        gz[i] = gy[i] * d;
    }
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ z, const float* __restrict__ gy,
                                                       float* __restrict__ gz, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) gz[i] = gy[i] * gelu_grad_f(z[i]);
}

// ---- LayerNorm for narrow rows (C % 4 == 0, C <= 256): a lane owns four consecutive channels (16-byte accesses), LPR lanes
// share a row and a wave works on 64 / LPR rows at once.  The wave-per-row kernels above leave 25-60 % of a wave idle at C = 40
// ... 96 (Swin stages) and move 4 bytes per lane and instruction: 1.6 TB/s at 65536 x 96; this form reaches the HBM regime.
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int LPR>
__global__ __launch_bounds__(256) void layernorm_fwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ y,
                                                                float* __restrict__ mean, float* __restrict__ rstd, int T, int C,
                                                                float eps, int y_bf16) {
    constexpr int RPW = 64 / LPR;
    const int lane = lane_id(), l = lane % LPR, rr = lane / LPR;
    const long long row = ((long long)blockIdx.x * 4 + wave_id()) * RPW + rr;
    const bool okc = 4 * l < C, ok = okc && row < T;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f}, gm = v, bt = v;
    if (ok) v = *reinterpret_cast<const f32x4*>(x + row * C + 4 * l);
    if (okc) { gm = *reinterpret_cast<const f32x4*>(gamma + 4 * l); bt = *reinterpret_cast<const f32x4*>(beta + 4 * l); }
    const float mu = row_sum<LPR>((v[0] + v[1]) + (v[2] + v[3])) / C;
    f32x4 d;
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = okc ? v[k] - mu : 0.f;
    const float rs = rsqrtf(row_sum<LPR>((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) / C + eps);
    if (ok) {
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = d[k] * rs * gm[k] + bt[k];
        if (y_bf16) *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(y) + row * C + 4 * l) = bf16x4{(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
        else *reinterpret_cast<f32x4*>(y + row * C + 4 * l) = o;
        if (l == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

// Forward for rows of exactly 4 LPR NV floats (round 6; see layernorm_bwd_vecn_kernel): a lane holds NV 16-byte chunks of a row, a wave
// 64 / LPR rows, no idle lanes at the widths 96 / 192 / 384 / 768 (NV = 3).
template <int LPR, int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_vecn_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* __restrict__ y,
                                                                 float* __restrict__ mean, float* __restrict__ rstd, int T, int C,
                                                                 float eps, int y_bf16) {
    constexpr int RPW = 64 / LPR, CS = 4 * LPR;
    const int lane = lane_id(), l = lane % LPR, rr = lane / LPR;
    const long long row = ((long long)blockIdx.x * 4 + wave_id()) * RPW + rr;
    const bool ok = row < T;
    const long long o = (ok ? row : 0) * C + 4 * l;             // masked rows read row 0
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        v[c] = *reinterpret_cast<const f32x4*>(x + o + CS * c);
        s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
    }
    const float invC = 1.f / C;
    const float mu = row_sum<LPR>(s) * invC;
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[c][k] -= mu; var += v[c][k] * v[c][k]; }
    const float rs = rsqrtf(row_sum<LPR>(var) * invC + eps);
    if (ok) {
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 4 * l + CS * c), bt = *reinterpret_cast<const f32x4*>(beta + 4 * l + CS * c);
            f32x4 q;
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = v[c][k] * rs * gm[k] + bt[k];
            if (y_bf16) *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(y) + o + CS * c) = bf16x4{(__bf16)q[0], (__bf16)q[1], (__bf16)q[2], (__bf16)q[3]};
            else *reinterpret_cast<f32x4*>(y + o + CS * c) = q;
        }
        if (l == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

template <int LPR, int NW = 4>                 // NW waves per workgroup
__global__ __launch_bounds__(64 * NW) void layernorm_bwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gy, const float* __restrict__ gadd,
                                                                float* __restrict__ gx, float* ggamma, float* gbeta, int T, int C,
                                                                int rows_per_block, int gy_bf16, LnLowp lp) {
    extern __shared__ float sm[];            // [2][C] per-block partial of ggamma, gbeta
    constexpr int RPW = 64 / LPR;
    for (int c = threadIdx.x; c < 2 * C; c += 64 * NW) sm[c] = 0.f;
    __syncthreads();
    const int lane = lane_id(), w = wave_id(), l = lane % LPR, rr = lane / LPR;
    const bool okc = 4 * l < C;
    f32x4 gm = f32x4{0.f, 0.f, 0.f, 0.f}, pg = gm, pb = gm;
    if (okc) gm = *reinterpret_cast<const f32x4*>(gamma + 4 * l);
    const long long row0 = (long long)blockIdx.x * rows_per_block;
    const __bf16* gh = reinterpret_cast<const __bf16*>(gy);
    // UB row groups of a wave in flight at once (round 5: the one-group loop was a chain of dependent HBM round trips -- 38 us for the
    // 100 MB of Swin's 65536 x 96 layers; the loads of four groups travel together now)
#ifndef LN_BWD_UB
#define LN_BWD_UB 4          // 8 was measured in round 6 (250 VGPRs, two waves per SIMD): <32> 37.4 -> 38.9 us, <64> unchanged
#endif
    constexpr int UB = LN_BWD_UB;
    for (int it0 = w * RPW; it0 < rows_per_block; it0 += NW * RPW * UB) {
        f32x4 xv[UB], gv[UB], av[UB];
        float mu[UB], rs[UB];
        bool ok[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int it = it0 + NW * RPW * u;
            const long long row = row0 + it + rr;
            ok[u] = okc && row < T && it + rr < rows_per_block;
            xv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            gv[u] = xv[u];
            av[u] = xv[u];
            mu[u] = 0.f;
            rs[u] = 0.f;
            if (ok[u]) {
                const long long o = row * C + 4 * l;
                xv[u] = *reinterpret_cast<const f32x4*>(x + o);
                if (gy_bf16) {
                    const bf16x4 hv = *reinterpret_cast<const bf16x4*>(gh + o);
#pragma unroll
                    for (int k = 0; k < 4; ++k) gv[u][k] = (float)hv[k];
                } else {
                    gv[u] = *reinterpret_cast<const f32x4*>(gy + o);
                }
                if (gadd) av[u] = *reinterpret_cast<const f32x4*>(gadd + o);
                mu[u] = mean[row];
                rs[u] = rstd[row];
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const long long row = row0 + it0 + NW * RPW * u + rr;
            f32x4 xh, gg;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                xh[k] = ok[u] ? (xv[u][k] - mu[u]) * rs[u] : 0.f;
                gg[k] = gv[u][k] * gm[k];
                s1 += gg[k];
                s2 += gg[k] * xh[k];
                pg[k] += gv[u][k] * xh[k];
                pb[k] += gv[u][k];
            }
            s1 = row_sum<LPR>(s1) / C;
            s2 = row_sum<LPR>(s2) / C;
            if (ok[u]) {
                f32x4 o4;
#pragma unroll
                for (int k = 0; k < 4; ++k) o4[k] = rs[u] * (gg[k] - s1 - xh[k] * s2) + av[u][k];
                *reinterpret_cast<f32x4*>(gx + row * C + 4 * l) = o4;
                ln_lowp_store4(lp, ln_lowp_scale(lp, row), row * C + 4 * l, o4);
            }
        }
    }
    // fold the row slots of a wave (lanes l, l + LPR, ...), then the four waves one after the other (no LDS float atomics)
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
        for (int k = 0; k < 4; ++k) { pg[k] += __shfl_xor(pg[k], o); pb[k] += __shfl_xor(pb[k], o); }
    if (NW == 4) {
        for (int ww = 0; ww < 4; ++ww) {
            if (w == ww && rr == 0 && okc) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { sm[4 * l + k] += pg[k]; sm[C + 4 * l + k] += pb[k]; }
            }
            __syncthreads();
        }
    } else {                                 // many waves: LDS float adds instead of one barrier per wave
        if (rr == 0 && okc) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { atomicAdd(&sm[4 * l + k], pg[k]); atomicAdd(&sm[C + 4 * l + k], pb[k]); }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < C; c += 64 * NW) {
        atomic_add_f32(&ggamma[c], sm[c]);
        atomic_add_f32(&gbeta[c], sm[C + c]);
    }
}

// Rows whose width is EXACTLY 4 LPR NV floats (round 6: NV = 3 covers every feature width of the C4 / C5 token models -- 96, 192, 384, 768 --
// with no idle lanes: the one-chunk kernel above runs 96 columns on 32 lanes per row with 8 of them idle, 192 on 64 with 16 idle, and the
// wide-row kernel below runs 384 as two 256-column chunks with half of the second one idle).  A lane holds NV 16-byte chunks of a row
// (columns 4 l + 4 LPR c: every load instruction of the 64 / LPR rows of a wave is a set of whole 16 LPR-byte segments), UB row groups of
// a wave are in flight at once, eight waves per workgroup and one workgroup per CU (ln_bwd_waves), LDS float adds fold the waves' column
// partials, then the 2 C float atomics.
template <int LPR, int NV, int NW, int UB>
__global__ __launch_bounds__(64 * NW) void layernorm_bwd_vecn_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                     const float* __restrict__ gy, const float* __restrict__ gadd,
                                                                     float* __restrict__ gx, float* ggamma, float* gbeta, int T, int C,
                                                                     int rows_per_block, int gy_bf16, LnLowp lp) {
    extern __shared__ float sm[];            // [2][C] per-block partial of ggamma, gbeta
    constexpr int RPW = 64 / LPR, CS = 4 * LPR;      // rows per wave and group; column stride between a lane's chunks
    for (int c = threadIdx.x; c < 2 * C; c += 64 * NW) sm[c] = 0.f;
    __syncthreads();
    const int lane = lane_id(), w = wave_id(), l = lane % LPR, rr = lane / LPR;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 gm[NV], pg[NV], pb[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        gm[c] = *reinterpret_cast<const f32x4*>(gamma + 4 * l + CS * c);
        pg[c] = zero; pb[c] = zero;
    }
    const long long row0 = (long long)blockIdx.x * rows_per_block;
    const __bf16* gh = reinterpret_cast<const __bf16*>(gy);
    const float invC = 1.f / C;
    for (int it0 = w * RPW; it0 < rows_per_block; it0 += NW * RPW * UB) {
        f32x4 xv[UB][NV], gv[UB][NV], av[UB][NV];
        float mu[UB], rs[UB];
        bool ok[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int it = it0 + NW * RPW * u;
            const long long row = row0 + it + rr;
            ok[u] = row < T && it + rr < rows_per_block;
            const long long rc = ok[u] ? row : 0;                 // masked rows read row 0 (valid memory): no branch around the loads
            const long long o = rc * C + 4 * l;
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                xv[u][c] = *reinterpret_cast<const f32x4*>(x + o + CS * c);
                if (gy_bf16) {
                    const bf16x4 hv = *reinterpret_cast<const bf16x4*>(gh + o + CS * c);
                    gv[u][c] = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
                } else {
                    gv[u][c] = *reinterpret_cast<const f32x4*>(gy + o + CS * c);
                }
                av[u][c] = gadd ? *reinterpret_cast<const f32x4*>(gadd + o + CS * c) : zero;
            }
            mu[u] = mean[rc];
            rs[u] = rstd[rc];
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const long long row = row0 + it0 + NW * RPW * u + rr;
            f32x4 xh[NV], gg[NV];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int c = 0; c < NV; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float g0 = ok[u] ? gv[u][c][k] : 0.f;
                    xh[c][k] = ok[u] ? (xv[u][c][k] - mu[u]) * rs[u] : 0.f;
                    gg[c][k] = g0 * gm[c][k];
                    s1 += gg[c][k];
                    s2 += gg[c][k] * xh[c][k];
                    pg[c][k] += g0 * xh[c][k];
                    pb[c][k] += g0;
                }
            s1 = row_sum<LPR>(s1) * invC;
            s2 = row_sum<LPR>(s2) * invC;
            if (ok[u]) {
                const float lsc = ln_lowp_scale(lp, row);
#pragma unroll
                for (int c = 0; c < NV; ++c) {
                    f32x4 o4;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o4[k] = rs[u] * (gg[c][k] - s1 - xh[c][k] * s2) + av[u][c][k];
                    *reinterpret_cast<f32x4*>(gx + row * C + 4 * l + CS * c) = o4;
                    ln_lowp_store4(lp, lsc, row * C + 4 * l + CS * c, o4);
                }
            }
        }
    }
    // fold the row slots of a wave (lanes l, l + LPR, ...), then the waves through LDS float adds
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
        for (int c = 0; c < NV; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) { pg[c][k] += __shfl_xor(pg[c][k], o); pb[c][k] += __shfl_xor(pb[c][k], o); }
    if (rr == 0) {
#pragma unroll
        for (int c = 0; c < NV; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) { atomicAdd(&sm[4 * l + CS * c + k], pg[c][k]); atomicAdd(&sm[C + 4 * l + CS * c + k], pb[c][k]); }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 64 * NW) {
        atomic_add_f32(&ggamma[c], sm[c]);
        atomic_add_f32(&gbeta[c], sm[C + c]);
    }
}

// Wide rows (256 < C <= 1024, C % 4 == 0; FourCastNet's 768, Pangu's 384): a lane owns NV float4 groups of a row (16-byte accesses
// instead of the scalar kernel's 4-byte ones), a wave walks its rows with the next row's loads in flight, and the grid is sized to
// the resident slots (a workgroup's 2 C column partials end in float atomics on the same 2 C addresses: their number, not C, sets
// the tail -- the scalar kernel's 1013 workgroups at 16200 x 768 spent a third of its 79 us there).
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_wide_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const float* __restrict__ gy, const float* __restrict__ gadd,
                                                                 float* __restrict__ gx, float* ggamma, float* gbeta, int T, int C,
                                                                 int rows_per_block, int gy_bf16, LnLowp lp) {
    extern __shared__ float sm[];            // [2][C] per-block partial of ggamma, gbeta
    for (int c = threadIdx.x; c < 2 * C; c += 256) sm[c] = 0.f;
    __syncthreads();
    const int lane = lane_id(), w = wave_id();
    const long long row0 = (long long)blockIdx.x * rows_per_block;
    const __bf16* gh = reinterpret_cast<const __bf16*>(gy);
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 gm[NV], pg[NV], pb[NV], xv[NV], gv[NV], av[NV], xn[NV], gn[NV], an[NV];
    bool okc[NV];
    int col[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        okc[q] = 4 * lane + 256 * q < C;
        col[q] = okc[q] ? 4 * lane + 256 * q : 0;          // clamped: loads of a masked group read (and ignore) column 0
        gm[q] = okc[q] ? *reinterpret_cast<const f32x4*>(gamma + col[q]) : zero;
        pg[q] = zero; pb[q] = zero;
    }
    auto load_row = [&](long long row, f32x4 (&X)[NV], f32x4 (&G)[NV], f32x4 (&A)[NV], float& mu, float& rs) {
        const long long rc = min(row, (long long)T - 1) * C;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            X[q] = *reinterpret_cast<const f32x4*>(x + rc + col[q]);
            if (gy_bf16) {
                const bf16x4 hv = *reinterpret_cast<const bf16x4*>(gh + rc + col[q]);
                G[q] = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
            } else {
                G[q] = *reinterpret_cast<const f32x4*>(gy + rc + col[q]);
            }
            A[q] = gadd ? *reinterpret_cast<const f32x4*>(gadd + rc + col[q]) : zero;
        }
        mu = mean[min(row, (long long)T - 1)];
        rs = rstd[min(row, (long long)T - 1)];
    };
    float mu, rs, mun = 0.f, rsn = 0.f;
    load_row(row0 + w, xv, gv, av, mu, rs);
    const float invC = 1.f / C;
    for (int rr = w; rr < rows_per_block; rr += 4) {
        const long long row = row0 + rr;
        if (row >= T) break;
        const bool more = rr + 4 < rows_per_block && row + 4 < T;
        if (more) load_row(row + 4, xn, gn, an, mun, rsn);
        float s1 = 0.f, s2 = 0.f;
        f32x4 xh[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float g0 = okc[q] ? gv[q][k] : 0.f;
                xh[q][k] = okc[q] ? (xv[q][k] - mu) * rs : 0.f;
                const float gg = g0 * gm[q][k];
                s1 += gg;
                s2 += gg * xh[q][k];
                pg[q][k] += g0 * xh[q][k];
                pb[q][k] += g0;
            }
        s1 = wave_sum64(s1) * invC;
        s2 = wave_sum64(s2) * invC;
const float lsc = ln_lowp_scale(lp, row);
#pragma unroll
        for (int q = 0; q < NV; ++q)
            if (okc[q]) {
                f32x4 o4;
#pragma unroll
                for (int k = 0; k < 4; ++k) o4[k] = rs * (gv[q][k] * gm[q][k] - s1 - xh[q][k] * s2) + av[q][k];
                *reinterpret_cast<f32x4*>(gx + row * C + col[q]) = o4;
                ln_lowp_store4(lp, lsc, row * C + col[q], o4);
            }
        if (more) {
#pragma unroll
            for (int q = 0; q < NV; ++q) { xv[q] = xn[q]; gv[q] = gn[q]; av[q] = an[q]; }
            mu = mun; rs = rsn;
        }
    }
    for (int ww = 0; ww < 4; ++ww) {          // the four waves' column partials, one wave at a time (no LDS float atomics)
        if (w == ww) {
#pragma unroll
            for (int q = 0; q < NV; ++q)
                if (okc[q]) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { sm[col[q] + k] += pg[q][k]; sm[C + col[q] + k] += pb[q][k]; }
                }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        atomic_add_f32(&ggamma[c], sm[c]);
        atomic_add_f32(&gbeta[c], sm[C + c]);
    }
}

// out[n] += sum_t g[t][n].  Tall matrices (T > 16): a workgroup sums a slab of rows for 256 columns -- 16-byte loads, eight
// independent rows in flight per thread (the first version walked its rows one dependent-free but serial 4-byte load at a
// time: 28 us per call at the FourCastNet-scale shapes, 4.9 % of that step) -- and adds its partial with float atomics.
// Flat matrices (T <= 16: the batch sum behind a position embedding's gradient, N = tokens x channels in the millions): one
// thread per four columns, no atomics at all (nothing else writes `out` in that launch).
template <int VEC>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ g, float* out, int T, int N, int rows_per_block, int g_bf16 = 0) {
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 64 + lane) * VEC;
    const int row0 = blockIdx.y * rows_per_block, row1 = min(T, row0 + rows_per_block);
    __shared__ float red[4][64 * VEC];
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    if (n0 < N) {
        for (int r = row0 + part; r < row1; r += 32) {
            float v[8][VEC];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int rr = r + 4 * u;
                const long long po = (long long)(rr < row1 ? rr : r) * N + n0;
                const float* p = g + po;
                if (g_bf16) {
                    const __bf16* hp = reinterpret_cast<const __bf16*>(g) + po;
                    if (VEC == 4) {
                        const bf16x4 t = *reinterpret_cast<const bf16x4*>(hp);
#pragma unroll
                        for (int k = 0; k < VEC; ++k) v[u][k] = (float)t[k];
                    } else {
                        v[u][0] = (float)*hp;
                    }
                } else if (VEC == 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) v[u][k] = t[k];
                } else {
                    v[u][0] = *p;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (r + 4 * u < row1) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] += v[u][k];
                }
        }
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) red[part][lane * VEC + k] = acc[k];
    __syncthreads();
    if (part == 0 && n0 < N) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float t = (red[0][lane * VEC + k] + red[1][lane * VEC + k]) + (red[2][lane * VEC + k] + red[3][lane * VEC + k]);
            if (n0 + k < N) atomic_add_f32(&out[n0 + k], t);
        }
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void colsum_flat_kernel(const float* __restrict__ g, float* __restrict__ out, int T, long long N, int overwrite) {
    const long long stride = (long long)gridDim.x * 256 * VEC;
    for (long long n = ((long long)blockIdx.x * 256 + threadIdx.x) * VEC; n < N; n += stride) {
        if (VEC == 4) {
            f32x4 s = overwrite ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(out + n);
            for (int t = 0; t < T; ++t) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(g + (long long)t * N + n);
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] += v[k];
            }
            *reinterpret_cast<f32x4*>(out + n) = s;
        } else {
            float s = overwrite ? 0.f : out[n];
            for (int t = 0; t < T; ++t) s += g[(long long)t * N + n];
            out[n] = s;
        }
    }
}

}  // namespace

namespace {
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long long n) {
    const long long n4 = n / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
        reinterpret_cast<bf16x4*>(dst)[i] = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    }
    for (long long i = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = (__bf16)src[i];
}
// dst = bf16(src * scale[sample]) for `per` elements per sample (per % 4 == 0): the gradient of a DropPath branch on its way into the
// branch's bf16 backward products
__global__ __launch_bounds__(256) void cast_bf16_scaled_kernel(const float* __restrict__ src, const float* __restrict__ scale,
                                                               __bf16* __restrict__ dst, long long per4, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float sc = scale[i / per4];
        const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
        reinterpret_cast<bf16x4*>(dst)[i] = bf16x4{(__bf16)(v[0] * sc), (__bf16)(v[1] * sc), (__bf16)(v[2] * sc), (__bf16)(v[3] * sc)};
    }
}
}  // namespace

extern "C" int dlwp_cast_bf16_scaled(const float* src, const float* scale, void* dst, int nsamples, long long per_sample, void* stream) {
    DLWP_REQUIRE(src && scale && dst && nsamples >= 0 && per_sample >= 0, DLWP_E_INVALID, "cast_bf16_scaled: bad argument");
    DLWP_REQUIRE((uintptr_t)src % 16 == 0 && (uintptr_t)dst % 8 == 0 && per_sample % 4 == 0, DLWP_E_INVALID,
                 "cast_bf16_scaled: buffers must be 16 / 8 byte aligned and per_sample a multiple of 4");
    const long long n4 = (long long)nsamples * (per_sample / 4);
    if (n4 == 0) return DLWP_OK;
    const long long blocks = std::min<long long>((n4 + 255) / 256, 4096);
    hipLaunchKernelGGL(cast_bf16_scaled_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, scale, (__bf16*)dst,
                       per_sample / 4, n4);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

namespace {
// dst_m [cols][rows] (bf16) = transpose of src_m [rows][cols] (fp32) for a list of matrices that live in two flat buffers (round 6): the
// per-step TRANSPOSED bf16 copies of the Linear weights.  The input-gradient products gx = g W contract over W's ROW index -- with W as
// stored ([out][in]) that is the "[k][n]" operand form, whose fragments the 128 x 128 LDS-DMA kernel reads through two transposing LDS
// reads each (a K-step measured 2080 cycles against 1270 for the k-contiguous form at 8192 x 384 x 1536); with the transposed copy the
// product is y = g (W^T)^T, the k-contiguous form.  One launch for every matrix: grid (tiles of the largest matrix, matrices), 64 x 64
// tiles through LDS, descriptors (source offset, destination offset, rows, cols) in a device array that is built once.
__global__ __launch_bounds__(256) void transpose_cast_many_kernel(const float* __restrict__ src, __bf16* __restrict__ dst,
                                                                  const long long* __restrict__ descs) {
    __shared__ float tile[64][65];
    const long long* d = descs + 4 * blockIdx.y;
    const long long so = d[0], dofs = d[1];
    const int rows = (int)d[2], cols = (int)d[3];
    const int tc = (cols + 63) / 64, tr = (rows + 63) / 64;
    if ((int)blockIdx.x >= tc * tr) return;
    const int r0 = 64 * (blockIdx.x / tc), c0 = 64 * (blockIdx.x % tc);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty + 4 * i, c = c0 + tx;
        tile[ty + 4 * i][tx] = (r < rows && c < cols) ? src[so + (long long)r * cols + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i, r = r0 + tx;             // destination row = source column
        if (c < cols && r < rows) dst[dofs + (long long)c * rows + r] = (__bf16)tile[tx][ty + 4 * i];
    }
}
}  // namespace

extern "C" int dlwp_transpose_cast_bf16_many(const float* src_base, void* dst_base, const long long* descs, int n, int max_tiles, void* stream) {
    DLWP_REQUIRE(src_base && dst_base && descs && n >= 0 && max_tiles >= 0, DLWP_E_INVALID, "transpose_cast_bf16_many: bad argument");
    if (n == 0 || max_tiles == 0) return DLWP_OK;
    DLWP_REQUIRE(n <= 65535, DLWP_E_UNSUPPORTED, "transpose_cast_bf16_many: at most 65535 matrices (got %d)", n);
    hipLaunchKernelGGL(transpose_cast_many_kernel, dim3((unsigned)max_tiles, (unsigned)n), dim3(256), 0, (hipStream_t)stream, src_base,
                       static_cast<__bf16*>(dst_base), descs);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_cast_bf16(const float* src, void* dst, long long n, void* stream) {
    DLWP_REQUIRE(src && dst && n >= 0, DLWP_E_INVALID, "cast_bf16: bad argument");
    DLWP_REQUIRE((uintptr_t)src % 16 == 0 && (uintptr_t)dst % 8 == 0, DLWP_E_INVALID, "cast_bf16: buffers must be 16 / 8 byte aligned");
    if (n == 0) return DLWP_OK;
    const long long blocks = std::min<long long>((n / 4 + 255) / 256 + 1, 4096);
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, n);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_layernorm_fwd_ex(const float* x, const float* gamma, const float* beta, void* y, float* mean,
                                     float* rstd, int T, int C, float eps, int y_bf16, void* stream) {
    DLWP_REQUIRE(x && gamma && beta && y && mean && rstd && T > 0 && C > 0, DLWP_E_INVALID, "layernorm_fwd: bad argument");
    const bool narrow = C % 4 == 0 && C <= 256 && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 8 == 0 && (uintptr_t)gamma % 16 == 0 &&
                        (uintptr_t)beta % 16 == 0 && (y_bf16 || (uintptr_t)y % 16 == 0);
    // live accounting: x read, y written (fp32 or bf16), the two statistics written; ~8 flops per element
    const double pbytes = (double)T * C * (4 + (y_bf16 ? 2 : 4)) + 8.0 * T, pflops = 8.0 * T * C;
    const int lpr3 = C == 96 ? 8 : C == 192 ? 16 : C == 384 ? 32 : C == 768 ? 64 : 0;
    // (from 4 M elements: 65536 x 96 12.6 -> 10.3 us, 32768 x 192 12.3 -> 10.0 us on cold buffers; the smaller inputs sit at the ~ 8 us of a launch either way)
    const int fv3 = dlwp_tune_or("LN_FWD_V3", 1);
    if (lpr3 && T >= 2048 && (fv3 == 2 || (fv3 == 1 && (long long)T * C >= (4 << 20))) && (uintptr_t)x % 16 == 0 && (uintptr_t)y % (y_bf16 ? 8 : 16) == 0 &&
        (uintptr_t)gamma % 16 == 0 && (uintptr_t)beta % 16 == 0) {
        dlwp_prof_scope prof((hipStream_t)stream, pflops, pbytes, "layernorm_fwd_vecn_kernel<%d, 3>", lpr3);
#define LN_FWD_N(LPR) hipLaunchKernelGGL((layernorm_fwd_vecn_kernel<LPR, 3>), dim3(ceil_div(T, 4 * (64 / LPR))), dim3(256), 0, (hipStream_t)stream, \
                                          x, gamma, beta, (float*)y, mean, rstd, T, C, eps, y_bf16 ? 1 : 0)
        if (lpr3 == 8) LN_FWD_N(8); else if (lpr3 == 16) LN_FWD_N(16); else if (lpr3 == 32) LN_FWD_N(32); else LN_FWD_N(64);
#undef LN_FWD_N
        DLWP_LAUNCH_CHECK();
        return DLWP_OK;
    }
    if (narrow) {
        const int lpr = C <= 32 ? 8 : C <= 64 ? 16 : C <= 128 ? 32 : 64;
        dlwp_prof_scope prof((hipStream_t)stream, pflops, pbytes, "layernorm_fwd_vec_kernel<%d>", lpr);
#define LN_FWD_V(LPR) hipLaunchKernelGGL(layernorm_fwd_vec_kernel<LPR>, dim3(ceil_div(T, 4 * (64 / LPR))), dim3(256), 0, (hipStream_t)stream, \
                                          x, gamma, beta, (float*)y, mean, rstd, T, C, eps, y_bf16 ? 1 : 0)
        if (lpr == 8) LN_FWD_V(8); else if (lpr == 16) LN_FWD_V(16); else if (lpr == 32) LN_FWD_V(32); else LN_FWD_V(64);
#undef LN_FWD_V
        DLWP_LAUNCH_CHECK();
        return DLWP_OK;
    }
    const bool widef = C % 4 == 0 && C > 256 && C <= 1024 && (uintptr_t)x % 16 == 0 && (uintptr_t)y % (y_bf16 ? 8 : 16) == 0 &&
                       (uintptr_t)gamma % 16 == 0 && (uintptr_t)beta % 16 == 0;
    if (widef) {
        dlwp_prof_scope prof((hipStream_t)stream, pflops, pbytes, "layernorm_fwd_wide_kernel<%d>", C <= 512 ? 2 : C <= 768 ? 3 : 4);
#define LN_FWD_W(NV) hipLaunchKernelGGL(layernorm_fwd_wide_kernel<NV>, dim3(ceil_div(T, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, \
                                        (float*)y, mean, rstd, T, C, eps, y_bf16)
        if (C <= 512) LN_FWD_W(2); else if (C <= 768) LN_FWD_W(3); else LN_FWD_W(4);
#undef LN_FWD_W
        DLWP_LAUNCH_CHECK();
        return DLWP_OK;
    }
    dlwp_prof_scope prof((hipStream_t)stream, pflops, pbytes, "layernorm_fwd_kernel");
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(ceil_div(T, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, (float*)y,
                       mean, rstd, T, C, eps, y_bf16 ? 1 : 0);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                  float* rstd, int T, int C, float eps, void* stream) {
    return dlwp_layernorm_fwd_ex(x, gamma, beta, y, mean, rstd, T, C, eps, 0, stream);
}

// waves per workgroup of layernorm_bwd_vec_kernel (round 6).  The launch ends with 2 C float atomics per workgroup on
// the same 2 C addresses (~ 17 ns each: 8 - 10 us of a 512-workgroup launch, tools/probe_layernorm_bwd.py with the tail switched off), so
// the large inputs run the same number of waves in HALF the workgroups: 256 workgroups of eight waves, one per CU (65536 x 96: 30.4 -> 25.8 us,
// 32768 x 192: 30.8 -> 28.2 us, 16384 x 192: 22.2 -> 19.3 us back to back; Swin C4 398 -> 405 samples/s).  Sixteen waves (128 workgroups) leave
// half the CUs idle and lose; the wide-row kernel (one row per wave and step) does not gain from eight waves (8192 x 384 21.3 -> 22.3 us,
// 16200 x 768 39.6 -> 46.0 us) and stays at four.
// A grouped slab + last-workgroup sum of the partials was measured as well: the returning atomics and the ticket cost more than the tail
// they replace (profiles/r06_experiments.md section 8).  LN_BWD_NW: 4 or 8 overrides.
static int ln_bwd_waves(int T, int C) {
    const int env = dlwp_tune("LN_BWD_NW");
    if (env == 4 || env == 8) return env;
    return T >= 2048 && (long long)T * C >= (1 << 21) ? 8 : 4;
}

static int layernorm_bwd_impl(const float* x, const float* gamma, const float* mean, const float* rstd, const float* gy,
                              const float* gadd, float* gx, float* ggamma, float* gbeta, int T, int C, int gy_bf16,
                              void* stream, LnLowp lp = LnLowp{nullptr, nullptr, 1}) {
    DLWP_REQUIRE(x && gamma && mean && rstd && gy && gx && ggamma && gbeta && T > 0 && C > 0, DLWP_E_INVALID,
                 "layernorm_bwd: bad argument");
    DLWP_REQUIRE(C <= 2048, DLWP_E_UNSUPPORTED, "layernorm_bwd: C <= 2048 supported (got %d)", C);
    // Every workgroup ends with 2C float atomics on the same addresses, which serialise at ~25 ns each, so the
    // tail costs (number of workgroups) x 25 ns whatever C is: ~128-256 workgroups balance that against the rows
    // each wave walks serially.
    // each wave then walks serially (one wave per SIMD hides no latency).  Large inputs get up to ~1024 workgroups.
    const int want_env = dlwp_tune("LN_BWD_WANT");
    // (measured, tools/probe_layernorm.py: 8192 x 256 wants 256 workgroups (14.7 vs 18.7 us at 128); the 1024 x 64 calls of the 64 x 64
    // AFNO rollout want the fewer, longer ones: 128 -> 256 cost that step 4 %)
    const long long want = want_env != DLWP_TUNE_UNSET ? want_env
                                    : std::min<long long>(512, std::max<long long>((long long)T * C >= (1 << 21) ? 256 : 128, (long long)T * C / 16384));
    int rpb = 256;
    while (rpb > 4 && ceil_div(T, rpb) < want) rpb >>= 1;
    const dim3 grid(ceil_div(T, rpb));
    const size_t lds = 2 * C * sizeof(float);
    const bool narrow = C % 4 == 0 && C <= 256 && (uintptr_t)x % 16 == 0 && (uintptr_t)gy % 8 == 0 && (uintptr_t)gx % 16 == 0 &&
                        (uintptr_t)gamma % 16 == 0 && (gy_bf16 || (uintptr_t)gy % 16 == 0) && (!gadd || (uintptr_t)gadd % 16 == 0);
    const bool wide = C % 4 == 0 && C > 256 && C <= 1024 && (uintptr_t)x % 16 == 0 && (uintptr_t)gx % 16 == 0 &&
                      (uintptr_t)gamma % 16 == 0 && (uintptr_t)gy % (gy_bf16 ? 8 : 16) == 0 && (!gadd || (uintptr_t)gadd % 16 == 0);
    const bool no_wide = dlwp_tune_on("LN_BWD_NOWIDE");
    // live accounting: x, gy (fp32 or bf16) and the optional residual gradient read, gx written, the statistics read; ~14 flops / element
    const double pbytes = (double)T * C * (4 + (gy_bf16 ? 2 : 4) + (gadd ? 4 : 0) + 4) + 8.0 * T, pflops = 14.0 * T * C;
    // widths 96 / 192 / 384 / 768 (4 LPR x 3 floats per row): the three-chunk kernel, one workgroup of eight waves per CU
    // (768 -- FourCastNet -- stays on the wide-row kernel: 16200 x 768 39.5 us there, 45 us here.)  Workgroups: one per CU, one per TWO CUs
    // for inputs of at most 4 M elements -- the launch ends in (workgroups) x 17 ns of float atomics per gradient address, and these small
    // inputs are bound by that and by the launch, not by bandwidth: 16384 x 192 19.4 -> 17.0 us, 8192 x 384 23.0 -> 20.5 us
    // (tools/probe_layernorm_bwd.py; 384 workgroups and more lose everywhere)
    const int v3_env = dlwp_tune_or("LN_BWD_V3", 1);
    const int lpr3 = C == 96 ? 8 : C == 192 ? 16 : C == 384 ? 32 : (C == 768 && v3_env == 2) ? 64 : 0;
    if (lpr3 && (narrow || wide) && T >= 2048 && v3_env != 0) {
        dlwp_prof_scope prof((hipStream_t)stream, pflops, pbytes, "layernorm_bwd_vecn_kernel<%d, 3, 8, %d>", lpr3, lpr3 == 64 ? 2 : 4);
        const int rpg = 8 * (64 / lpr3);                                           // rows of one workgroup-wide row group
        const int wgs = dlwp_tune_or("LN_BWD_WGS", (long long)T * C <= (4 << 20) ? 128 : 256);
        const int rpbn = std::max(rpg, ceil_div(ceil_div(T, wgs), rpg) * rpg);
        const dim3 gridn(ceil_div(T, rpbn));
#define LN_BWD_N(LPR, UB_) hipLaunchKernelGGL((layernorm_bwd_vecn_kernel<LPR, 3, 8, UB_>), gridn, dim3(512), lds, (hipStream_t)stream, x, gamma, mean, rstd, gy, \
                                              gadd, gx, ggamma, gbeta, T, C, rpbn, gy_bf16, lp)
        if (lpr3 == 8) LN_BWD_N(8, 4); else if (lpr3 == 16) LN_BWD_N(16, 4); else if (lpr3 == 32) LN_BWD_N(32, 4); else LN_BWD_N(64, 2);
#undef LN_BWD_N
        DLWP_LAUNCH_CHECK();
        return DLWP_OK;
    }
    if (wide && !no_wide) {
        dlwp_prof_scope prof((hipStream_t)stream, pflops, pbytes, "layernorm_bwd_wide_kernel<%d>", C <= 512 ? 2 : C <= 768 ? 3 : 4);
        // one round of resident workgroups (~110 VGPRs: four per CU at most; 512-768 keep the atomic tail short)
        const int wg_env = dlwp_tune("LN_BWD_WGS");
        const int slots = wg_env != DLWP_TUNE_UNSET ? wg_env : 384;
        const int rpw = std::max(4, ceil_div(ceil_div(T, slots), 4) * 4);
        const dim3 gridw(ceil_div(T, rpw));
#define LN_BWD_W(NV) hipLaunchKernelGGL(layernorm_bwd_wide_kernel<NV>, gridw, dim3(256), lds, (hipStream_t)stream, x, gamma, mean, rstd, gy, \
                                        gadd, gx, ggamma, gbeta, T, C, rpw, gy_bf16, lp)
        if (C <= 512) LN_BWD_W(2); else if (C <= 768) LN_BWD_W(3); else LN_BWD_W(4);
#undef LN_BWD_W
        DLWP_LAUNCH_CHECK();
        return DLWP_OK;
    }
    if (narrow) {
        const int nw = ln_bwd_waves(T, C);
        dlwp_prof_scope prof((hipStream_t)stream, pflops, pbytes, "layernorm_bwd_vec_kernel<%d, %d>", C <= 32 ? 8 : C <= 64 ? 16 : C <= 128 ? 32 : 64, nw);
        const int rpbn = nw == 8 ? std::max(8, ceil_div(ceil_div(T, 256), 8) * 8) : rpb;      // eight waves: one workgroup per CU
        const dim3 gridn(ceil_div(T, rpbn));
#define LN_BWD_V2(LPR, NW_) hipLaunchKernelGGL((layernorm_bwd_vec_kernel<LPR, NW_>), gridn, dim3(64 * NW_), lds, (hipStream_t)stream, x, gamma, mean, rstd, gy, \
                                          gadd, gx, ggamma, gbeta, T, C, rpbn, gy_bf16, lp)
#define LN_BWD_V(LPR) do { if (nw == 8) LN_BWD_V2(LPR, 8); else LN_BWD_V2(LPR, 4); } while (0)
        if (C <= 32) LN_BWD_V(8); else if (C <= 64) LN_BWD_V(16); else if (C <= 128) LN_BWD_V(32); else LN_BWD_V(64);
#undef LN_BWD_V2
#undef LN_BWD_V
        DLWP_LAUNCH_CHECK();
        return DLWP_OK;
    }
#define LN_BWD(NQ)                                                                                                   \
    hipLaunchKernelGGL(layernorm_bwd_kernel<NQ>, grid, dim3(256), lds, (hipStream_t)stream, x, gamma, mean, rstd, gy, gadd, \
                       gx, ggamma, gbeta, T, C, rpb, gy_bf16, lp)
    dlwp_prof_scope prof((hipStream_t)stream, pflops, pbytes, "layernorm_bwd_kernel");
    if (C <= 64) LN_BWD(1);
    else if (C <= 128) LN_BWD(2);
    else if (C <= 256) LN_BWD(4);
    else if (C <= 512) LN_BWD(8);
    else if (C <= 1024) LN_BWD(16);
    else LN_BWD(32);
#undef LN_BWD
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_layernorm_bwd_res(const float* x, const float* gamma, const float* mean, const float* rstd,
                                      const float* gy, const float* gadd, float* gx, float* ggamma, float* gbeta, int T,
                                      int C, void* stream) {
    return layernorm_bwd_impl(x, gamma, mean, rstd, gy, gadd, gx, ggamma, gbeta, T, C, 0, stream);
}

extern "C" int dlwp_layernorm_bwd_ex(const float* x, const float* gamma, const float* mean, const float* rstd, const void* gy,
                                     int gy_bf16, const float* gadd, float* gx, float* ggamma, float* gbeta, int T, int C,
                                     void* stream) {
    return layernorm_bwd_impl(x, gamma, mean, rstd, (const float*)gy, gadd, gx, ggamma, gbeta, T, C, gy_bf16 ? 1 : 0, stream);
}

extern "C" int dlwp_layernorm_bwd_lowp(const float* x, const float* gamma, const float* mean, const float* rstd, const void* gy,
                                       int gy_bf16, const float* gadd, float* gx, float* ggamma, float* gbeta, int T, int C,
                                       void* gx_bf16, const float* row_scale, int rows_per_sample, void* stream) {
    DLWP_REQUIRE(gx_bf16 && rows_per_sample > 0 && T % rows_per_sample == 0 && (uintptr_t)gx_bf16 % 8 == 0, DLWP_E_INVALID,
                 "layernorm_bwd_lowp: a bf16 output (8-byte aligned) and whole samples of %d rows (T = %d)", rows_per_sample, T);
    return layernorm_bwd_impl(x, gamma, mean, rstd, (const float*)gy, gadd, gx, ggamma, gbeta, T, C, gy_bf16 ? 1 : 0, stream,
                              LnLowp{static_cast<__bf16*>(gx_bf16), row_scale, rows_per_sample});
}

extern "C" int dlwp_layernorm_bwd(const float* x, const float* gamma, const float* mean, const float* rstd,
                                  const float* gy, float* gx, float* ggamma, float* gbeta, int T, int C, void* stream) {
    return layernorm_bwd_impl(x, gamma, mean, rstd, gy, nullptr, gx, ggamma, gbeta, T, C, 0, stream);
}

extern "C" int dlwp_gelu_bwd(const float* z, const float* gy, float* gz, long long n, void* stream) {
    DLWP_REQUIRE(z && gy && gz && n >= 0, DLWP_E_INVALID, "gelu_bwd: NULL argument");
    if (n == 0) return DLWP_OK;
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, z, gy, gz, n);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_act_bwd(const float* z, const float* gy, float* gz, long long n, int act, float act_param, void* stream) {
    DLWP_REQUIRE(z && gy && gz && n >= 0 && act >= 1 && act <= 3, DLWP_E_INVALID, "act_bwd: bad argument");
    if (n == 0) return DLWP_OK;
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(act_bwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, z, gy, gz, n, act, act_param);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

static int colsum_impl(const float* g, float* out, int T, int N, int overwrite, void* stream, int g_bf16 = 0);
extern "C" int dlwp_colsum(const float* g, float* out, int T, int N, void* stream) { return colsum_impl(g, out, T, N, 0, stream); }
extern "C" int dlwp_colsum_ex(const float* g, float* out, int T, int N, int overwrite, void* stream) {
    return colsum_impl(g, out, T, N, overwrite ? 1 : 0, stream);
}
extern "C" int dlwp_colsum_bf16(const void* g, float* out, int T, int N, void* stream) {
    DLWP_REQUIRE(T > 16, DLWP_E_UNSUPPORTED, "colsum_bf16: the tall form only (T > 16 rows)");
    return colsum_impl(static_cast<const float*>(g), out, T, N, 0, stream, 1);
}

static int colsum_impl(const float* g, float* out, int T, int N, int overwrite, void* stream, int g_bf16) {
    DLWP_REQUIRE(g && out && T > 0 && N > 0, DLWP_E_INVALID, "colsum: bad argument");
    const bool vec = N % 4 == 0 && (uintptr_t)g % (g_bf16 ? 8 : 16) == 0 && (uintptr_t)out % 16 == 0;
    if (T <= 16) {
        const long long units = vec ? N / 4 : N;
        const int grid = (int)std::min<long long>((units + 255) / 256, 4096);
        if (vec) hipLaunchKernelGGL(colsum_flat_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, out, T, (long long)N, overwrite);
        else hipLaunchKernelGGL(colsum_flat_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, out, T, (long long)N, overwrite);
    } else {
        if (overwrite) {       // the slab kernel adds with atomics: start from zero.  A zero-fill KERNEL, not hipMemsetAsync: this path
                               // is inside captured steps (SFNO position-embedding gradient at per-GPU batch > 16) and captured
                               // memset nodes were seen writing garbage on later replays (common.hip.h)
            if (int zrc = dlwp_zero_f32(out, N, stream)) return zrc;
        }
        // row slabs sized so that the launch has ~2048 workgroups (fills the chip, bounds the atomics per column)
        const int cols = ceil_div(N, vec ? 256 : 64);
        int slabs = std::max(1, std::min(ceil_div(T, 64), ceil_div(2048, cols)));
        const int rpb = ceil_div(T, slabs);
        slabs = ceil_div(T, rpb);
        if (vec) hipLaunchKernelGGL(colsum_kernel<4>, dim3(cols, slabs), dim3(256), 0, (hipStream_t)stream, g, out, T, N, rpb, g_bf16);
        else hipLaunchKernelGGL(colsum_kernel<1>, dim3(cols, slabs), dim3(256), 0, (hipStream_t)stream, g, out, T, N, rpb, g_bf16);
    }
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}


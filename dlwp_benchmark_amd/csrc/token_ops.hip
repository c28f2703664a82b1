// Token-level building blocks shared by the AFNO / Swin / Pangu blocks: fp32 MFMA GEMM with fused
// bias / GELU / residual epilogue (nn.Linear, patch embedding, heads), LayerNorm forward/backward,
// GELU backward and column sums (bias gradients).
// Reference call sites (file:line under /root/reference/src/nsbench/models):
//   nn.Linear      fourcastnet/fourcastnet.py:44-46,233 (Mlp.fc1/fc2, head); swintransformer/
//                  swin_transformer.py:35-39,131,153,274 (Mlp, qkv, proj, PatchMerging.reduction)
//   nn.LayerNorm   fourcastnet.py:213 (eps 1e-6), swin_transformer.py:187,194 (eps 1e-5)
//   nn.GELU        fourcastnet.py:45, swin_transformer.py:37
//
// GEMM: C[M,N] = epilogue(op(A)[M,K] . op(B)[K,N]), row-major operands with leading dimensions and
// transpose flags so that the three products of a Linear layer (y = x W^T, gx = gy W, gW = gy^T x)
// use one kernel.  64x64 output tile per workgroup, K-step 16, four waves of 32x32 (2x2 MFMA
// 16x16x4 f32 blocks, permuted-k operand order), operands staged through LDS.
#include "common.cuh"
#include "dlwpmi_internal.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 16;
constexpr int LDA = BK + 4;    // As[m][k]: b128 fragment reads stay 16-byte aligned
constexpr int LDB = BN + 4;    // Bs[k][n]: 4*LDB % 32 == 16 -> conflict-free b32 reads

struct GemmDev {
    const float *A, *B, *bias, *residual;
    float *C, *preact;
    int M, N, K, lda, ldb, ldc, transA, transB, act, accumulate;
};

__global__ __launch_bounds__(256) void gemm_kernel(GemmDev a) {
    __shared__ __attribute__((aligned(16))) float As[BM * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id(), r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int wm = (w >> 1) * 32, wn = (w & 1) * 32;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < a.K; k0 += BK) {
        // stage A tile [BM][BK]: thread mapping follows the contiguous dimension of the source
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + q * 256;
            int m, k;
            if (a.transA) { m = idx % BM; k = idx / BM; } else { k = idx % BK; m = idx / BK; }
            float v = 0.f;
            if (m0 + m < a.M && k0 + k < a.K)
                v = a.transA ? a.A[(long long)(k0 + k) * a.lda + m0 + m] : a.A[(long long)(m0 + m) * a.lda + k0 + k];
            As[m * LDA + k] = v;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + q * 256;
            int k, n;
            if (a.transB) { k = idx % BK; n = idx / BK; } else { n = idx % BN; k = idx / BN; }
            float v = 0.f;
            if (n0 + n < a.N && k0 + k < a.K)
                v = a.transB ? a.B[(long long)(n0 + n) * a.ldb + k0 + k] : a.B[(long long)(k0 + k) * a.ldb + n0 + n];
            Bs[k * LDB + n] = v;
        }
        __syncthreads();
        f32x4 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f32x4*>(&As[(wm + i * 16 + r) * LDA + 4 * g]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s) bf[j][s] = Bs[(4 * g + s) * LDB + wn + j * 16 + r];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma16_chunk(af[i], bf[j], acc[i][j]);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + wm + i * 16 + 4 * g + q, n = n0 + wn + j * 16 + r;
                if (m < a.M && n < a.N) {
                    const long long o = (long long)m * a.ldc + n;
                    float v = acc[i][j][q];
                    if (a.bias) v += a.bias[n];
                    if (a.preact) a.preact[o] = v;
                    if (a.act == 1) v = gelu_f(v);
                    if (a.residual) v += a.residual[o];
                    a.C[o] = a.accumulate ? a.C[o] + v : v;
                }
            }
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
                                                            int C, float eps) {
    const int row = blockIdx.x * 4 + wave_id(), lane = lane_id();
    if (row >= T) return;
    const float* xr = x + (long long)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mu = wave_sum64(s) / C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mu; v += d * d; }
    const float rs = rsqrtf(wave_sum64(v) / C + eps);
    for (int c = lane; c < C; c += 64) y[(long long)row * C + c] = (xr[c] - mu) * rs * gamma[c] + beta[c];
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// gx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat)); ggamma/gbeta partials via atomics
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gy, float* __restrict__ gx,
                                                            float* ggamma, float* gbeta, int T, int C, int rows_per_block) {
    extern __shared__ float sm[];            // [2][C] per-block partial of ggamma, gbeta
    float* sg = sm;
    float* sb = sm + C;
    for (int c = threadIdx.x; c < 2 * C; c += 256) sm[c] = 0.f;
    __syncthreads();
    const int lane = lane_id(), w = wave_id();
    const int row0 = blockIdx.x * rows_per_block;
    // each wave owns the columns c = lane, lane+64, ... for its rows; accumulate column partials in registers
    // (C <= 64*8 assumed by the host wrapper), then add them to LDS once per wave
    float pg[8], pb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { pg[q] = 0.f; pb[q] = 0.f; }
    for (int rr = w; rr < rows_per_block; rr += 4) {
        const int row = row0 + rr;
        if (row >= T) break;
        const float mu = mean[row], rs = rstd[row];
        const float* xr = x + (long long)row * C;
        const float* gr = gy + (long long)row * C;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c = lane + 64 * q;
            if (c < C) {
                const float xh = (xr[c] - mu) * rs, gg = gr[c] * gamma[c];
                s1 += gg;
                s2 += gg * xh;
                pg[q] += gr[c] * xh;
                pb[q] += gr[c];
            }
        }
        s1 = wave_sum64(s1) / C;
        s2 = wave_sum64(s2) / C;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c = lane + 64 * q;
            if (c < C) {
                const float xh = (xr[c] - mu) * rs;
                gx[(long long)row * C + c] = rs * (gr[c] * gamma[c] - s1 - xh * s2);
            }
        }
    }
    // combine the four waves' column partials one wave at a time (no LDS float atomics)
    for (int ww = 0; ww < 4; ++ww) {
        if (w == ww) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int c = lane + 64 * q;
                if (c < C) { sg[c] += pg[q]; sb[c] += pb[q]; }
            }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        atomic_add_f32(&ggamma[c], sg[c]);
        atomic_add_f32(&gbeta[c], sb[c]);
    }
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ z, const float* __restrict__ gy,
                                                       float* __restrict__ gz, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) gz[i] = gy[i] * gelu_grad_f(z[i]);
}

// out[n] += sum_t g[t][n]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ g, float* out, int T, int N, int rows_per_block) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    const int row0 = blockIdx.y * rows_per_block;
    __shared__ float red[4][64];
    float s = 0.f;
    if (n < N)
        for (int rr = part; rr < rows_per_block; rr += 4) {
            const int row = row0 + rr;
            if (row < T) s += g[(long long)row * N + n];
        }
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && n < N) atomic_add_f32(&out[n], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

}  // namespace

extern "C" int dlwp_gemm(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                         int transA, int transB, const float* bias, int act, float* preact, const float* residual,
                         int accumulate, void* stream) {
    DLWP_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, DLWP_E_INVALID, "gemm: NULL argument or empty shape");
    DLWP_REQUIRE(act == 0 || act == 1, DLWP_E_INVALID, "gemm: act must be 0 (none) or 1 (gelu)");
    GemmDev a{A, B, bias, residual, C, preact, M, N, K, lda, ldb, ldc, transA, transB, act, accumulate};
    hipLaunchKernelGGL(gemm_kernel, dim3(ceil_div(N, BN), ceil_div(M, BM)), dim3(256), 0, (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                  float* rstd, int T, int C, float eps, void* stream) {
    DLWP_REQUIRE(x && gamma && beta && y && mean && rstd && T > 0 && C > 0, DLWP_E_INVALID, "layernorm_fwd: bad argument");
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(ceil_div(T, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                       mean, rstd, T, C, eps);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_layernorm_bwd(const float* x, const float* gamma, const float* mean, const float* rstd,
                                  const float* gy, float* gx, float* ggamma, float* gbeta, int T, int C, void* stream) {
    DLWP_REQUIRE(x && gamma && mean && rstd && gy && gx && ggamma && gbeta && T > 0 && C > 0, DLWP_E_INVALID,
                 "layernorm_bwd: bad argument");
    DLWP_REQUIRE(C <= 512, DLWP_E_UNSUPPORTED, "layernorm_bwd: C <= 512 supported (got %d)", C);
    const int rpb = 64;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(ceil_div(T, rpb)), dim3(256), 2 * C * sizeof(float), (hipStream_t)stream,
                       x, gamma, mean, rstd, gy, gx, ggamma, gbeta, T, C, rpb);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
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

extern "C" int dlwp_colsum(const float* g, float* out, int T, int N, void* stream) {
    DLWP_REQUIRE(g && out && T > 0 && N > 0, DLWP_E_INVALID, "colsum: bad argument");
    const int rpb = 256;
    hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(N, 64), ceil_div(T, rpb)), dim3(256), 0, (hipStream_t)stream, g, out,
                       T, N, rpb);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

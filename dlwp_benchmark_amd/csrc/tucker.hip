// Complex mode-n products for Tucker-factorised FNO weights (TFNO): W = core x1 U_i x2 U_o x3 U_x x4 U_y.
// Reference call sites: neuralop.models.TFNO built at src/dlwpbench/models/fno/fno.py:136-146 (rank from
// configs/model/fno.yaml:11) — third-party arithmetic (tltorch TuckerTensor), SURVEY.md App. A-1: parity
// unpinned.  The factorised weight is expanded once per optimizer step into the dense mode-major layout
// the spectral kernels consume; these products are tiny (<= a few MFLOP), one thread per output element.
#include "common.hip.h"
#include "dlwpmi_internal.h"

namespace {

// out[o][n][i] = sum_r in[o][r][i] * op(U)[n][r];  op(U)[n][r] = U[n*ldu_n + r*ldu_r], optionally conjugated
__global__ __launch_bounds__(256) void cmode_kernel(const float2* __restrict__ in, const float2* __restrict__ U,
                                                    float2* __restrict__ out, int O, int R, int N, int I, int ldu_n,
                                                    int ldu_r, int conj_u) {
    const long long total = (long long)O * N * I;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int i = (int)(e % I), n = (int)((e / I) % N), o = (int)(e / ((long long)I * N));
        const float2* src = in + ((long long)o * R) * I + i;
        float re = 0.f, im = 0.f;
        for (int r = 0; r < R; ++r) {
            const float2 v = src[(long long)r * I];
            float2 u = U[n * ldu_n + r * ldu_r];
            if (conj_u) u.y = -u.y;
            re += v.x * u.x - v.y * u.y;
            im += v.x * u.y + v.y * u.x;
        }
        out[e] = make_float2(re, im);
    }
}

// gU[n][r] = sum_{o,i} gout[o][n][i] * conj(in[o][r][i])      (one workgroup per (n, r))
__global__ __launch_bounds__(256) void cmode_gradu_kernel(const float2* __restrict__ in, const float2* __restrict__ gout,
                                                          float2* __restrict__ gU, int O, int R, int N, int I) {
    const int n = blockIdx.x / R, r = blockIdx.x % R;
    float re = 0.f, im = 0.f;
    const long long total = (long long)O * I;
    for (long long e = threadIdx.x; e < total; e += 256) {
        const int i = (int)(e % I), o = (int)(e / I);
        const float2 g = gout[((long long)o * N + n) * I + i];
        const float2 v = in[((long long)o * R + r) * I + i];
        re += g.x * v.x + g.y * v.y;
        im += g.y * v.x - g.x * v.y;
    }
    __shared__ float2 part[4];
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) { re += __shfl_xor(re, s); im += __shfl_xor(im, s); }
    if (lane_id() == 0) part[wave_id()] = make_float2(re, im);
    __syncthreads();
    if (threadIdx.x == 0)
        gU[blockIdx.x] = make_float2(part[0].x + part[1].x + part[2].x + part[3].x,
                                     part[0].y + part[1].y + part[2].y + part[3].y);
}

}  // namespace

// out[O,N,I] = in[O,R,I] x_mode U[N,R]      (complex, interleaved re/im)
extern "C" int dlwp_cmode_product(const float* in, const float* U, float* out, int O, int R, int N, int I, void* stream) {
    DLWP_REQUIRE(in && U && out && O > 0 && R > 0 && N > 0 && I > 0, DLWP_E_INVALID, "cmode_product: bad argument");
    const long long total = (long long)O * N * I;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cmode_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float2*>(in),
                       reinterpret_cast<const float2*>(U), reinterpret_cast<float2*>(out), O, R, N, I, R, 1, 0);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

// backward of dlwp_cmode_product: gin[O,R,I] = gout x_mode conj(U)^T ; gU[N,R] = sum gout conj(in)
extern "C" int dlwp_cmode_product_bwd(const float* in, const float* U, const float* gout, float* gin, float* gU, int O,
                                      int R, int N, int I, void* stream) {
    DLWP_REQUIRE(in && U && gout && gin && gU && O > 0 && R > 0 && N > 0 && I > 0, DLWP_E_INVALID,
                 "cmode_product_bwd: bad argument");
    const long long total = (long long)O * R * I;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    // gin[o][r][i] = sum_n gout[o][n][i] conj(U[n][r]): the "U" of this product is indexed [r][n] -> strides (1, R)
    hipLaunchKernelGGL(cmode_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float2*>(gout),
                       reinterpret_cast<const float2*>(U), reinterpret_cast<float2*>(gin), O, N, R, I, 1, R, 1);
    hipLaunchKernelGGL(cmode_gradu_kernel, dim3(N * R), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float2*>(in), reinterpret_cast<const float2*>(gout),
                       reinterpret_cast<float2*>(gU), O, R, N, I);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

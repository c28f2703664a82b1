// Spherical (SFNO) spectral convolution helpers.
// Reference call site: torch_harmonics SphericalFourierNeuralOperatorNet constructed at
// src/dlwpbench/models/fno/fno.py:183-200 and models/fourcastnet/fourcastnet.py:411-428 (third-party package,
// not vendored: SURVEY.md App. A-2).  Its "driscoll-healy" spectral layer multiplies every spherical-harmonic
// coefficient X[b, i, l, m] by a complex weight that depends on the degree l only:
//     Y[b, o, l, m] = sum_i X[b, i, l, m] * W[i, o, l]            ("bixy,iox->boxy")
//
// MI355X mapping: the spectral tensor is kept degree-major and channels-last, X[l][(b, m)][re|im][C], so that the
// contraction is ONE strided-batched real GEMM per call (batch = l, M = B * mmax rows, K = N = 2C) on the fp32 MFMA
// GEMM of token_ops.hip, against the real 2C x 2C image of the complex weight
//     [ Wr   Wi ]
//     [-Wi   Wr ]
// built by cweight_expand below (once per call: 2 * (2C)^2 * L floats).  The weight gradient comes back from the
// GEMM in the same 2C x 2C form and cweight_fold reduces it to the complex parameter's gradient.
#include "common.hip.h"
#include "dlwpmi_internal.h"

namespace {

// Both kernels move between the parameter's layout [Cin][Cout][L][2] (degree fastest) and the GEMM's [L][2Cin][2Cout]
// (output channel fastest): a workgroup owns one input channel i and 64 output channels, i.e. a contiguous 64 x L x 2 slab of
// the parameter, and transposes through LDS so that both sides are accessed in whole 256-byte rows (the direct form
// touched the parameter with a 8-byte access every 2 L floats: 52 us per fold at C = 256, L = 32).
constexpr int OT = 64;        // output channels per workgroup
constexpr int LT = 32;        // degrees per LDS pass

// w: [Cin][Cout][L][2] (view_as_real of the complex parameter) -> wexp: [L][2Cin][2Cout]
__global__ __launch_bounds__(256) void cweight_expand_kernel(const float* __restrict__ w, float* __restrict__ wexp, int Cin,
                                                             int Cout, int L) {
    __shared__ float2 tile[OT][LT + 1];
    const int i = blockIdx.y, o0 = blockIdx.x * OT, tid = threadIdx.x;
    const int no = min(OT, Cout - o0);
    for (int l0 = 0; l0 < L; l0 += LT) {
        const int nl = min(LT, L - l0);
        // parameter slab rows (o) of nl complex values: consecutive threads read consecutive degrees
        for (int e = tid; e < no * nl; e += 256) {
            const int o = e / nl, l = e - o * nl;
            tile[o][l] = *reinterpret_cast<const float2*>(&w[(((long long)i * Cout + o0 + o) * L + l0 + l) * 2]);
        }
        __syncthreads();
        for (int e = tid; e < nl * no; e += 256) {
            const int l = e / no, o = e - l * no;
            const float2 v = tile[o][l];
            float* base = wexp + (long long)(l0 + l) * 4 * Cin * Cout;
            base[(long long)i * 2 * Cout + o0 + o] = v.x;
            base[(long long)i * 2 * Cout + Cout + o0 + o] = v.y;
            base[(long long)(Cin + i) * 2 * Cout + o0 + o] = -v.y;
            base[(long long)(Cin + i) * 2 * Cout + Cout + o0 + o] = v.x;
        }
        __syncthreads();
    }
}

// gw[i][o][l] += (G[l][i][o] + G[l][Cin+i][Cout+o],  G[l][i][Cout+o] - G[l][Cin+i][o])
__global__ __launch_bounds__(256) void cweight_fold_kernel(const float* __restrict__ g, float* __restrict__ gw, int Cin, int Cout,
                                                           int L) {
    __shared__ float2 tile[OT][LT + 1];
    const int i = blockIdx.y, o0 = blockIdx.x * OT, tid = threadIdx.x;
    const int no = min(OT, Cout - o0);
    for (int l0 = 0; l0 < L; l0 += LT) {
        const int nl = min(LT, L - l0);
        for (int e = tid; e < nl * no; e += 256) {
            const int l = e / no, o = e - l * no;
            const float* base = g + (long long)(l0 + l) * 4 * Cin * Cout;
            const float a = base[(long long)i * 2 * Cout + o0 + o], b = base[(long long)i * 2 * Cout + Cout + o0 + o];
            const float c = base[(long long)(Cin + i) * 2 * Cout + o0 + o], d = base[(long long)(Cin + i) * 2 * Cout + Cout + o0 + o];
            tile[o][l] = make_float2(a + d, b - c);
        }
        __syncthreads();
        for (int e = tid; e < no * nl; e += 256) {
            const int o = e / nl, l = e - o * nl;
            float2* dst = reinterpret_cast<float2*>(&gw[(((long long)i * Cout + o0 + o) * L + l0 + l) * 2]);
            float2 v = *dst;
            const float2 t = tile[o][l];
            v.x += t.x;
            v.y += t.y;
            *dst = v;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int dlwp_cweight_expand(const float* w, float* wexp, int Cin, int Cout, int L, void* stream) {
    DLWP_REQUIRE(w && wexp && Cin > 0 && Cout > 0 && L > 0, DLWP_E_INVALID, "cweight_expand: bad argument");
    DLWP_REQUIRE(Cin <= 65535, DLWP_E_UNSUPPORTED, "cweight_expand: more than 65535 input channels");
    hipLaunchKernelGGL(cweight_expand_kernel, dim3(ceil_div(Cout, OT), Cin), dim3(256), 0, (hipStream_t)stream, w, wexp, Cin, Cout,
                       L);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_cweight_fold(const float* gexp, float* gw, int Cin, int Cout, int L, void* stream) {
    DLWP_REQUIRE(gexp && gw && Cin > 0 && Cout > 0 && L > 0, DLWP_E_INVALID, "cweight_fold: bad argument");
    DLWP_REQUIRE(Cin <= 65535, DLWP_E_UNSUPPORTED, "cweight_fold: more than 65535 input channels");
    hipLaunchKernelGGL(cweight_fold_kernel, dim3(ceil_div(Cout, OT), Cin), dim3(256), 0, (hipStream_t)stream, gexp, gw, Cin, Cout,
                       L);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

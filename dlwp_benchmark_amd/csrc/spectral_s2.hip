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
#include "common.cuh"
#include "dlwpmi_internal.h"

namespace {

// w: [Cin][Cout][L][2] (view_as_real of the complex parameter) -> wexp: [L][2Cin][2Cout]
__global__ __launch_bounds__(256) void cweight_expand_kernel(const float* __restrict__ w, float* __restrict__ wexp, int Cin,
                                                             int Cout, int L) {
    const long long n = (long long)Cin * Cout * L;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        // e = (l * Cin + i) * Cout + o : consecutive threads write consecutive o
        const int o = (int)(e % Cout);
        const long long t = e / Cout;
        const int i = (int)(t % Cin), l = (int)(t / Cin);
        const float2 v = *reinterpret_cast<const float2*>(&w[(((long long)i * Cout + o) * L + l) * 2]);
        float* base = wexp + (long long)l * 4 * Cin * Cout;
        base[(long long)i * 2 * Cout + o] = v.x;
        base[(long long)i * 2 * Cout + Cout + o] = v.y;
        base[(long long)(Cin + i) * 2 * Cout + o] = -v.y;
        base[(long long)(Cin + i) * 2 * Cout + Cout + o] = v.x;
    }
}

// gw[i][o][l] += (G[l][i][o] + G[l][Cin+i][Cout+o],  G[l][i][Cout+o] - G[l][Cin+i][o])
__global__ __launch_bounds__(256) void cweight_fold_kernel(const float* __restrict__ g, float* __restrict__ gw, int Cin, int Cout,
                                                           int L) {
    const long long n = (long long)Cin * Cout * L;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int o = (int)(e % Cout);
        const long long t = e / Cout;
        const int i = (int)(t % Cin), l = (int)(t / Cin);
        const float* base = g + (long long)l * 4 * Cin * Cout;
        const float a = base[(long long)i * 2 * Cout + o], b = base[(long long)i * 2 * Cout + Cout + o];
        const float c = base[(long long)(Cin + i) * 2 * Cout + o], d = base[(long long)(Cin + i) * 2 * Cout + Cout + o];
        float2* dst = reinterpret_cast<float2*>(&gw[(((long long)i * Cout + o) * L + l) * 2]);
        float2 v = *dst;
        v.x += a + d;
        v.y += b - c;
        *dst = v;
    }
}

int grid_for(long long n) { return (int)std::min<long long>((n + 255) / 256, 4096); }

}  // namespace

extern "C" int dlwp_cweight_expand(const float* w, float* wexp, int Cin, int Cout, int L, void* stream) {
    DLWP_REQUIRE(w && wexp && Cin > 0 && Cout > 0 && L > 0, DLWP_E_INVALID, "cweight_expand: bad argument");
    hipLaunchKernelGGL(cweight_expand_kernel, dim3(grid_for((long long)Cin * Cout * L)), dim3(256), 0, (hipStream_t)stream, w,
                       wexp, Cin, Cout, L);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_cweight_fold(const float* gexp, float* gw, int Cin, int Cout, int L, void* stream) {
    DLWP_REQUIRE(gexp && gw && Cin > 0 && Cout > 0 && L > 0, DLWP_E_INVALID, "cweight_fold: bad argument");
    hipLaunchKernelGGL(cweight_fold_kernel, dim3(grid_for((long long)Cin * Cout * L)), dim3(256), 0, (hipStream_t)stream, gexp,
                       gw, Cin, Cout, L);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

// W-axis pruned DFT of one image row held in LDS (shared by the FNO row / spatial kernels and by the lifting MLP, which
// appends it to its epilogue so that the first spectral block needs no separate rows launch).
#pragma once
#include "common.hip.h"

// x1s[wave][c][n] = partial over the wave's 16-pixel chunks (w, w+4, ...) of sum_w act(tile[c][w]) * ft[n][w];
// store_x1 sums the four per-wave partials (no LDS float atomics: they run at <1 lane-op/clk/CU).
template <int NCB, int NBN>
__device__ __forceinline__ void tile_rows_dft(const float* tile, const float* ft, float* x1s, int LDP,
                                              int NP, int nwb, bool act) {
    const int lane = lane_id(), w = wave_id(), r = lane & 15, g = lane >> 4;
    f32x4 xacc[NCB][NBN];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int nb = 0; nb < NBN; ++nb) xacc[cb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kc = w; kc < nwb; kc += 4) {
        f32x4 b4[NBN];
#pragma unroll
        for (int nb = 0; nb < NBN; ++nb)
            b4[nb] = *reinterpret_cast<const f32x4*>(&ft[(nb * 16 + r) * LDP + kc * 16 + 4 * g]);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            f32x4 a4 = *reinterpret_cast<const f32x4*>(&tile[(cb * 16 + r) * LDP + kc * 16 + 4 * g]);
            if (act) {
                a4 = gelu4(a4);
            }
#pragma unroll
            for (int nb = 0; nb < NBN; ++nb) xacc[cb][nb] = mfma16_chunk(a4, b4[nb], xacc[cb][nb]);
        }
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int nb = 0; nb < NBN; ++nb)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                x1s[((w * NCB + cb) * 16 + 4 * g + j) * NP + nb * 16 + r] = xacc[cb][nb][j];
}

__device__ __forceinline__ void store_x1(const float* x1s, float2* x1_out, int b, int h, int H, int m2c,
                                         int C, int C_pad, int NP) {
    float2* dst = x1_out + ((long long)(b * H + h) * m2c) * C;
    for (int idx = threadIdx.x; idx < m2c * C; idx += blockDim.x) {
        const int kx = idx / C, c = idx % C;
        float re = 0.f, im = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            re += x1s[(w * C_pad + c) * NP + 2 * kx];
            im += x1s[(w * C_pad + c) * NP + 2 * kx + 1];
        }
        dst[idx] = make_float2(re, im);
    }
}


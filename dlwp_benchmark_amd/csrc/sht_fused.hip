// Fused real spherical harmonic transforms for channels-last fields: the longitude DFT and the Legendre
// transform of one transform run in ONE launch, with the intermediate (latitude x order) plane kept in LDS.
//
// Reference semantics: torch_harmonics.RealSHT / InverseRealSHT (third party, constructed at
// src/dlwpbench/models/fno/fno.py:183-200 and models/fourcastnet/fourcastnet.py:411-428).  The two
// stages and their tables are exactly those of dlwp_benchmark_amd/sht.py's two strided-batched GEMMs (which remain the
// general-shape path); this file only changes where the intermediate lives.
//
//   analysis   X[l][b][m][ri][c] = sum_k A2[m][l][k] * T[k][2m+ri][c],   T[k][q][c] = sum_n A1[q][n] * x[b][k][n][c]
//   synthesis  x[b][k][n][c]     = sum_q S2[q][n]   * T[k][q][c],        T[k][2m+ri][c] = sum_l S1[m][l][k] * X[l][b][m][ri][c]
//
// RealSHT = analysis(A1 = F, A2 = quadrature-weighted Legendre), InverseRealSHT = synthesis(S1 = Legendre, S2 = G^T); each
// backward pass is the other kernel with the transposed tables (the adjoint of analysis(A1, A2) is synthesis(S1 = A2,
// S2 = A1)).  The synthesis entry takes its tables pre-transposed (S1t [M][K][L], S2t [N][2M]) so that both kernels
// stage their tables with linear 16-byte copies.
//
// MI355X mapping (exact-f32 MFMA 16x16x4): a workgroup owns 16 channels of one sample and, in analysis, 8 orders m (all
// latitudes: stage 2 contracts over k), in synthesis, 8 latitudes k (all orders: stage 2 contracts over q) -- 256
// workgroups at the C3 shape (B=4, 32x64, C=256).  Each wave streams its own 4 KB field tiles (next tile's 16-byte loads
// in flight during the current tile's MFMAs, wave-private LDS: no workgroup barrier inside a stage).
#include "common.cuh"
#include "dlwpmi_internal.h"

namespace {

constexpr int CB = 16;          // channels per workgroup
constexpr int GRP = 8;          // orders (analysis) / latitudes (synthesis) per workgroup
constexpr int XLD = CB + 4;     // row stride of [row][16 channel] tiles (lanes l / l+16 of a half-wave on different banks)

struct ShtDev {
    const float* in;            // analysis: x [B][K][N][C];  synthesis: X [L][B][M][2][C]
    float* out;                 // analysis: X;               synthesis: x
    const float* T1;            // analysis: A1 [2M][N];      synthesis: S1t [M][K][L]
    const float* T2;            // analysis: A2 [M][L][K];    synthesis: S2t [N][2M]
    int B, K, N, C, M, L;
    int Kp, Lp, Np, Qp;         // padded to 16 (zero-filled in LDS)
};

__device__ __forceinline__ int pad16(int v) { return (v + 15) & ~15; }

// ---------------------------------------------------------------------------------------------------------------------
// analysis: grid (C/16, M/8, B)
__global__ __launch_bounds__(256) void sht_analysis_kernel(ShtDev a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NLD = a.Np + 4, KLD = a.Kp + 4;
    float* A1s = smem;                                   // [16 q][NLD]
    float* A2s = A1s + 16 * NLD;                         // [GRP][Lp][KLD]
    float* Ts = A2s + GRP * a.Lp * KLD;                  // [Kp][16 q][XLD]
    float* xs = Ts + a.Kp * 16 * XLD;                    // [4 waves][Np][XLD]
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id(), r = lane & 15, g = lane >> 4;
    const int c0 = blockIdx.x * CB, m0 = blockIdx.y * GRP, b = blockIdx.z;
    float* xw = xs + w * a.Np * XLD;

    // this wave's latitudes: k = w, w+4, ...; a tile is x[b][k][0:N][c0:c0+16] (N rows of 64 bytes), 4 rows per 16 lanes
    const long long xrow = a.C;
    const float* xb = a.in + (long long)b * a.K * a.N * a.C + c0;
    const int nld4 = (a.N * (CB / 4) + 63) / 64;         // float4 per lane and tile (N = 64 -> 4)
    float4 pre[8];
    auto issue = [&](int k) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = lane + 64 * u, n = idx >> 2, c4 = idx & 3;
            pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < nld4 && n < a.N && k < a.K)
                pre[u] = *reinterpret_cast<const float4*>(xb + ((long long)k * a.N + n) * xrow + 4 * c4);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = lane + 64 * u, n = idx >> 2, c4 = idx & 3;
            if (u < nld4 && n < a.Np) *reinterpret_cast<float4*>(&xw[n * XLD + 4 * c4]) = pre[u];
        }
    };
    issue(w);
    // tables: A1 rows q = 2 m0 .. 2 m0 + 15 ([16][N] contiguous), A2 [GRP][L][K] contiguous from m0
    for (int e = tid; e < 16 * a.Np; e += 256) {
        const int q = e / a.Np, n = e - q * a.Np;
        A1s[q * NLD + n] = n < a.N ? a.T1[(long long)(2 * m0 + q) * a.N + n] : 0.f;
    }
    for (int e = tid; e < GRP * a.Lp * a.Kp; e += 256) {
        const int ml = e / (a.Lp * a.Kp), rem = e - ml * a.Lp * a.Kp, l = rem / a.Kp, k = rem - l * a.Kp;
        A2s[(ml * a.Lp + l) * KLD + k] = (l < a.L && k < a.K) ? a.T2[((long long)(m0 + ml) * a.L + l) * a.K + k] : 0.f;
    }
    for (int e = tid; e < (a.Kp - a.K) * 16 * XLD; e += 256) Ts[a.K * 16 * XLD + e] = 0.f;      // padded latitudes
    __syncthreads();

    // ---- stage 1: T[k][q][c] = sum_n A1[q][n] x[k][n][c]
    for (int k = w; k < a.K; k += 4) {
        commit();
        if (k + 4 < a.K) issue(k + 4);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int n0 = 0; n0 < a.Np; n0 += 16) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(&A1s[r * NLD + n0 + 4 * g]);
            f32x4 b4;
#pragma unroll
            for (int s = 0; s < 4; ++s) b4[s] = xw[(n0 + 4 * g + s) * XLD + r];
            acc = mfma16_chunk(a4, b4, acc);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) Ts[(k * 16 + 4 * g + j) * XLD + r] = acc[j];
    }
    __syncthreads();

    // ---- stage 2: X[l][m][ri][c] = sum_k A2[m][l][k] T[k][2m+ri][c]
    for (int ml = w; ml < GRP; ml += 4) {
        const int m = m0 + ml;
        if (m >= a.M) break;
        for (int lt = 0; lt < a.Lp / 16; ++lt) {
            f32x4 acc[2];
            acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[1] = acc[0];
            for (int k0 = 0; k0 < a.Kp; k0 += 16) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(&A2s[(ml * a.Lp + lt * 16 + r) * KLD + k0 + 4 * g]);
#pragma unroll
                for (int ri = 0; ri < 2; ++ri) {
                    f32x4 b4;
#pragma unroll
                    for (int s = 0; s < 4; ++s) b4[s] = Ts[((k0 + 4 * g + s) * 16 + 2 * ml + ri) * XLD + r];
                    acc[ri] = mfma16_chunk(a4, b4, acc[ri]);
                }
            }
#pragma unroll
            for (int ri = 0; ri < 2; ++ri)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int l = lt * 16 + 4 * g + j;
                    if (l < a.L) a.out[((((long long)l * a.B + b) * a.M + m) * 2 + ri) * a.C + c0 + r] = acc[ri][j];
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// synthesis: grid (C/16, ceil(K/8), B)
__global__ __launch_bounds__(256) void sht_synthesis_kernel(ShtDev a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Q = 2 * a.M;
    const int LLD = a.Lp + 4, QLD = a.Qp + 4;
    float* S1s = smem;                                   // [M][GRP][LLD]    S1s[m][kl][l] = S1[m][l][k0 + kl]
    float* S2s = S1s + a.M * GRP * LLD;                  // [Np][QLD]        S2s[n][q]     = S2[q][n]
    float* Ts = S2s + a.Np * QLD;                        // [GRP][Qp][XLD]
    float* xs = Ts + GRP * a.Qp * XLD;                   // [4 waves][2 Lp][XLD]   rows (l, ri)
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id(), r = lane & 15, g = lane >> 4;
    const int c0 = blockIdx.x * CB, k0g = blockIdx.y * GRP, b = blockIdx.z;
    float* xw = xs + w * 2 * a.Lp * XLD;

    // this wave's orders: m = w, w+4, ...; a tile is X[0:L][b][m][0:2][c0:c0+16]: 2 L rows of 64 bytes
    const float* Xb = a.in + (((long long)b * a.M) * 2) * a.C + c0;
    const long long lstride = (long long)a.B * a.M * 2 * a.C;
    const int nld4 = (2 * a.L * (CB / 4) + 63) / 64;     // L = 32 -> 4
    float4 pre[8];
    auto issue = [&](int m) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = lane + 64 * u, row = idx >> 2, c4 = idx & 3, l = row >> 1, ri = row & 1;
            pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < nld4 && l < a.L && m < a.M)
                pre[u] = *reinterpret_cast<const float4*>(Xb + l * lstride + ((long long)m * 2 + ri) * a.C + 4 * c4);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = lane + 64 * u, row = idx >> 2, c4 = idx & 3;
            if (u < nld4 && row < 2 * a.Lp) *reinterpret_cast<float4*>(&xw[row * XLD + 4 * c4]) = pre[u];
        }
    };
    issue(w);
    for (int e = tid; e < a.M * GRP * a.Lp; e += 256) {
        const int m = e / (GRP * a.Lp), rem = e - m * GRP * a.Lp, kl = rem / a.Lp, l = rem - kl * a.Lp;
        const int k = k0g + kl;
        S1s[(m * GRP + kl) * LLD + l] = (l < a.L && k < a.K) ? a.T1[((long long)m * a.K + k) * a.L + l] : 0.f;
    }
    for (int e = tid; e < a.Np * a.Qp; e += 256) {
        const int n = e / a.Qp, q = e - n * a.Qp;
        S2s[n * QLD + q] = (n < a.N && q < Q) ? a.T2[(long long)n * Q + q] : 0.f;
    }
    for (int e = tid; e < GRP * (a.Qp - Q) * XLD; e += 256) {       // padded orders
        const int kl = e / ((a.Qp - Q) * XLD), rem = e - kl * (a.Qp - Q) * XLD;
        Ts[(kl * a.Qp + Q) * XLD + rem] = 0.f;
    }
    // rows of the wave tile beyond 2 L stay zero for the whole kernel
    for (int e = lane; e < (2 * a.Lp - 2 * a.L) * XLD; e += 64) xw[2 * a.L * XLD + e] = 0.f;
    __syncthreads();

    // ---- stage 1: T[kl][2m+ri][c] = sum_l S1[m][l][k0g+kl] X[l][m][ri][c]   (rows kl < 8 of the 16-row tile are real)
    for (int m = w; m < a.M; m += 4) {
        commit();
        if (m + 4 < a.M) issue(m + 4);
        f32x4 acc[2];
        acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[1] = acc[0];
        for (int l0 = 0; l0 < a.Lp; l0 += 16) {
            f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
            if (r < GRP) a4 = *reinterpret_cast<const f32x4*>(&S1s[(m * GRP + r) * LLD + l0 + 4 * g]);
#pragma unroll
            for (int ri = 0; ri < 2; ++ri) {
                f32x4 b4;
#pragma unroll
                for (int s = 0; s < 4; ++s) b4[s] = xw[((l0 + 4 * g + s) * 2 + ri) * XLD + r];
                acc[ri] = mfma16_chunk(a4, b4, acc[ri]);
            }
        }
        if (g < GRP / 4) {
#pragma unroll
            for (int ri = 0; ri < 2; ++ri)
#pragma unroll
                for (int j = 0; j < 4; ++j) Ts[((4 * g + j) * a.Qp + 2 * m + ri) * XLD + r] = acc[ri][j];
        }
    }
    __syncthreads();

    // ---- stage 2: x[k][n][c] = sum_q S2[q][n] T[kl][q][c]
    for (int kl = w; kl < GRP; kl += 4) {
        const int k = k0g + kl;
        if (k >= a.K) break;
        for (int nt = 0; nt < a.Np / 16; ++nt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int q0 = 0; q0 < a.Qp; q0 += 16) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(&S2s[(nt * 16 + r) * QLD + q0 + 4 * g]);
                f32x4 b4;
#pragma unroll
                for (int s = 0; s < 4; ++s) b4[s] = Ts[(kl * a.Qp + q0 + 4 * g + s) * XLD + r];
                acc = mfma16_chunk(a4, b4, acc);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = nt * 16 + 4 * g + j;
                if (n < a.N) a.out[(((long long)b * a.K + k) * a.N + n) * a.C + c0 + r] = acc[j];
            }
        }
    }
}

size_t analysis_lds(int K, int N, int L) {
    const int Kp = round_up(K, 16), Np = round_up(N, 16), Lp = round_up(L, 16);
    return sizeof(float) * ((size_t)16 * (Np + 4) + (size_t)GRP * Lp * (Kp + 4) + (size_t)Kp * 16 * XLD + (size_t)4 * Np * XLD);
}
size_t synthesis_lds(int N, int M, int L) {
    const int Np = round_up(N, 16), Lp = round_up(L, 16), Qp = round_up(2 * M, 16);
    return sizeof(float) * ((size_t)M * GRP * (Lp + 4) + (size_t)Np * (Qp + 4) + (size_t)GRP * Qp * XLD + (size_t)4 * 2 * Lp * XLD);
}
constexpr size_t LDS_LIMIT = 156 * 1024;

bool shape_ok(int K, int N, int C, int M, int L) {
    // 8 float4 per lane cover a wave tile of at most 128 rows of 16 channels; orders come in groups of 8
    return C % CB == 0 && M % GRP == 0 && N <= 128 && 2 * L <= 128 && K >= 1 && L >= 1 &&
           analysis_lds(K, N, L) <= LDS_LIMIT && synthesis_lds(N, M, L) <= LDS_LIMIT;
}

ShtDev make(const float* in, const float* t1, const float* t2, float* out, int B, int K, int N, int C, int M, int L) {
    ShtDev a{in, out, t1, t2, B, K, N, C, M, L, round_up(K, 16), round_up(L, 16), round_up(N, 16), round_up(2 * M, 16)};
    return a;
}

}  // namespace

extern "C" int dlwp_sht_fused_supported(int nlat, int nlon, int C, int mmax, int lmax) {
    return shape_ok(nlat, nlon, C, mmax, lmax) ? 1 : 0;
}

extern "C" int dlwp_sht_analysis(const float* x, const float* A1, const float* A2, float* X, int B, int nlat, int nlon, int C,
                                 int mmax, int lmax, void* stream) {
    DLWP_REQUIRE(x && A1 && A2 && X && B > 0, DLWP_E_INVALID, "sht_analysis: NULL argument or empty batch");
    DLWP_REQUIRE(shape_ok(nlat, nlon, C, mmax, lmax), DLWP_E_UNSUPPORTED,
                 "sht_analysis: shape (nlat %d, nlon %d, C %d, mmax %d, lmax %d) is outside the fused kernel's range "
                 "(C %% 16, mmax %% 8, nlon <= 128, lmax <= 64, tables within LDS): use the GEMM path", nlat, nlon, C, mmax, lmax);
    DLWP_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && B <= 65535, DLWP_E_INVALID, "sht_analysis: x must be 16-byte aligned");
    const ShtDev a = make(x, A1, A2, X, B, nlat, nlon, C, mmax, lmax);
    const size_t lds = analysis_lds(nlat, nlon, lmax);
    int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(sht_analysis_kernel), lds, "sht_analysis");
    if (rc) return rc;
    hipLaunchKernelGGL(sht_analysis_kernel, dim3(C / CB, mmax / GRP, B), dim3(256), lds, (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_sht_synthesis(const float* X, const float* S1t, const float* S2t, float* x, int B, int nlat, int nlon,
                                  int C, int mmax, int lmax, void* stream) {
    DLWP_REQUIRE(X && S1t && S2t && x && B > 0, DLWP_E_INVALID, "sht_synthesis: NULL argument or empty batch");
    DLWP_REQUIRE(shape_ok(nlat, nlon, C, mmax, lmax), DLWP_E_UNSUPPORTED,
                 "sht_synthesis: shape (nlat %d, nlon %d, C %d, mmax %d, lmax %d) is outside the fused kernel's range "
                 "(C %% 16, mmax %% 8, nlon <= 128, lmax <= 64, tables within LDS): use the GEMM path", nlat, nlon, C, mmax, lmax);
    DLWP_REQUIRE((reinterpret_cast<uintptr_t>(X) & 15) == 0 && B <= 65535, DLWP_E_INVALID, "sht_synthesis: X must be 16-byte aligned");
    const ShtDev a = make(X, S1t, S2t, x, B, nlat, nlon, C, mmax, lmax);
    const size_t lds = synthesis_lds(nlon, mmax, lmax);
    int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(sht_synthesis_kernel), lds, "sht_synthesis");
    if (rc) return rc;
    hipLaunchKernelGGL(sht_synthesis_kernel, dim3(C / CB, ceil_div(nlat, GRP), B), dim3(256), lds, (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

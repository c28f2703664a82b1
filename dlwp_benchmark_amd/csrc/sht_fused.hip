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
#include "common.hip.h"
#include "dlwpmi_internal.h"

namespace {

constexpr int CB = 16;          // channels per workgroup
constexpr int GRP = 8;          // orders (analysis) / latitudes (synthesis) per workgroup
constexpr int NW = 16;          // waves per workgroup: four per SIMD hide each other's LDS round trips and dependent MFMA
                                // chains (C3 step with these kernels forced on in bf16 GEMM mode: 4 waves 623, 8 waves 652,
                                // 16 waves 661 samples/s)
constexpr int XLD = CB + 4;     // row stride of [row][16 channel] tiles (lanes l / l+16 of a half-wave on different banks)

struct ShtDev {
    const float* in;            // analysis: x [B][K][N][C];  synthesis: X [L][B][M][2][C]
    float* out;                 // analysis: X;               synthesis: x
    const float* T1;            // analysis: A1 [2M][N];      synthesis: S1t [M][K][L]
    const float* T2;            // analysis: A2 [M][L][K];    synthesis: S2t [N][2M]
    int B, K, N, C, M, L;
    int Kp, Lp, Np, Qp;         // padded to 16 (zero-filled operands)
    FastDiv dQ4;                // division by 2M/4 (table staging of the synthesis kernel)
};

// 16 contiguous bytes of a table row, or zeros past the row's end (row lengths are multiples of 4, rows 16-byte aligned);
// the address is clamped so that the load itself is unconditional
__device__ __forceinline__ f32x4 table4(const float* __restrict__ row, int col, int ncols, bool row_ok) {
    const bool ok = row_ok && col + 3 < ncols;
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + (ok ? col : 0));
    return ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
}

// A 16 x 16 accumulator tile (lane (r, g) holds rows 4g+j, column r) leaves through a wave-private LDS tile so that every
// lane stores 16 bytes (row = lane / 4): one wave-instruction per tile instead of four 64-byte-segment dword stores each.
// dst_row(row) -> pointer of the 16-channel row in global memory, or nullptr for rows outside the tensor.
template <typename F>
__device__ __forceinline__ void store_tile16(const f32x4 acc, float* wt, int lane, F dst_row) {
    const int r = lane & 15, g = lane >> 4;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) wt[(4 * g + j) * XLD + r] = acc[j];
    __builtin_amdgcn_wave_barrier();
    const int row = lane >> 2, c4 = lane & 3;
    const f32x4 v = *reinterpret_cast<const f32x4*>(&wt[row * XLD + 4 * c4]);
    float* d = dst_row(row);
    if (d) *reinterpret_cast<f32x4*>(d + 4 * c4) = v;
    __builtin_amdgcn_wave_barrier();
}

constexpr int MAXCH = 8;        // 16-wide chunks of a longitude row (nlon <= 128)
constexpr int MAXKC = 4;        // 16-wide chunks of a latitude / degree row (nlat, lmax <= 64)

// ---------------------------------------------------------------------------------------------------------------------
// analysis: grid (C/16, M/8, B).  Tables never touch LDS: the A1 rows of this workgroup's 16 (order, re|im) pairs stay in
// registers for the whole latitude loop, the A2 fragments of the next (order, degree tile) are fetched while the current
// one is multiplied.
__global__ __launch_bounds__(NW * 64) void sht_analysis_kernel(ShtDev a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ts = smem;                                    // [Kp][16 q][XLD]
    float* xs = Ts + a.Kp * 16 * XLD;                    // [NW waves][Np][XLD]
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id(), r = lane & 15, g = lane >> 4;
    const int c0 = blockIdx.x * CB, m0 = blockIdx.y * GRP, b = blockIdx.z;
    float* xw = xs + w * a.Np * XLD;
    const int nch = a.Np / 16, kch = a.Kp / 16, ltn = a.Lp / 16;

    // this wave's latitudes: k = w, w+4, ...; a tile is x[b][k][0:N][c0:c0+16] (N rows of 64 bytes)
    const float* xb = a.in + (long long)b * a.K * a.N * a.C + c0;
    const int nld4 = (a.N * (CB / 4) + 63) / 64;         // float4 per lane and tile (N = 64 -> 4)
    float4 pre[8];
    auto issue = [&](int k) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = lane + 64 * u, n = idx >> 2, c4 = idx & 3;
            const bool ok = u < nld4 && n < a.N && k < a.K;
            const float4 v = *reinterpret_cast<const float4*>(xb + ((long long)(ok ? k : 0) * a.N + (ok ? n : 0)) * a.C + 4 * c4);
            pre[u] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
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
    f32x4 a1[MAXCH];
    {
        const float* row = a.T1 + (long long)(2 * m0 + r) * a.N;
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) a1[c] = c < nch ? table4(row, 16 * c + 4 * g, a.N, true) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // A2 fragments of (order ml, degree tile lt): rows l = 16 lt + r, columns k
    f32x4 a2c[MAXKC], a2n[MAXKC];
    auto issue_a2 = [&](int ml, int lt) {
        const int l = 16 * lt + r;
        const bool row_ok = l < a.L && m0 + ml < a.M;
        const float* row = a.T2 + ((long long)(m0 + (row_ok ? ml : 0)) * a.L + (row_ok ? l : 0)) * a.K;
#pragma unroll
        for (int c = 0; c < MAXKC; ++c) a2n[c] = c < kch ? table4(row, 16 * c + 4 * g, a.K, row_ok) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    issue_a2(w, 0);
    for (int e = tid; e < (a.Kp - a.K) * 16 * XLD; e += NW * 64) Ts[a.K * 16 * XLD + e] = 0.f;      // padded latitudes

    // ---- stage 1: T[k][q][c] = sum_n A1[q][n] x[k][n][c]
    for (int k = w; k < a.K; k += NW) {
        commit();
        if (k + NW < a.K) issue(k + NW);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
            if (c < nch) {
                f32x4 b4;
#pragma unroll
                for (int s = 0; s < 4; ++s) b4[s] = xw[(16 * c + 4 * g + s) * XLD + r];
                acc = mfma16_chunk(a1[c], b4, acc);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) Ts[(k * 16 + 4 * g + j) * XLD + r] = acc[j];
    }
    __syncthreads();

    // ---- stage 2: X[l][m][ri][c] = sum_k A2[m][l][k] T[k][2m+ri][c]
    for (int ml = w; ml < GRP; ml += NW) {
        const int m = m0 + ml;
        for (int lt = 0; lt < ltn; ++lt) {
#pragma unroll
            for (int c = 0; c < MAXKC; ++c) a2c[c] = a2n[c];
            if (lt + 1 < ltn) issue_a2(ml, lt + 1);
            else if (ml + NW < GRP) issue_a2(ml + NW, 0);
            f32x4 acc[2];
            acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[1] = acc[0];
#pragma unroll
            for (int c = 0; c < MAXKC; ++c) {
                if (c < kch) {
#pragma unroll
                    for (int ri = 0; ri < 2; ++ri) {
                        f32x4 b4;
#pragma unroll
                        for (int s = 0; s < 4; ++s) b4[s] = Ts[((16 * c + 4 * g + s) * 16 + 2 * ml + ri) * XLD + r];
                        acc[ri] = mfma16_chunk(a2c[c], b4, acc[ri]);
                    }
                }
            }
#pragma unroll
            for (int ri = 0; ri < 2; ++ri)
                store_tile16(acc[ri], xw, lane, [&](int row) -> float* {
                    const int l = 16 * lt + row;
                    return (l < a.L && m < a.M) ? a.out + ((((long long)l * a.B + b) * a.M + m) * 2 + ri) * a.C + c0 : nullptr;
                });
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// synthesis: grid (C/16, ceil(K/8), B).  S1 fragments (8 latitudes x degrees of one order) travel with the order's field
// tile; S2 (every order is contracted by every workgroup) is an LDS image filled with 16-byte copies.
__global__ __launch_bounds__(NW * 64) void sht_synthesis_kernel(ShtDev a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Q = 2 * a.M, QLD = a.Qp + 4;
    float* S2s = smem;                                   // [Np][QLD]        S2s[n][q] = S2[q][n]
    float* Ts = S2s + a.Np * QLD;                        // [GRP][Qp][XLD]
    float* xs = Ts + GRP * a.Qp * XLD;                   // [NW waves][2 Lp][XLD]   rows (l, ri)
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id(), r = lane & 15, g = lane >> 4;
    const int c0 = blockIdx.x * CB, k0g = blockIdx.y * GRP, b = blockIdx.z;
    float* xw = xs + w * 2 * a.Lp * XLD;
    const int lch = a.Lp / 16, qch = a.Qp / 16;

    // this wave's orders: m = w, w+4, ...; a tile is X[0:L][b][m][0:2][c0:c0+16]: 2 L rows of 64 bytes
    const float* Xb = a.in + (((long long)b * a.M) * 2) * a.C + c0;
    const long long lstride = (long long)a.B * a.M * 2 * a.C;
    const int nld4 = (2 * a.L * (CB / 4) + 63) / 64;     // L = 32 -> 4
    float4 pre[8];
    f32x4 s1n[MAXKC], s1c[MAXKC];
    auto issue = [&](int m) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = lane + 64 * u, row = idx >> 2, c4 = idx & 3, l = row >> 1, ri = row & 1;
            const bool ok = u < nld4 && l < a.L && m < a.M;
            const float4 v = *reinterpret_cast<const float4*>(Xb + (ok ? l : 0) * lstride + ((long long)(ok ? m : 0) * 2 + ri) * a.C + 4 * c4);
            pre[u] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int k = k0g + r;
        const bool row_ok = r < GRP && k < a.K && m < a.M;
        const float* srow = a.T1 + ((long long)(row_ok ? m : 0) * a.K + (row_ok ? k : 0)) * a.L;
#pragma unroll
        for (int c = 0; c < MAXKC; ++c) s1n[c] = c < lch ? table4(srow, 16 * c + 4 * g, a.L, row_ok) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = lane + 64 * u, row = idx >> 2, c4 = idx & 3;
            if (u < nld4 && row < 2 * a.Lp) *reinterpret_cast<float4*>(&xw[row * XLD + 4 * c4]) = pre[u];
        }
#pragma unroll
        for (int c = 0; c < MAXKC; ++c) s1c[c] = s1n[c];
    };
    issue(w);
    {   // S2t [N][Q] -> S2s rows of QLD floats (Q is a multiple of 16: orders come in groups of 8); rows n >= N are zero
        const int q4n = Q / 4;
        for (int u = tid; u < a.Np * q4n; u += NW * 64) {
            const int n = fastdiv(u, a.dQ4), q4 = u - n * q4n;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n < a.N) v = *reinterpret_cast<const f32x4*>(a.T2 + (long long)n * Q + 4 * q4);
            *reinterpret_cast<f32x4*>(&S2s[n * QLD + 4 * q4]) = v;
        }
    }
    // rows of the wave tile beyond 2 L stay zero for the whole kernel
    for (int e = lane; e < (2 * a.Lp - 2 * a.L) * XLD; e += 64) xw[2 * a.L * XLD + e] = 0.f;

    // ---- stage 1: T[kl][2m+ri][c] = sum_l S1[m][l][k0g+kl] X[l][m][ri][c]   (rows kl < 8 of the 16-row tile are real)
    for (int m = w; m < a.M; m += NW) {
        commit();
        if (m + NW < a.M) issue(m + NW);
        f32x4 acc[2];
        acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[1] = acc[0];
#pragma unroll
        for (int c = 0; c < MAXKC; ++c) {
            if (c < lch) {
#pragma unroll
                for (int ri = 0; ri < 2; ++ri) {
                    f32x4 b4;
#pragma unroll
                    for (int s = 0; s < 4; ++s) b4[s] = xw[((16 * c + 4 * g + s) * 2 + ri) * XLD + r];
                    acc[ri] = mfma16_chunk(s1c[c], b4, acc[ri]);
                }
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
    for (int kl = w; kl < GRP; kl += NW) {
        const int k = k0g + kl;
        for (int nt = 0; nt < a.Np / 16; ++nt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < qch; ++c) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(&S2s[(nt * 16 + r) * QLD + 16 * c + 4 * g]);
                f32x4 b4;
#pragma unroll
                for (int s = 0; s < 4; ++s) b4[s] = Ts[(kl * a.Qp + 16 * c + 4 * g + s) * XLD + r];
                acc = mfma16_chunk(a4, b4, acc);
            }
            store_tile16(acc, xw, lane, [&](int row) -> float* {
                const int n = nt * 16 + row;
                return (n < a.N && k < a.K) ? a.out + (((long long)b * a.K + k) * a.N + n) * a.C + c0 : nullptr;
            });
        }
    }
}

size_t analysis_lds(int K, int N, int L) {
    const int Kp = round_up(K, 16), Np = round_up(N, 16);
    (void)L;
    return sizeof(float) * ((size_t)Kp * 16 * XLD + (size_t)NW * Np * XLD);
}
size_t synthesis_lds(int N, int M, int L) {
    const int Np = round_up(N, 16), Lp = round_up(L, 16), Qp = round_up(2 * M, 16);
    return sizeof(float) * ((size_t)Np * (Qp + 4) + (size_t)GRP * Qp * XLD + (size_t)NW * 2 * Lp * XLD);
}
constexpr size_t LDS_LIMIT = 156 * 1024;

bool shape_ok(int K, int N, int C, int M, int L) {
    // 8 float4 per lane cover a wave tile of at most 128 rows of 16 channels; orders come in groups of 8; table rows
    // (lengths nlon, nlat, lmax) are read as 16-byte fragments, at most MAXCH / MAXKC chunks of 16 per row
    return C % CB == 0 && M % GRP == 0 && N <= 16 * MAXCH && L <= 16 * MAXKC && K <= 16 * MAXKC && N % 4 == 0 && K % 4 == 0 &&
           L % 4 == 0 && K >= 4 && L >= 4 && analysis_lds(K, N, L) <= LDS_LIMIT && synthesis_lds(N, M, L) <= LDS_LIMIT;
}

ShtDev make(const float* in, const float* t1, const float* t2, float* out, int B, int K, int N, int C, int M, int L) {
    ShtDev a{in, out, t1, t2, B, K, N, C, M, L, round_up(K, 16), round_up(L, 16), round_up(N, 16), round_up(2 * M, 16),
             make_fastdiv(2 * M / 4)};
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
                 "(C %% 16, mmax %% 8, nlon <= 128, nlat and lmax <= 64, all multiples of 4): use the GEMM path", nlat, nlon, C, mmax, lmax);
    DLWP_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && B <= 65535, DLWP_E_INVALID, "sht_analysis: x must be 16-byte aligned");
    const ShtDev a = make(x, A1, A2, X, B, nlat, nlon, C, mmax, lmax);
    const size_t lds = analysis_lds(nlat, nlon, lmax);
    int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(sht_analysis_kernel), lds, "sht_analysis");
    if (rc) return rc;
    hipLaunchKernelGGL(sht_analysis_kernel, dim3(C / CB, mmax / GRP, B), dim3(NW * 64), lds, (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_sht_synthesis(const float* X, const float* S1t, const float* S2t, float* x, int B, int nlat, int nlon,
                                  int C, int mmax, int lmax, void* stream) {
    DLWP_REQUIRE(X && S1t && S2t && x && B > 0, DLWP_E_INVALID, "sht_synthesis: NULL argument or empty batch");
    DLWP_REQUIRE(shape_ok(nlat, nlon, C, mmax, lmax), DLWP_E_UNSUPPORTED,
                 "sht_synthesis: shape (nlat %d, nlon %d, C %d, mmax %d, lmax %d) is outside the fused kernel's range "
                 "(C %% 16, mmax %% 8, nlon <= 128, nlat and lmax <= 64, all multiples of 4): use the GEMM path", nlat, nlon, C, mmax, lmax);
    DLWP_REQUIRE((reinterpret_cast<uintptr_t>(X) & 15) == 0 && B <= 65535, DLWP_E_INVALID, "sht_synthesis: X must be 16-byte aligned");
    const ShtDev a = make(X, S1t, S2t, x, B, nlat, nlon, C, mmax, lmax);
    const size_t lds = synthesis_lds(nlon, mmax, lmax);
    int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(sht_synthesis_kernel), lds, "sht_synthesis");
    if (rc) return rc;
    hipLaunchKernelGGL(sht_synthesis_kernel, dim3(C / CB, ceil_div(nlat, GRP), B), dim3(NW * 64), lds, (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

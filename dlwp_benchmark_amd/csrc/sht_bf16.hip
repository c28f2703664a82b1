// Fused real spherical harmonic transforms on the bf16 matrix cores (round 4): longitude DFT + Legendre transform of one
// transform in ONE launch, the (latitude x order) plane in LDS, for the bf16-storage chain of the SFNO blocks.
//
// Reference semantics: torch_harmonics.RealSHT / InverseRealSHT (third party; constructed for the SFNO at
// /root/reference/src/dlwpbench/models/fno/fno.py:183-200 and models/fourcastnet/fourcastnet.py:411-428);
// stages and tables are those of dlwp_benchmark_amd/sht.py's two strided-batched GEMMs and of csrc/sht_fused.hip (exact-f32):
//   analysis   X[l][b][m][ri][c] = sum_k A2[m][l][k] T[k][2m+ri][c],   T[k][q][c] = sum_n A1[q][n] x[b][k][n][c]
//   synthesis  x[b][k][n][c] = sum_q S2[n][q] T[k][q][c] (+ res),      T[k][2m+ri][c] = sum_l S1t[m][k][l] X[l][b][m][ri][c]
// x fp32 channels-last fields (the residual stream), X bf16 spectra (the GEMM-to-GEMM tensors of the chain), tables bf16 with the
// contraction index contiguous.  Each backward pass is the other kernel with the transposed tables.
//
// The two batched table GEMMs cost 8.6 + 10.1 us per transform at the C3 shape (B = 4, 32 x 64, C = 256) -- 128 launches,
// 1.2 ms of the 3.3 ms step -- for 0.8 GFLOP and 12 MB of traffic: prologue / epilogue latency of a generic GEMM on 2-K-step
// products.  Here a workgroup owns 16 channels of one sample and 8 orders (analysis) or 8 latitudes (synthesis): 256 workgroups.
// In both stages the FIELD side is the MFMA A operand, read from a [contraction index][16 channels] LDS image through the
// transposing read (ds_read_b64_tr_b16), and the TABLE side is the B operand, 16 contiguous bytes per lane straight from L2
// (held in registers where a workgroup reuses it); an accumulator lane then owns four consecutive channels of one output
// column: 8-byte LDS / bf16 stores, 16-byte fp32 stores.  Analysis converts its fp32 rows in registers (one latitude per wave
// at a time, the next one's loads in flight); synthesis fills its spectrum image by LDS-DMA (global_load_lds_dwordx4).
#include <algorithm>
#include <cstdlib>
#include "common.hip.h"
#include "dlwpmi_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
constexpr int CB = 16;          // channels per workgroup
constexpr int GRP = 8;          // orders (analysis) / latitudes (synthesis) per workgroup
constexpr int NW = 8;           // waves per workgroup

struct ShtB {
    const void* in;             // analysis: x fp32 [B][K][N][C];  synthesis: X bf16 [L][B][M][2][C]
    void* out;                  // analysis: X bf16;               synthesis: x fp32
    const float* res;           // synthesis: fp32 addend with x's layout (nullable)
    const __bf16* T1;           // analysis: A1 [2M][N];   synthesis: S1t [M][K][L]
    const __bf16* T2;           // analysis: A2 [M][L][K]; synthesis: S2 [N][2M]
    int B, K, N, C, M, L, ncb, ngrp;
    int tri;                    // synthesis: orders m > l of X are zero by spherical truncation and are not read
    int field_bf16;             // the field tensor (analysis input / synthesis output) is a bf16 array instead of fp32
};

// 16 contiguous table elements row[col .. col + 7] or zeros (row_ok && col + 7 < ncols; ncols % 8 == 0); the load is unconditional
__device__ __forceinline__ bf16x8 tab8(const __bf16* __restrict__ row, int col, int ncols, bool row_ok) {
    const bool ok = row_ok && col + 7 < ncols;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(row + (ok ? col : 0));
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = ok ? v[e] : (__bf16)0.f;
    return z;
}

// A fragment of a [contraction index][16 channels] bf16 image (32-byte rows): rows = channels, k = 32 kk + 8 g .. + 7
__device__ __forceinline__ bf16x8 field_frag(const __bf16* img, int kk, int r, int g) {
    const __bf16* p0 = img + (32 * kk + 8 * g + (r >> 2)) * CB + 4 * (r & 3);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * CB));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// A fragment of plane q of the synthesis kernel's spectrum image [degree][position][16 channels] (rows of QP * 32 bytes, plane q of
// degree l at position q ^ f(l)): rows = channels, k = degrees 32 kk + 8 g .. + 7
template <int QP>
__device__ __forceinline__ bf16x8 spec_frag(const __bf16* img, int q, int kk, int r, int g) {
    const int l0 = 32 * kk + 8 * g + (r >> 2), f = (r >> 2) | ((g & 1) << 2);          // f(l0) = f(l0 + 4)
    const __bf16* p0 = img + (l0 * QP + (q ^ f)) * CB + 4 * (r & 3);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * QP * CB));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__device__ __forceinline__ void block_ids(const ShtB& a, int& b, int& cb, int& grp) {
    // Speed only (any mapping is correct).  Workgroups i and i + 8 share an XCD under round-robin placement.  A 16-channel block
    // of a channels-last tensor is a 32-byte (bf16 spectrum) or 64-byte (fp32 field) piece of a 128-byte line, so the FOUR
    // channel blocks that share the lines of a spectrum go to one XCD (its L2 then fetches / writes whole lines once; dealt to
    // four XCDs every line crossed the fabric four times), and so do the groups of orders / latitudes of a (sample, channel
    // block), which read the same field: their workgroups run back to back on that XCD.
    const int per = a.B * a.ncb;
    int rest;
    if (per % 32 == 0 && a.ncb % 4 == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;            // j: position in the XCD's own sequence
        grp = j % a.ngrp;
        const int t = j / a.ngrp, cbl = t & 3, sc = t >> 2;            // four channel blocks of one line group, then the next group
        const int fid = sc * 8 + xcd, nq = a.ncb >> 2;                  // (sample, line group) index
        b = fid / nq;
        cb = 4 * (fid - b * nq) + cbl;
        return;
    }
    if (per % 8 == 0) {
        // eight fields (one per XCD under round-robin placement) at a time, their groups back to back: the groups of a field run on
        // one XCD and close in time, so the field is fetched from HBM once (B = 16: 1024 workgroups, four rounds)
        const int chunk = blockIdx.x / (8 * a.ngrp), in = blockIdx.x - chunk * 8 * a.ngrp;
        grp = in >> 3;
        rest = chunk * 8 + (in & 7);
    } else {
        grp = blockIdx.x / per;
        rest = blockIdx.x - grp * per;
    }
    b = rest / a.ncb;
    cb = rest - b * a.ncb;
}

// NKS: longitude k-steps (N <= 32 NKS); KKS: latitude k-steps (K <= 32 KKS); LT: degree tiles (L <= 16 LT)
template <int NKS, int KKS, int LT>
__global__ __launch_bounds__(512) void sht_analysis_bf16_kernel(ShtB a) {
    constexpr int NP = 32 * NKS, KPAD = 32 * KKS, NV = NP / 16;
    extern __shared__ __attribute__((aligned(16))) float sht_smem[];
    __bf16* Timg = reinterpret_cast<__bf16*>(sht_smem);            // [16 q][KPAD lat][CB]
    __bf16* tiles = Timg + 16 * KPAD * CB;                         // [NW][NP][CB] wave-private field rows of one latitude
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    int b, cb, mg;
    block_ids(a, b, cb, mg);
    const int c0 = cb * CB, q0 = 2 * GRP * mg, K = a.K, N = a.N, C = a.C, M = a.M, L = a.L;
    const float* x = static_cast<const float*>(a.in);
    __bf16* X = static_cast<__bf16*>(a.out);
    DLWP_STAMP(0);
    if (K < KPAD) {                                                // padding latitudes meet zero table entries: no NaN garbage
        for (int e = tid; e < 16 * (KPAD - K) * CB / 4; e += 512) {
            const int per = (KPAD - K) * CB / 4, q = e / per, o = e - q * per;
            *reinterpret_cast<bf16x4*>(Timg + (q * KPAD + K) * CB + 4 * o) = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        }
    }
    // ---- stage 1: T[k][q0 + r][c] for this wave's latitudes k = w, w + 8, ...
    bf16x8 t1[NKS];
#pragma unroll
    for (int kk = 0; kk < NKS; ++kk) t1[kk] = tab8(a.T1 + (long long)min(q0 + r, 2 * M - 1) * N, 32 * kk + 8 * g, N, q0 + r < 2 * M);
    __bf16* tile = tiles + w * NP * CB;
    auto load_rows = [&](int k, f32x4 (&v)[NV]) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int n = (lane >> 2) + 16 * i;
            const long long o = (((long long)b * K + min(k, K - 1)) * N + min(n, N - 1)) * C + c0 + 4 * (lane & 3);
            f32x4 u;
            if (a.field_bf16) {          // (a gradient that the block tail left as a bf16 array: half the bytes, the same rounding as below)
                const bf16x4 h = *reinterpret_cast<const bf16x4*>(static_cast<const __bf16*>(a.in) + o);
                u = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
            } else {
                u = *reinterpret_cast<const f32x4*>(x + o);
            }
            v[i] = n < N ? u : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // the wave's latitudes k = w, w + 8, ...: ALL their rows are requested before the first product where the registers allow
    // (round 4 kept one latitude ahead: four dependent HBM round trips of ~1,900 cycles each in the stamps)
    constexpr int NIT = 4 * KKS;                                   // latitudes per wave (K <= 32 KKS over NW = 8 waves)
    constexpr int DA = NIT * NV <= 32 ? NIT : 2;
    f32x4 rows[DA][NV];
#pragma unroll
    for (int i = 0; i < DA; ++i)
        if (w + NW * i < K) load_rows(w + NW * i, rows[i]);
    // stage 2's table fragments (order m = GRP mg + w) travel during stage 1
    const int m = GRP * mg + w;
    bf16x8 t2[LT][KKS];
#pragma unroll
    for (int lt = 0; lt < LT; ++lt)
#pragma unroll
        for (int kk = 0; kk < KKS; ++kk) {
            const int l = 16 * lt + r;
            t2[lt][kk] = tab8(a.T2 + ((long long)min(m, M - 1) * L + min(l, L - 1)) * K, 32 * kk + 8 * g, K, m < M && l < L);
        }
    DLWP_STAMP(1);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int k = w + NW * it;
        if (k < K) {
            const f32x4 (&cur)[NV] = rows[it % DA];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int n = (lane >> 2) + 16 * i;
                *reinterpret_cast<bf16x4*>(tile + n * CB + 4 * (lane & 3)) = bf16x4{(__bf16)cur[i][0], (__bf16)cur[i][1], (__bf16)cur[i][2], (__bf16)cur[i][3]};
            }
            __builtin_amdgcn_wave_barrier();
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NKS; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(field_frag(tile, kk, r, g), t1[kk], acc, 0, 0, 0);
            // lane (r, g): channels 4g .. 4g + 3 of column q0 + r
            *reinterpret_cast<bf16x4*>(Timg + (r * KPAD + k) * CB + 4 * g) = bf16x4{(__bf16)acc[0], (__bf16)acc[1], (__bf16)acc[2], (__bf16)acc[3]};
            __builtin_amdgcn_wave_barrier();
        }
        if (it + DA < NIT && w + NW * (it + DA) < K) load_rows(w + NW * (it + DA), rows[it % DA]);
    }
    DLWP_STAMP(2);
    // ---- stage 2: this wave's order m = GRP mg + w, both parts (its table fragments were requested before stage 1)
    lds_barrier();
    DLWP_STAMP(3);
    if (m < M) {
#pragma unroll
        for (int ri = 0; ri < 2; ++ri) {
            const __bf16* plane = Timg + (2 * w + ri) * KPAD * CB;
            f32x4 acc[LT];
#pragma unroll
            for (int lt = 0; lt < LT; ++lt) acc[lt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KKS; ++kk) {
                const bf16x8 f = field_frag(plane, kk, r, g);
#pragma unroll
                for (int lt = 0; lt < LT; ++lt) acc[lt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, t2[lt][kk], acc[lt], 0, 0, 0);
            }
#pragma unroll
            for (int lt = 0; lt < LT; ++lt) {
                const int l = 16 * lt + r;
                if (l < L)
                    *reinterpret_cast<bf16x4*>(X + ((((long long)l * a.B + b) * M + m) * 2 + ri) * C + c0 + 4 * g) =
                        bf16x4{(__bf16)acc[lt][0], (__bf16)acc[lt][1], (__bf16)acc[lt][2], (__bf16)acc[lt][3]};
            }
        }
    }
    DLWP_STAMP(4);
#ifdef DLWP_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DLWP_STAMP(5);
#endif
}

// LKS: degree k-steps (L <= 32 LKS); QKS: order k-steps (2 M <= 32 QKS); NT: longitude tiles (N <= 16 NT)
template <int LKS, int QKS, int NT>
__global__ __launch_bounds__(512) void sht_synthesis_bf16_kernel(ShtB a) {
    constexpr int LP = 32 * LKS, QP = 32 * QKS;
    extern __shared__ __attribute__((aligned(16))) float sht_smem[];
    __bf16* Ximg = reinterpret_cast<__bf16*>(sht_smem);            // [LP degrees][QP positions][CB]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    int b, cb, kg;
    block_ids(a, b, cb, kg);
    const int c0 = cb * CB, k0 = GRP * kg, K = a.K, N = a.N, C = a.C, M = a.M, L = a.L;
    __bf16* Timg = Ximg + LP * QP * CB;                            // [GRP lat][QP][CB]
    const __bf16* X = static_cast<const __bf16*>(a.in);
    float* x = static_cast<float*>(a.out);
    DLWP_STAMP(8);
    // ---- spectrum image by LDS-DMA: a wave-instruction moves 32 planes x 32 bytes of ONE degree (round 4 moved 32 degrees of one
    // plane: 32 pieces 2^17 bytes apart at B = 4 -- one memory channel per instruction; the pieces of a degree are 512 bytes apart).
    // Image [degree][position][CB]; position p of degree l holds plane p ^ f(l), f(l) = (l & 3) | 4 (l >> 3 & 1): the eight degrees a
    // transposing fragment read touches (l0 .. l0 + 3 and l0 + 8 .. l0 + 11) then sit in eight different 32-byte bank groups.
    // Truncated spectra (tri): degree l has orders m <= l only -- 48 % of the (degree, plane) pairs at lmax = mmax = 32.  Their
    // lanes issue no request (the image is filled at the L2 request rate: 32-byte pieces) and store zeros instead.
    for (int p = w; p < L * QKS; p += NW) {
        const int l = p / QKS, part = p - l * QKS, f = (l & 3) | (((l >> 3) & 1) << 2);
        const int plane = (32 * part + (lane >> 1)) ^ f;
        const __bf16* src = X + (((long long)l * a.B + b) * 2 * M + min(plane, 2 * M - 1)) * C + c0 + 8 * (lane & 1);
        __bf16* dst = Ximg + (l * QP + 32 * part) * CB;
        if (!a.tri || (plane >> 1) <= l) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        } else {
            *reinterpret_cast<bf16x8*>(dst + 8 * lane) = bf16x8{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        }
    }
    if (L < LP) {                                                  // padding degrees meet zero table entries
        for (int e = tid; e < (LP - L) * QP * CB / 4; e += 512)
            *reinterpret_cast<bf16x4*>(Ximg + L * QP * CB + 4 * e) = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    }
    if (2 * M < QP) {                                              // padding orders meet zero table entries
        for (int e = tid; e < GRP * (QP - 2 * M) * CB / 4; e += 512) {
            const int per = (QP - 2 * M) * CB / 4, kl = e / per, o = e - kl * per;
            *reinterpret_cast<bf16x4*>(Timg + (kl * QP + 2 * M) * CB + 4 * o) = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        }
    }
    // everything else this wave will read from global memory is requested now, behind the image: the stage-2 table (constant for
    // the workgroup), ALL stage-1 table fragments of the wave's planes (round 4 fetched them one plane ahead: eight dependent L2
    // round trips of ~740 cycles each in the stamps), and the residual rows of the wave's output latitude
    bf16x8 t2[NT][QKS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int kk = 0; kk < QKS; ++kk) {
            const int n = 16 * nt + r;
            t2[nt][kk] = tab8(a.T2 + (long long)min(n, N - 1) * 2 * M, 32 * kk + 8 * g, 2 * M, n < N);
        }
    constexpr int NQ = 4 * QKS;                                    // planes per wave: 2 M <= 32 QKS over NW = 8 waves
    constexpr int D1 = NQ * LKS + NT * QKS + 2 * NT <= 40 ? NQ : 2;    // stage-1 fragments in flight (all of them where the registers allow)
    const int klat = k0 + r;
    auto load_t1 = [&](int i, bf16x8 (&t)[LKS]) {
        const int q = w + NW * i;
#pragma unroll
        for (int kk = 0; kk < LKS; ++kk)
            t[kk] = tab8(a.T1 + ((long long)min(q >> 1, M - 1) * K + min(klat, K - 1)) * L, 32 * kk + 8 * g, L,
                         r < GRP && klat < K && q < 2 * M);
    };
    bf16x8 t1[D1][LKS];
#pragma unroll
    for (int i = 0; i < D1; ++i) load_t1(i, t1[i]);
    const int k = k0 + w;
    f32x4 rv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = 16 * nt + r;
        rv[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.res) rv[nt] = *reinterpret_cast<const f32x4*>(a.res + (((long long)b * K + min(k, K - 1)) * N + min(n, N - 1)) * C + c0 + 4 * g);
    }
    DLWP_STAMP(9);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DLWP_STAMP(10);
    lds_barrier();
    DLWP_STAMP(11);
    // ---- stage 1: T[k0 + r][q][c] for the planes q = w, w + 8, ... (columns r >= GRP of the tile are not used)
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int q = w + NW * i;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        if (q < 2 * M) {
#pragma unroll
            for (int kk = 0; kk < LKS; ++kk)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(spec_frag<QP>(Ximg, q, kk, r, g), t1[i % D1][kk], acc, 0, 0, 0);
            if (r < GRP)
                *reinterpret_cast<bf16x4*>(Timg + (r * QP + q) * CB + 4 * g) = bf16x4{(__bf16)acc[0], (__bf16)acc[1], (__bf16)acc[2], (__bf16)acc[3]};
        }
        if (i + D1 < NQ) load_t1(i + D1, t1[i % D1]);
    }
    DLWP_STAMP(12);
    lds_barrier();
    DLWP_STAMP(13);
    // ---- stage 2: latitude k0 + w, every longitude tile
    if (k < K) {
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = rv[nt];
#pragma unroll
        for (int kk = 0; kk < QKS; ++kk) {
            const bf16x8 f = field_frag(Timg + w * QP * CB, kk, r, g);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, t2[nt][kk], acc[nt], 0, 0, 0);
        }
        if (a.field_bf16) {              // y of an SFNO block: read once, by the block tail, as a bf16 operand addend
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = 16 * nt + r;
                if (n < N)
                    *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(a.out) + (((long long)b * K + k) * N + n) * C + c0 + 4 * g) =
                        bf16x4{(__bf16)acc[nt][0], (__bf16)acc[nt][1], (__bf16)acc[nt][2], (__bf16)acc[nt][3]};
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = 16 * nt + r;
                if (n < N) *reinterpret_cast<f32x4*>(x + (((long long)b * K + k) * N + n) * C + c0 + 4 * g) = acc[nt];
            }
        }
    }
    DLWP_STAMP(14);
#ifdef DLWP_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DLWP_STAMP(15);
#endif
}

bool shape_ok(int K, int N, int C, int M, int L) {
    // 16-byte table chunks and field pieces; the synthesis image 2 M x round32(L) x 32 bytes and the analysis tiles must fit the LDS
    if (K < 1 || N < 8 || C < CB || M < 1 || L < 1) return false;
    if (C % CB || N % 8 || K % 8 || L % 8 || (2 * M) % 8) return false;
    if (N > 128 || K > 64 || L > 64 || 2 * M > 128) return false;
    const size_t syn = (size_t)(round_up(L, 32) * round_up(2 * M, 32) + GRP * round_up(2 * M, 32)) * CB * 2;
    const size_t ana = (size_t)(16 * round_up(K, 32) + NW * round_up(N, 32)) * CB * 2;
    return syn <= 150 * 1024 && ana <= 150 * 1024;
}

template <int NKS, int KKS>
int analysis_lt(const ShtB& a, int LT, size_t lds, dim3 grid, hipStream_t s) {
#define DLWP_ANA(lt)                                                                                                       \
    case lt: {                                                                                                             \
        auto kern = sht_analysis_bf16_kernel<NKS, KKS, lt>;                                                                \
        if (int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(kern), lds, "sht_analysis_bf16")) return rc;            \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, a);                                                              \
        break;                                                                                                             \
    }
    switch (LT) {
        DLWP_ANA(1) DLWP_ANA(2) DLWP_ANA(3) DLWP_ANA(4)
        default: dlwp_set_error("sht_analysis_bf16: lmax too large"); return DLWP_E_UNSUPPORTED;
    }
#undef DLWP_ANA
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

template <int LKS, int QKS>
int synthesis_nt(const ShtB& a, int NT, size_t lds, dim3 grid, hipStream_t s) {
#define DLWP_SYN(nt)                                                                                                       \
    case nt: {                                                                                                             \
        auto kern = sht_synthesis_bf16_kernel<LKS, QKS, nt>;                                                               \
        if (int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(kern), lds, "sht_synthesis_bf16")) return rc;           \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, a);                                                              \
        break;                                                                                                             \
    }
    switch (NT) {
        DLWP_SYN(1) DLWP_SYN(2) DLWP_SYN(3) DLWP_SYN(4) DLWP_SYN(5) DLWP_SYN(6) DLWP_SYN(7) DLWP_SYN(8)
        default: dlwp_set_error("sht_synthesis_bf16: nlon too large"); return DLWP_E_UNSUPPORTED;
    }
#undef DLWP_SYN
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

#ifdef DLWP_STAMPS
extern "C" int dlwp_debug_stamps_sht(unsigned long long* host_out) {
    DLWP_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dlwp_stamps), sizeof(unsigned long long) * 32));
    return DLWP_OK;
}
#endif

extern "C" int dlwp_sht_bf16_supported(int nlat, int nlon, int C, int mmax, int lmax) { return shape_ok(nlat, nlon, C, mmax, lmax) ? 1 : 0; }

extern "C" int dlwp_sht_analysis_bf16_ex(const void* x, const void* A1, const void* A2, void* X, int B, int nlat, int nlon, int C, int mmax,
                                         int lmax, int flags, void* stream) {
    DLWP_REQUIRE(x && A1 && A2 && X && B > 0, DLWP_E_INVALID, "sht_analysis_bf16: null pointer / empty batch");
    DLWP_REQUIRE(shape_ok(nlat, nlon, C, mmax, lmax), DLWP_E_UNSUPPORTED, "sht_analysis_bf16: shape %d x %d, C %d, mmax %d, lmax %d unsupported "
                 "(dlwp_sht_bf16_supported)", nlat, nlon, C, mmax, lmax);
    DLWP_REQUIRE(aligned16(x) && aligned16(A1) && aligned16(A2) && aligned16(X), DLWP_E_INVALID, "sht_analysis_bf16: 16-byte alignment");
    DLWP_REQUIRE((flags & ~DLWP_SHT_FIELD_BF16) == 0, DLWP_E_INVALID, "sht_analysis_bf16: unknown flag bits %d", flags);
    ShtB a{x, X, nullptr, static_cast<const __bf16*>(A1), static_cast<const __bf16*>(A2), B, nlat, nlon, C, mmax, lmax, C / CB,
           ceil_div(mmax, GRP), 0, (flags & DLWP_SHT_FIELD_BF16) ? 1 : 0};
    const int NKS = ceil_div(nlon, 32), KKS = ceil_div(nlat, 32), LT = ceil_div(lmax, 16);
    const size_t lds = (size_t)(16 * 32 * KKS + NW * 32 * NKS) * CB * sizeof(__bf16);
    const dim3 grid(B * a.ncb * a.ngrp);
    const hipStream_t s = (hipStream_t)stream;
    if (KKS == 1) {
        if (NKS == 1) return analysis_lt<1, 1>(a, LT, lds, grid, s);
        if (NKS == 2) return analysis_lt<2, 1>(a, LT, lds, grid, s);
        if (NKS == 3) return analysis_lt<3, 1>(a, LT, lds, grid, s);
        return analysis_lt<4, 1>(a, LT, lds, grid, s);
    }
    if (NKS == 1) return analysis_lt<1, 2>(a, LT, lds, grid, s);
    if (NKS == 2) return analysis_lt<2, 2>(a, LT, lds, grid, s);
    if (NKS == 3) return analysis_lt<3, 2>(a, LT, lds, grid, s);
    return analysis_lt<4, 2>(a, LT, lds, grid, s);
}

extern "C" int dlwp_sht_synthesis_bf16_ex(const void* X, const void* S1t, const void* S2, const float* residual, void* x, int B, int nlat,
                                          int nlon, int C, int mmax, int lmax, int flags, void* stream) {
    DLWP_REQUIRE(X && S1t && S2 && x && B > 0, DLWP_E_INVALID, "sht_synthesis_bf16: null pointer / empty batch");
    DLWP_REQUIRE(shape_ok(nlat, nlon, C, mmax, lmax), DLWP_E_UNSUPPORTED, "sht_synthesis_bf16: shape %d x %d, C %d, mmax %d, lmax %d unsupported "
                 "(dlwp_sht_bf16_supported)", nlat, nlon, C, mmax, lmax);
    DLWP_REQUIRE(aligned16(X) && aligned16(S1t) && aligned16(S2) && aligned16(x) && aligned16(residual), DLWP_E_INVALID,
                 "sht_synthesis_bf16: 16-byte alignment");
    DLWP_REQUIRE((flags & ~(DLWP_SHT_TRIANGULAR | DLWP_SHT_FIELD_BF16)) == 0, DLWP_E_INVALID, "sht_synthesis_bf16: unknown flag bits %d", flags);
    DLWP_REQUIRE(!(flags & DLWP_SHT_FIELD_BF16) || !residual, DLWP_E_INVALID, "sht_synthesis_bf16: a bf16 output takes no residual");
    ShtB a{X, x, residual, static_cast<const __bf16*>(S1t), static_cast<const __bf16*>(S2), B, nlat, nlon, C, mmax, lmax, C / CB,
           ceil_div(nlat, GRP), (flags & DLWP_SHT_TRIANGULAR) ? 1 : 0, (flags & DLWP_SHT_FIELD_BF16) ? 1 : 0};
    const int LKS = ceil_div(lmax, 32), QKS = ceil_div(2 * mmax, 32), NT = ceil_div(nlon, 16);
    const size_t lds = (size_t)(32 * LKS * 32 * QKS + GRP * 32 * QKS) * CB * sizeof(__bf16);
    const dim3 grid(B * a.ncb * a.ngrp);
    const hipStream_t s = (hipStream_t)stream;
    if (LKS == 1) {
        if (QKS == 1) return synthesis_nt<1, 1>(a, NT, lds, grid, s);
        if (QKS == 2) return synthesis_nt<1, 2>(a, NT, lds, grid, s);
        if (QKS == 3) return synthesis_nt<1, 3>(a, NT, lds, grid, s);
        return synthesis_nt<1, 4>(a, NT, lds, grid, s);
    }
    if (QKS == 1) return synthesis_nt<2, 1>(a, NT, lds, grid, s);
    if (QKS == 2) return synthesis_nt<2, 2>(a, NT, lds, grid, s);
    if (QKS == 3) return synthesis_nt<2, 3>(a, NT, lds, grid, s);
    return synthesis_nt<2, 4>(a, NT, lds, grid, s);
}

extern "C" int dlwp_sht_analysis_bf16(const float* x, const void* A1, const void* A2, void* X, int B, int nlat, int nlon, int C, int mmax,
                                      int lmax, void* stream) {
    return dlwp_sht_analysis_bf16_ex(x, A1, A2, X, B, nlat, nlon, C, mmax, lmax, 0, stream);
}

extern "C" int dlwp_sht_synthesis_bf16(const void* X, const void* S1t, const void* S2, const float* residual, float* x, int B, int nlat,
                                       int nlon, int C, int mmax, int lmax, void* stream) {
    return dlwp_sht_synthesis_bf16_ex(X, S1t, S2, residual, x, B, nlat, nlon, C, mmax, lmax, 0, stream);
}

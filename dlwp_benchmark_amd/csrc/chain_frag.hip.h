// Fragment plumbing shared by the one-launch token-MLP chains (mlp_chain.hip: SFNO block tail; sfno_io.hip: SFNO encoder / decoder):
// weights as the MFMA A operand straight from a fragment-order image in L2, tokens as the B operand from a swizzled LDS image.
//   image[(tile * KS + kk) * 64 + lane][e] = W'[16 tile + (lane & 15)][32 kk + 8 (lane >> 4) + e]      (dlwp_mlp_chain_pack)
//   LDS image [token row][feature]: bf16 rows of L elements in 16-byte chunks, chunk c of row r stored at c ^ (r & mask(c)); the
//   mask covers the aligned power-of-two group of chunks that contains c (15 for rows that are multiples of 128 elements), so a
//   ds_read_b128 fragment read (16 rows x one chunk column per 16-lane group) touches 16 different 16-byte slots of a bank row.
#pragma once
#include "common.hip.h"

namespace chainfrag {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// swizzle mask of chunk c in a row of L elements (L % 32 == 0): rows are cut into aligned groups of 16, 8 and 4 chunks
template <int L>
__device__ __forceinline__ int cmask(int c) {
    constexpr int CH = L / 8, FULL = CH & ~15, REM = CH - FULL;
    static_assert(L % 32 == 0, "image rows are multiples of 32 elements");
    if (REM == 0) return 15;
    if (c < FULL) return 15;
    if (REM == 4) return 3;
    if (REM == 8) return 7;
    return (c - FULL) < 8 ? 7 : 3;          // REM == 12
}

template <int NT, int KH> struct WFrag { bf16x8 f[NT][KH]; };

// the fragments of k-steps k0 .. k0 + KH of this wave's n-tiles (tile = w + 8 ni; clamped: surplus tiles repeat the last one)
template <int NT, int KS, int KH, int NTL>
__device__ __forceinline__ void wload(WFrag<NT, KH>& wf, const __bf16* __restrict__ img, int w, int lane, int k0) {
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        const int tile = min(w + 8 * ni, NTL - 1);
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) {
            const bf16x8* p = reinterpret_cast<const bf16x8*>(img + ((long long)(tile * KS + k0 + kk) * 64 + lane) * 8);
#ifdef DLWP_CHAIN_NT          // measurement build: weight fragments are read once per CU -- non-temporal loads (no L1 allocation)
            wf.f[ni][kk] = __builtin_nontemporal_load(p);
#else
            wf.f[ni][kk] = *p;
#endif
        }
    }
}

// acc[mi][ni] += W'[tile ni rows][k] . act[token tile mi][k] over k-steps k0 .. k0 + KH of the LDS image (rows of LROW elements)
template <int MT, int NT, int KH, int LROW>
__device__ __forceinline__ void mma(f32x4 (&acc)[MT][NT], const WFrag<NT, KH>& wf, const __bf16* img, int k0, int r, int g) {
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) {
        bf16x8 tf[MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const int row = 16 * mi + r, c0 = 4 * (k0 + kk) + g, c = c0 ^ (row & cmask<LROW>(c0));
            tf[mi] = *reinterpret_cast<const bf16x8*>(img + row * LROW + 8 * c);
        }
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf.f[ni][kk], tf[mi], acc[mi][ni], 0, 0, 0);
    }
}

template <int MT, int NT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[MT][NT]) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ bf16x4 to_bf4(const float (&v)[4]) { return bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]}; }

// element offset of features n .. n + 3 (n % 4 == 0) of image row `row`
template <int LROW>
__device__ __forceinline__ int img_off(int row, int n) {
    const int c = n >> 3;
    return row * LROW + 8 * (c ^ (row & cmask<LROW>(c))) + (n & 7);
}
template <int LROW>
__device__ __forceinline__ void img_store(__bf16* img, int row, int n, bf16x4 v) {
    *reinterpret_cast<bf16x4*>(img + img_off<LROW>(row, n)) = v;
}

// ---- 16-byte epilogue pieces.  An accumulator lane (r, g) owns FOUR consecutive features of token r: 8-byte bf16 pieces, and the
// epilogues of these chains are bound by the ISSUE of their stores (stamps: 64 KB of 8-byte stores per 32-token workgroup take
// 9 - 10 k cycles, ~7 B / cycle / CU, whatever the memory system does).  v_permlane16_swap_b32 (gfx950) exchanges the odd 16-lane
// rows of one register with the even rows of another: applied to the packed pieces of TWO token tiles (same features) it leaves
// every lane with EIGHT consecutive features of one token -- lanes of even g: features 4 g .. 4 g + 7 of tile `mi`, lanes of odd g:
// features 4 (g - 1) .. 4 g + 3 of tile `mi + 1` -- i.e. half as many, 16-byte, store instructions (global and LDS).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 pair_rows(bf16x4 lo_tile, bf16x4 hi_tile) {
    const u32x2 a = __builtin_bit_cast(u32x2, lo_tile), b = __builtin_bit_cast(u32x2, hi_tile);
    const u32x2 s0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
    const u32x2 s1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
    return __builtin_bit_cast(bf16x8, (u32x4{s0[0], s1[0], s0[1], s1[1]}));
}
// after pair_rows on tiles (mi, mi + 1) of n-tile `tile`: the token row (inside the workgroup's tile) and first feature of this lane's piece
__device__ __forceinline__ int pair_row(int mi, int r, int g) { return 16 * (mi + (g & 1)) + r; }
__device__ __forceinline__ int pair_col(int tile, int g) { return 16 * tile + 8 * (g >> 1); }
template <int LROW>
__device__ __forceinline__ void img_store8(__bf16* img, int row, int n, bf16x8 v) {          // n % 8 == 0: one whole 16-byte chunk
    const int c = n >> 3;
    *reinterpret_cast<bf16x8*>(img + row * LROW + 8 * (c ^ (row & cmask<LROW>(c)))) = v;
}

// Epilogue of a hidden stage: activation (forward: GELU(acc + add); backward: acc * d) into the next stage's LDS image and to
// global memory (`gact`); the forward pass stores d = GELU'(acc + add) to `gz` -- evaluated together with the activation from the
// fp32 pre-activation (three more instructions), so that the backward pass multiplies by a stored factor instead of evaluating
// an exponential and a reciprocal per element (its epilogues were VALU-bound on exactly that); the backward product also goes
// out in fp32 to `gf32` (nullable).
// add(mi, ni): the forward addend of tile (mi, ni) (bias, or bias + a per-token term)
template <int MT, int NT, int N, int NTL, bool BWD, class AddFn>
__device__ __forceinline__ void chain_epilogue(const f32x4 (&acc)[MT][NT], AddFn add, const bf16x4 (&ez)[MT][NT],
                                               __bf16* img, __bf16* gact, __bf16* gz, float* gf32, int m0, int T, int w, int r, int g) {
    static_assert(MT % 2 == 0, "token tiles are paired for the 16-byte stores");
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        const int tile = w + 8 * ni;
        if (tile < NTL) {
#pragma unroll
            for (int mp = 0; mp < MT; mp += 2) {
                bf16x4 ab[2], zb[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int mi = mp + h;
                    float d[4], act[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (BWD) {
                            d[q] = 0.f;
                            act[q] = acc[mi][ni][q] * (float)ez[mi][ni][q];
                        } else {
                            gelu_both(acc[mi][ni][q] + add(mi, ni)[q], act[q], d[q]);
                        }
                    }
                    ab[h] = to_bf4(act);
                    zb[h] = to_bf4(d);
                    if (BWD && gf32) {
                        const long long m = m0 + 16 * mi + r;
                        if (m < T) *reinterpret_cast<f32x4*>(gf32 + m * N + 16 * tile + 4 * g) = f32x4{act[0], act[1], act[2], act[3]};
                    }
                }
                const bf16x8 a8 = pair_rows(ab[0], ab[1]);
                const int row = pair_row(mp, r, g), n = pair_col(tile, g);
                const long long m = m0 + row;
                img_store8<N>(img, row, n, a8);
                if (m < T) *reinterpret_cast<bf16x8*>(gact + m * N + n) = a8;
                if (!BWD) {
                    const bf16x8 z8 = pair_rows(zb[0], zb[1]);
                    if (m < T) *reinterpret_cast<bf16x8*>(gz + m * N + n) = z8;
                }
                // one tile pair at a time: left to itself the scheduler evaluates every GELU of the stage before the first store
                // (128 more live registers at MT = 4: 156 bytes of scratch per lane)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// Input tile [ROWS][K] fp32 -> swizzled bf16 LDS image (+ the bf16 copy a weight-gradient product reads), eight features per thread
// and piece: 32-byte loads, 16-byte LDS / global stores
template <int ROWS, int K>
struct InTile {
    static constexpr int XU = (ROWS * K / 8 + 511) / 512;
    float4 lo[XU], hi[XU];
    __device__ __forceinline__ void issue(const float* __restrict__ in, int m0, int T, int tid) {
#pragma unroll
        for (int i = 0; i < XU; ++i) {
            const int u = min(tid + 512 * i, ROWS * K / 8 - 1), row = u / (K / 8), c8 = u - row * (K / 8);
            const float4* p = reinterpret_cast<const float4*>(in + (long long)min(m0 + row, T - 1) * K + 8 * c8);
            lo[i] = p[0];
            hi[i] = p[1];
        }
    }
    __device__ __forceinline__ void commit(__bf16* img, __bf16* in_lp, int m0, int T, int tid) const {
#pragma unroll
        for (int i = 0; i < XU; ++i) {
            const int u = tid + 512 * i;
            if (u < ROWS * K / 8) {
                const int row = u / (K / 8), k = 8 * (u - row * (K / 8));
                const bf16x8 b = bf16x8{(__bf16)lo[i].x, (__bf16)lo[i].y, (__bf16)lo[i].z, (__bf16)lo[i].w,
                                        (__bf16)hi[i].x, (__bf16)hi[i].y, (__bf16)hi[i].z, (__bf16)hi[i].w};
                img_store8<K>(img, row, k, b);
                if (in_lp && m0 + row < T) *reinterpret_cast<bf16x8*>(in_lp + (long long)(m0 + row) * K + k) = b;
            }
        }
    }
};

}  // namespace chainfrag

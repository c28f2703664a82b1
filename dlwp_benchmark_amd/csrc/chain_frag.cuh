// Fragment plumbing shared by the one-launch token-MLP chains (mlp_chain.hip: SFNO block tail; sfno_io.hip: SFNO encoder / decoder):
// weights as the MFMA A operand straight from a fragment-order image in L2, tokens as the B operand from a swizzled LDS image.
//   image[(tile * KS + kk) * 64 + lane][e] = W'[16 tile + (lane & 15)][32 kk + 8 (lane >> 4) + e]      (dlwp_mlp_chain_pack)
//   LDS image [token row][feature]: bf16 rows of L elements in 16-byte chunks, chunk c of row r stored at c ^ (r & mask(c)); the
//   mask covers the aligned power-of-two group of chunks that contains c (15 for rows that are multiples of 128 elements), so a
//   ds_read_b128 fragment read (16 rows x one chunk column per 16-lane group) touches 16 different 16-byte slots of a bank row.
#pragma once
#include "common.cuh"

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

}  // namespace chainfrag

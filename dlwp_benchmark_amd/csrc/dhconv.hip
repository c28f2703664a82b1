// SFNO "driscoll-healy" spectral convolution (per-degree complex channel mixing) as dedicated bf16 kernels (round 4).
//
// Reference: the spectral layer of torch_harmonics' SphericalFourierNeuralOperatorNet (constructed at
// /root/reference/src/dlwpbench/models/fno/fno.py:183-200): Y[b, o, l, m] = sum_i X[b, i, l, m] W[i, o, l]
// ("bixy,iox->boxy", complex), weight [Cin][Cout][L] complex = fp32 [Cin][Cout][L][2].
// Spectra here are degree-major, channels-last bf16 arrays X[l][row][c] with row = (b, m, re|im): the (re, im) pair of one complex
// row are two CONSECUTIVE rows.  With that layout the contraction needs no expanded [[Wr, Wi], [-Wi, Wr]] image:
//     P = [X_re ; X_im] . [Wr | Wi]      (every row against both weight planes: the tensor as it lies, half the weight bytes)
//     Y_re = P[re row][Wr] - P[im row][Wi],      Y_im = P[re row][Wi] + P[im row][Wr]
// and the two P values a result needs sit in one MFMA tile: the weight image interleaves 8 output channels of Wr with the same 8
// of Wi per 16-row tile, so (weights = A operand, spectrum rows = B operand) lane (row r, g < 2) holds the Wr products of four
// channels and lane (row r ^ 1, g + 2) the matching Wi products -- one cross-lane read per value.  The input-gradient product
// gX = gY conj(W)^T is the same kernel on the transposed image with the sign flipped.
// Spherical truncation: X[l][b][m] = 0 for m > l.  A 16-row tile holds 8 orders of one sample, so tiles with m0 > l are neither
// loaded nor multiplied (their output rows are written as zeros): 37 % of the work at lmax = mmax = 32.
// The weight gradient gW[i, o, l] = sum_rows conj(X) gY is ONE product per degree over ALL lead times of a rollout (the caller
// hands in one (X, gY) pair per lead time): G = X^T [gY | gY'] with gY' = the row pairs of gY swapped and the second negated
// (done on the MFMA fragment in registers), into an [L][Cin][2 Cout] fp32 image that dlwp_dhconv_fold adds to the parameter
// gradient -- no per-lead-time launches, no end-of-backward callback.
// As batched token GEMMs on the expanded image the three products cost 16 + 15 + 21 us per block and lead time (0.84 ms of
// the 3.2 ms C3 step): 16-K-step products on the generic register-staged kernel.
#include <algorithm>
#include <cstdlib>
#include "chain_frag.hip.h"
#include "dlwpmi_internal.h"

namespace {

using namespace chainfrag;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
constexpr int RC = 256;          // spectrum rows per workgroup of the apply kernel (LDS image RC x K bf16)
constexpr int MAXS = DLWP_WGRAD_MAX_SEGMENTS;

// ---- weight images: img[which][l][tile][kk][lane][8], which = 0 forward (tiles over o, k over i), 1 backward (tiles over i, k over o);
// a tile's rows 0..7 are the real parts of its 8 channels, rows 8..15 the imaginary parts
constexpr int MAXW = 16;         // spectral weights per pack launch
struct PackMany { const float* w[MAXW]; __bf16* fimg[MAXW]; __bf16* bimg[MAXW]; };
__global__ __launch_bounds__(256) void dhconv_pack_kernel(PackMany pm, int Ci, int Co, int L) {
    extern __shared__ __attribute__((aligned(16))) float pk_smem[];      // [256 (i, o) pairs][2 L + 1]
    const int which = blockIdx.z & 1, layer = blockIdx.z >> 1, tid = threadIdx.x, LD = 2 * L + 1;
    const float* __restrict__ w = pm.w[layer];
    __bf16* __restrict__ fimg = pm.fimg[layer];
    __bf16* __restrict__ bimg = pm.bimg[layer];
    // forward: block (kk = 32 inputs, t = 8 outputs); backward: block (t = 8 inputs, kk = 32 outputs)
    const int kk = blockIdx.x, t = blockIdx.y;
    const int KS = (which ? Co : Ci) / 32, NT = (which ? Ci : Co) / 8;
    if (kk >= KS || t >= NT) return;
    const int i0 = which ? 8 * t : 32 * kk, o0 = which ? 32 * kk : 8 * t, ni = which ? 8 : 32, no = which ? 32 : 8;
    // parameter rows w[i][o][:][:] are 2 L contiguous floats
    for (int e = tid; e < ni * no * 2 * L; e += 256) {
        const int pair = e / (2 * L), j = e - pair * 2 * L, ii = pair / no, oo = pair - ii * no;
        pk_smem[pair * LD + j] = w[(((long long)(i0 + ii) * Co + o0 + oo) * L) * 2 + j];
    }
    __syncthreads();
    __bf16* img = which ? bimg : fimg;
    const int lane = tid & 63, r = lane & 15, g = lane >> 4, part = r >> 3, ch = r & 7;
    for (int l = tid >> 6; l < L; l += 4) {
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kx = 8 * g + e;                                   // index inside the 32-deep k block
            const int pair = which ? ch * no + kx : kx * no + ch;       // (i, o) pair in the staged slab
            v[e] = (__bf16)pk_smem[pair * LD + 2 * l + part];
        }
        *reinterpret_cast<bf16x8*>(img + ((((long long)l * NT + t) * KS + kk) * 64 + lane) * 8) = v;
    }
}

// Round 5: both images of a weight from ONE read of it.  dhconv_pack_kernel reads every weight twice (once per image, 256 (i, o)
// pairs per workgroup): 83 us per step for the four C3 layers (201 MB moved).  Here a workgroup stages a 32 x 32 block of (i, o)
// pairs (all degrees, 256 KB of fp32) as bf16 in LDS, transposed to [o][(l, part)][i] -- the forward image's eight consecutive
// inputs are then one 16-byte piece -- and writes the block's 4 + 4 tiles of both images: 134 MB moved.
constexpr int PK_PJ = 36;          // element pitch of one (o, l, part) row of 32 inputs.  The forward image's bf16x4 reads start at
                                   // (8 t + ch) * PO with PO = 72 L + 2 elements: 4-byte aligned for odd ch (LDS accesses of 8 bytes at
                                   // 4-byte alignment are legal on gfx950 in the default unaligned-access mode; the + 2 spreads the banks)
__global__ __launch_bounds__(512) void dhconv_pack2_kernel(PackMany pm, int Ci, int Co, int L) {
    extern __shared__ __attribute__((aligned(16))) float pk_smem[];
    __bf16* sm = reinterpret_cast<__bf16*>(pk_smem);                     // [32 o][2 L][PK_PJ] (+ 2 elements per o: bank spread)
    const int J = 2 * L, PO = J * PK_PJ + 2, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i0 = 32 * blockIdx.x, o0 = 32 * blockIdx.y, layer = blockIdx.z;
    const float* __restrict__ w = pm.w[layer];
    // ---- the block's 32 input rows: 32 outputs x 2 L floats contiguous each
    const int per_i = 32 * J / 4;                                        // float4 pieces per input row
    for (int ib = 0; ib < 32; ib += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4* row = reinterpret_cast<const float4*>(w + ((long long)(i0 + ib + u) * Co + o0) * J);
            v[u] = tid < per_i ? row[tid] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid < per_i) {
            const int o = (4 * tid) / J, j = 4 * tid - o * J;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                __bf16* d = sm + o * PO + j * PK_PJ + ib + u;
                d[0] = (__bf16)v[u].x; d[PK_PJ] = (__bf16)v[u].y; d[2 * PK_PJ] = (__bf16)v[u].z; d[3 * PK_PJ] = (__bf16)v[u].w;
            }
        }
    }
    __syncthreads();
    const int r = lane & 15, g = lane >> 4, part = r >> 3, ch = r & 7;
    // ---- forward image: tiles over o (8 per tile), k over i: lane holds inputs 8 g .. 8 g + 7 of output 8 t + ch
    {
        __bf16* __restrict__ img = pm.fimg[layer];
        const int NT = Co / 8, KS = Ci / 32, kk = i0 / 32;
        for (int e = wv; e < 4 * L; e += 8) {
            const int l = e >> 2, t = e & 3;
            const __bf16* src = sm + (8 * t + ch) * PO + (2 * l + part) * PK_PJ + 8 * g;
            const bf16x4 lo = *reinterpret_cast<const bf16x4*>(src), hi = *reinterpret_cast<const bf16x4*>(src + 4);
            *reinterpret_cast<bf16x8*>(img + ((((long long)l * NT + o0 / 8 + t) * KS + kk) * 64 + lane) * 8) =
                bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
    }
    // ---- backward image: tiles over i, k over o: lane holds outputs 8 g .. 8 g + 7 of input 8 t + ch
    {
        __bf16* __restrict__ img = pm.bimg[layer];
        const int NT = Ci / 8, KS = Co / 32, kk = o0 / 32;
        for (int e = wv; e < 4 * L; e += 8) {
            const int l = e >> 2, t = e & 3;
            const __bf16* src = sm + (8 * g) * PO + (2 * l + part) * PK_PJ + 8 * t + ch;
            bf16x8 v;
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = src[q * PO];
            *reinterpret_cast<bf16x8*>(img + ((((long long)l * NT + i0 / 8 + t) * KS + kk) * 64 + lane) * 8) = v;
        }
    }
}

struct DhDev {
    const __bf16* X;       // [L][R2][K]
    const __bf16* img;     // [L][NT][KS][64][8]
    __bf16* Y;             // [L][R2][N], N = 8 NT
    int R2, N, NT, M, sparse;
    float sgn;             // +1: Y = X W (forward), -1: gX = gY conj(W)^T
};

template <int K>
__global__ __launch_bounds__(512) void dhconv_apply_kernel(DhDev a) {
    constexpr int KS = K / 32, CPR = K / 8, RPI = 64 / CPR;            // k-steps, 16-byte chunks per row, rows per LDS-DMA instruction
    extern __shared__ __attribute__((aligned(16))) float dh_smem[];
    __bf16* img = reinterpret_cast<__bf16*>(dh_smem);                   // [RC][K], chunk c of row r at c ^ (r & cmask)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    const int tg = blockIdx.x, l = blockIdx.y, row0 = blockIdx.z * RC;
    const int tile = 4 * tg + (w & 3), half = w >> 2;
    auto tile_live = [&](int rt) {                                      // does row tile rt of this chunk hold any non-zero order?
        const int row = row0 + 16 * rt;
        if (row >= a.R2) return false;
        return !a.sparse || ((row % (2 * a.M)) >> 1) <= l;
    };
#ifdef DLWP_STAMPS
    const bool stamp_wg = blockIdx.x == 0 && blockIdx.y == gridDim.y - 1 && blockIdx.z == 0;      // the heaviest degree
#endif
    DLWP_STAMP_IF(stamp_wg, 0);
    // ---- spectrum rows of the live tiles -> LDS (LDS-DMA, swizzle on the source address)
    for (int q = w; q < RC / RPI; q += 8) {
        const int rowi = q * RPI;
        if (!tile_live(rowi >> 4)) continue;
        const int row = rowi + lane / CPR, pos = lane % CPR, c = pos ^ (row & cmask<K>(pos));
        const __bf16* src = a.X + ((long long)l * a.R2 + min(row0 + row, a.R2 - 1)) * K + 8 * c;
        lds_dma16(src, img + rowi * K);
    }
    WFrag<1, KS> wf;
    DLWP_STAMP_IF(stamp_wg, 1);
    if (tile < a.NT) wload<1, KS, KS, 1 << 30>(wf, a.img + (long long)l * a.NT * KS * 512, tile, lane, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DLWP_STAMP_IF(stamp_wg, 2);
    lds_barrier();
    DLWP_STAMP_IF(stamp_wg, 3);
    if (tile >= a.NT) return;
    const int n = 8 * tile + 4 * g;                                     // lanes g < 2 store channels n .. n + 3 of their row
    const float sg = (r & 1) ? a.sgn : -a.sgn;
    for (int rt = 8 * half; rt < 8 * half + 8; ++rt) {
        const long long row = row0 + 16 * rt + r;
        if (row0 + 16 * rt >= a.R2) break;
        bf16x4 out = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        if (tile_live(rt)) {
            f32x4 acc[1][1];
            zero_acc<1, 1>(acc);
            mma<1, 1, KS, K>(acc, wf, img + 16 * rt * K, 0, r, g);
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float oth = __shfl(acc[0][0][q], (lane ^ 1) + 32, 64);      // the other plane's product of the paired row
                v[q] = acc[0][0][q] + sg * oth;
            }
            out = to_bf4(v);
        }
        if (g < 2 && row < a.R2) *reinterpret_cast<bf16x4*>(a.Y + ((long long)l * a.R2 + row) * a.N + n) = out;
    }
    DLWP_STAMP_IF(stamp_wg, 4);
#ifdef DLWP_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DLWP_STAMP_IF(stamp_wg, 5);
#endif
}

// Round 5: the same product as a pipeline.  Stamps of dhconv_apply_kernel at the C3 shape (profiles/r05_sfno_stamps.txt: 20 k cycles
// for the heaviest degree) showed 9.9 k cycles of LDS-DMA issue + landing for the whole 160 KB of a workgroup before the first MFMA,
// then 8 k cycles for eight row tiles in a serially dependent LDS read -> MFMA -> cross-lane read -> store chain, and the 8
// workgroups that share the spectrum rows of a degree were dealt to 8 different XCDs (no L2 reuse).  Here
//   * a workgroup owns (degree, four output tiles) for ALL row chunks and walks them with two LDS buffers: chunk c + 1 lands while
//     chunk c is multiplied; the weight fragments stay in registers for the whole walk (B = 16: read once instead of four times);
//   * a wave multiplies its TPW row tiles of a chunk together (independent accumulators: all fragment reads in flight at once);
//   * the workgroups of one degree have equal blockIdx % 8, i.e. one XCD under round-robin placement (speed only): the rows of
//     a degree come from that XCD's L2 after their first fetch, and every XCD gets degrees of all four truncation bands.
template <int K, int RC>
__global__ __launch_bounds__(512) void dhconv_apply2_kernel(DhDev a) {
    constexpr int KS = K / 32, CPR = K / 8, RPI = 64 / CPR, RT = RC / 16, TPW = RT / 2;
    extern __shared__ __attribute__((aligned(16))) float dh_smem[];
    __bf16* img = reinterpret_cast<__bf16*>(dh_smem);                   // [2][RC][K], chunk c of row r at c ^ (r & cmask)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    const int NTG = a.NT / 4, L = gridDim.x / NTG;
    int tg, l;
    if (L % 8 == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        tg = j % NTG;
        l = (j / NTG) * 8 + xcd;
    } else {
        tg = blockIdx.x % NTG;
        l = blockIdx.x / NTG;
    }
    const int tile = 4 * tg + (w & 3), half = w >> 2;
    const int nch = (a.R2 + RC - 1) / RC;
    // truncation: the row tiles of a sample hold orders 0-7, 8-15, ...; with a power-of-two tile count per sample (the liveness test
    // sits in front of every LDS-DMA: a division there cost 1.7 us per launch) tile t of the sample is live when 8 t <= l
    const int tps = (2 * a.M) / 16;
    const bool sparse = a.sparse && tps > 0 && (tps & (tps - 1)) == 0;
    const int tmask = sparse ? tps - 1 : 0, lt8 = l >> 3;
    auto tile_live = [&](int rt_abs) {                                  // does absolute row tile rt_abs hold any non-zero order?
        if (16 * rt_abs >= a.R2) return false;
        return !sparse || (rt_abs & tmask) <= lt8;
    };
#ifdef DLWP_STAMPS
    const bool stamp_wg = tg == 0 && l == L - 1;                        // the heaviest degree
#endif
    DLWP_STAMP_IF(stamp_wg, 0);
    auto issue = [&](int c) {
        __bf16* buf = img + (c & 1) * RC * K;
#pragma unroll
        for (int i = 0; i < RC / RPI / 8; ++i) {
            const int q = w + 8 * i, rowi = q * RPI;
            if (!tile_live(c * RT + (rowi >> 4))) continue;
            const int row = rowi + lane / CPR, pos = lane % CPR, cc = pos ^ (row & cmask<K>(pos));
            const __bf16* src = a.X + ((long long)l * a.R2 + min(c * RC + row, a.R2 - 1)) * K + 8 * cc;
            lds_dma16(src, buf + rowi * K);
        }
    };
    WFrag<1, KS> wf;
    wload<1, KS, KS, 1 << 30>(wf, a.img + (long long)l * a.NT * KS * 512, tile, lane, 0);
    issue(0);
    DLWP_STAMP_IF(stamp_wg, 1);
    const int n = 8 * tile + 4 * g;                                     // lanes g < 2 store channels n .. n + 3 of their row
    const float sg = (r & 1) ? a.sgn : -a.sgn;
    for (int c = 0; c < nch; ++c) {
        // every vector-memory operation older than the previous chunk's TPW stores has landed: this chunk's rows (and the weights)
        if (c == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TPW) : "memory");
        lds_barrier();                                                  // ... for every wave; and all reads of the other buffer are over
        if (c == 0) DLWP_STAMP_IF(stamp_wg, 2);
        if (c + 1 < nch) issue(c + 1);
        const __bf16* buf = img + (c & 1) * RC * K + 16 * half * TPW * K;
        const int rt0 = c * RT + half * TPW;
        // live tiles of a wave form a prefix when its tiles lie inside one sample (TPW divides the tiles per sample)
        int nl = TPW;
        const bool prefix = sparse && (tps % TPW) == 0;
        if (prefix) {
            nl = 0;
#pragma unroll
            for (int t = 0; t < TPW; ++t) nl += tile_live(rt0 + t) ? 1 : 0;
        }
        f32x4 acc[TPW][1];
        zero_acc<TPW, 1>(acc);
        if (nl == TPW) {
            mma<TPW, 1, KS, K>(acc, wf, buf, 0, r, g);
        } else if (nl > 0) {
            if constexpr (TPW >= 4) {
                if (nl > 2) {
                    f32x4 a3[3][1];
                    zero_acc<3, 1>(a3);
                    mma<3, 1, KS, K>(a3, wf, buf, 0, r, g);
                    acc[0][0] = a3[0][0]; acc[1][0] = a3[1][0]; acc[2][0] = a3[2][0];
                } else if (nl > 1) {
                    f32x4 a2[2][1];
                    zero_acc<2, 1>(a2);
                    mma<2, 1, KS, K>(a2, wf, buf, 0, r, g);
                    acc[0][0] = a2[0][0]; acc[1][0] = a2[1][0];
                } else {
                    f32x4 a1[1][1];
                    zero_acc<1, 1>(a1);
                    mma<1, 1, KS, K>(a1, wf, buf, 0, r, g);
                    acc[0][0] = a1[0][0];
                }
            } else {
                f32x4 a1[1][1];
                zero_acc<1, 1>(a1);
                mma<1, 1, KS, K>(a1, wf, buf, 0, r, g);
                acc[0][0] = a1[0][0];
            }
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const long long row = (long long)16 * (rt0 + t) + r;
            const bool live = prefix ? t < nl : tile_live(rt0 + t);
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float oth = __shfl(acc[t][0][q], (lane ^ 1) + 32, 64);      // the other plane's product of the paired row
                v[q] = live ? acc[t][0][q] + sg * oth : 0.f;
            }
            // (a row past the end exists only in the last chunk, after which nothing is counted any more)
            if (g < 2 && row < a.R2) *reinterpret_cast<bf16x4*>(a.Y + ((long long)l * a.R2 + row) * a.N + n) = to_bf4(v);
        }
    }
    DLWP_STAMP_IF(stamp_wg, 3);
#ifdef DLWP_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DLWP_STAMP_IF(stamp_wg, 4);
    DLWP_STAMP_IF(stamp_wg, 5);
#endif
}

// ---- weight gradient: G[l][i][n] = sum over segments and rows of X[row][i] * gY~[row][n],  gY~ = [gY | gY'] (n < Co | n >= Co)
struct DwDev {
    const __bf16* X[MAXS];       // [L][R2][Ci]
    const __bf16* G[MAXS];       // [L][R2][Co]
    float* out;                  // [L][Ci][2 Co]
    int nseg, R2, Ci, Co, M, sparse, ntn, L;
};
constexpr int GT = 128, KD = 64;

__global__ __launch_bounds__(256) void dhconv_wgrad_kernel(DwDev a) {
    extern __shared__ __attribute__((aligned(16))) float dw_smem[];
    __bf16* lds = reinterpret_cast<__bf16*>(dw_smem);          // [2 stages][A | B][KD][128]
    constexpr int TILE = GT * KD, NI = KD / 16;
    const int lane = lane_id(), w = wave_id(), r = lane & 15, g = lane >> 4;
    // (speed only) the tiles of one degree read the same spectrum rows: one XCD per degree group, as in the apply kernel
    int l, tix;
    {
        const int ntile = gridDim.x / a.L, id = blockIdx.x;
        if (a.L % 8 == 0) {
            const int xcd = id & 7, j = id >> 3;
            tix = j % ntile;
            l = (j / ntile) * 8 + xcd;
        } else {
            tix = id % ntile;
            l = id / ntile;
        }
    }
    const int mt = tix / a.ntn, nt_ = tix - mt * a.ntn;
    const int m0 = mt * GT, n0 = nt_ * GT, Ci = a.Ci, Co = a.Co;
    const bool conj = n0 >= Co;                                // second half of the columns: gY' (pairs swapped, second negated)
    const int b0 = conj ? n0 - Co : n0;
    int krow[NI], acol[NI], bcol[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = (4 * i + w) * 64 + lane, kr = q >> 4, c = (q & 15) ^ (2 * (kr & 3));
        krow[i] = kr;
        acol[i] = min(m0 + 8 * c, Ci - 8);
        bcol[i] = min(b0 + 8 * c, Co - 8);
    }
    const int spk = (a.R2 + KD - 1) / KD, nsteps = a.nseg * spk;       // K-steps per segment / in all
    auto issue = [&](int stage, int it) {
        const int s = it / spk, k0 = (it - s * spk) * KD;
        const __bf16* A = a.X[s] + (long long)l * a.R2 * Ci;
        const __bf16* B = a.G[s] + (long long)l * a.R2 * Co;
        __bf16* As = lds + stage * 2 * TILE;
        __bf16* Bs = As + TILE;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const long long kk = min(k0 + krow[i], a.R2 - 1);
            lds_dma16(A + kk * Ci + acol[i], As + (4 * i + w) * 512);
            lds_dma16(B + kk * Co + bcol[i], Bs + (4 * i + w) * 512);
        }
    };
    const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto frag = [&](const __bf16* tile, int col0, int kk) {      // rows col0 + r of the operand, k = 32 kk + 8 g .. + 7
        const int kr = 32 * kk + 8 * g + (r >> 2), cc = col0 + 4 * (r & 3);
        const __bf16* p0 = tile + kr * GT + 8 * ((cc >> 3) ^ (2 * (kr & 3))) + (cc & 4);
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * GT));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    issue(0, 0);
    for (int it = 0; it < nsteps; ++it) {
        if (it + 1 < nsteps) {
            issue((it + 1) & 1, it + 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const __bf16* As = lds + (it & 1) * 2 * TILE;
        const __bf16* Bs = As + TILE;
        const int k0 = (it % spk) * KD;
#pragma unroll
        for (int kk = 0; kk < KD / 32; ++kk) {
            const int kb = k0 + 32 * kk;
            // rows past the end of the segment are clamped copies: zero them; rows of orders m > l are zero by truncation: skip
            if (kb >= a.R2 || (a.sparse && ((kb % (2 * a.M)) >> 1) > l)) continue;
            bf16x8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = frag(As, wm + 16 * i, kk);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = frag(Bs, wn + 16 * j, kk);
            if (kb + 32 > a.R2) {
                const int kl = a.R2 - (kb + 8 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (e >= kl) af[i][e] = (__bf16)0.f;
            }
            if (conj) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bf16x8 t = bf[j];
                    bf[j] = bf16x8{t[1], -t[0], t[3], -t[2], t[5], -t[4], t[7], -t[6]};
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    float* out = a.out + (long long)l * Ci * 2 * Co;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + wm + i * 16 + 4 * g + q, n = n0 + wn + j * 16 + r;
                if (m < Ci && n < 2 * Co) out[(long long)m * 2 * Co + n] = acc[i][j][q];
            }
}

// gw[i][o][l][part] += G[l][i][part * Co + o]
__global__ __launch_bounds__(256) void dhconv_fold_kernel(const float* __restrict__ G, float* __restrict__ gw, int Ci, int Co, int L) {
    extern __shared__ __attribute__((aligned(16))) float fd_smem[];      // [32 o][2 L + 1]
    const int i = blockIdx.y, o0 = blockIdx.x * 32, tid = threadIdx.x, LD = 2 * L + 1;
    const int no = min(32, Co - o0);
    for (int e = tid; e < L * 2 * 32; e += 256) {
        const int oo = e & 31, part = (e >> 5) & 1, l = e >> 6;
        if (oo < no) fd_smem[oo * LD + 2 * l + part] = G[((long long)l * Ci + i) * 2 * Co + (long long)part * Co + o0 + oo];
    }
    __syncthreads();
    float* dst = gw + ((long long)i * Co + o0) * L * 2;
    for (int e = tid; e < no * 2 * L; e += 256) {
        const int oo = e / (2 * L), j = e - oo * 2 * L;
        dst[e] += fd_smem[oo * LD + j];
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

#ifdef DLWP_STAMPS
extern "C" int dlwp_debug_stamps_dhconv(unsigned long long* host_out) {
    DLWP_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dlwp_stamps), sizeof(unsigned long long) * 32));
    return DLWP_OK;
}
#endif

extern "C" int dlwp_dhconv_supported(int Cin, int Cout, int L) {
    // apply kernel: contraction width with a compiled instantiation, output tiles in groups of four; weight gradient: 128-wide column
    // tiles must not straddle the [gY | gY'] halves; pack / fold slabs within the LDS
    const auto kok = [](int k) { return k == 64 || k == 128 || k == 256; };
    return kok(Cin) && kok(Cout) && Cin % 128 == 0 && Cout % 128 == 0 && L >= 1 && L <= 64;
}

extern "C" long long dlwp_dhconv_image_elems(int Cin, int Cout, int L) { return (long long)L * 2 * Cin * Cout; }

extern "C" int dlwp_dhconv_pack_many(const float* const* w, void* const* fwd_img, void* const* bwd_img, int n, int Cin, int Cout, int L,
                                     void* stream) {
    DLWP_REQUIRE(w && fwd_img && bwd_img && n >= 1 && n <= MAXW, DLWP_E_INVALID, "dhconv_pack: %d weights (1..%d) / null table", n, MAXW);
    DLWP_REQUIRE(dlwp_dhconv_supported(Cin, Cout, L), DLWP_E_UNSUPPORTED, "dhconv_pack: widths %d -> %d, %d degrees unsupported", Cin, Cout, L);
    PackMany pm{};
    for (int i = 0; i < n; ++i) {
        DLWP_REQUIRE(w[i] && fwd_img[i] && bwd_img[i] && aligned16(fwd_img[i]) && aligned16(bwd_img[i]), DLWP_E_INVALID,
                     "dhconv_pack: weight %d: null / unaligned pointer", i);
        pm.w[i] = w[i];
        pm.fimg[i] = static_cast<__bf16*>(fwd_img[i]);
        pm.bimg[i] = static_cast<__bf16*>(bwd_img[i]);
    }
    // one read of every weight for both images where a 32 x 32 block of (i, o) pairs fits the LDS as bf16.  The kernel loads the
    // 16 L float4 pieces of an input row with ONE thread each (tid < per_i, 512 threads): L <= 32, stated here and not left to the
    // LDS budget (L = 34 happens to exceed it today); 32-wide blocks need both channel counts to be multiples of 32
    const size_t lds2 = (size_t)32 * (2 * L * PK_PJ + 2) * sizeof(__bf16);
    if (L % 2 == 0 && 16 * L <= 512 && Cin % 32 == 0 && Cout % 32 == 0 && lds2 <= 150 * 1024 && dlwp_tune_or("DHCONV_PACK", 2) != 1) {
        if (int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(dhconv_pack2_kernel), lds2, "dhconv_pack2")) return rc;
        hipLaunchKernelGGL(dhconv_pack2_kernel, dim3(Cin / 32, Cout / 32, n), dim3(512), lds2, (hipStream_t)stream, pm, Cin, Cout, L);
        DLWP_LAUNCH_CHECK();
        return DLWP_OK;
    }
    const size_t lds = (size_t)256 * (2 * L + 1) * sizeof(float);
    if (int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(dhconv_pack_kernel), lds, "dhconv_pack")) return rc;
    // one grid for every image: x = 32-deep k blocks, y = 8-channel tiles (each image uses its own extents), z = (weight, image)
    const dim3 grid(std::max(Cin, Cout) / 32, std::max(Cin, Cout) / 8, 2 * n);
    hipLaunchKernelGGL(dhconv_pack_kernel, grid, dim3(256), lds, (hipStream_t)stream, pm, Cin, Cout, L);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_dhconv_pack(const float* w, void* fwd_img, void* bwd_img, int Cin, int Cout, int L, void* stream) {
    return dlwp_dhconv_pack_many(&w, &fwd_img, &bwd_img, 1, Cin, Cout, L, stream);
}

// Y[l][row][n] = sum_k X[l][row][k] (*) W   (transposed != 0: gX = gY conj(W)^T with the backward image)
extern "C" int dlwp_dhconv_apply(const void* X, const void* image, void* Y, int L, int rows, int K, int N, int mmax, int transposed,
                                 void* stream) {
    DLWP_REQUIRE(X && image && Y && aligned16(X) && aligned16(image) && aligned16(Y), DLWP_E_INVALID, "dhconv_apply: null / unaligned pointer");
    DLWP_REQUIRE(L >= 1 && rows >= 2 && rows % 2 == 0 && N % 32 == 0 && mmax >= 0, DLWP_E_INVALID,
                 "dhconv_apply: %d degrees, %d rows (re / im pairs), %d output channels (multiple of 32)", L, rows, N);
    // mmax > 0: the rows of a sample are its orders m = 0 .. mmax - 1 and orders m > l are known to be zero (spectra of RealSHT)
    DhDev a{static_cast<const __bf16*>(X), static_cast<const __bf16*>(image), static_cast<__bf16*>(Y), rows, N, N / 8, std::max(mmax, 1),
            mmax > 0 && (2 * mmax) % 16 == 0 && rows % (2 * mmax) == 0, transposed ? -1.f : 1.f};
    const hipStream_t s = (hipStream_t)stream;
    // DHCONV_APPLY = 1: the round-4 kernel (one 256-row chunk per workgroup); default: the pipelined kernel (chunks of 128 rows)
    if (dlwp_tune_or("DHCONV_APPLY", 2) == 1) {
        const dim3 grid(N / 32, L, ceil_div(rows, RC));
#define DLWP_DH(KV)                                                                                                      \
    case KV: {                                                                                                           \
        const size_t lds = (size_t)RC * KV * sizeof(__bf16);                                                             \
        if (int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(dhconv_apply_kernel<KV>), lds, "dhconv_apply")) return rc; \
        hipLaunchKernelGGL(dhconv_apply_kernel<KV>, grid, dim3(512), lds, s, a);                                         \
        break;                                                                                                           \
    }
        switch (K) {
            DLWP_DH(64) DLWP_DH(128) DLWP_DH(256)
            default: dlwp_set_error("dhconv_apply: contraction width %d has no kernel (64, 128, 256)", K); return DLWP_E_UNSUPPORTED;
        }
#undef DLWP_DH
    } else {
        const dim3 grid((N / 32) * L);
#define DLWP_DH2(KV, RCV)                                                                                                \
    case KV: {                                                                                                           \
        const size_t lds = (size_t)2 * RCV * KV * sizeof(__bf16);                                                        \
        auto kern = dhconv_apply2_kernel<KV, RCV>;                                                                       \
        if (int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(kern), lds, "dhconv_apply2")) return rc;              \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, a);                                                            \
        break;                                                                                                           \
    }
        const int rcsel = dlwp_tune_or("DHCONV_RC", 128);
        if (rcsel == 64) {
            switch (K) {
                DLWP_DH2(64, 64) DLWP_DH2(128, 64) DLWP_DH2(256, 64)
                default: dlwp_set_error("dhconv_apply: contraction width %d has no kernel (64, 128, 256)", K); return DLWP_E_UNSUPPORTED;
            }
        } else {
            switch (K) {
                DLWP_DH2(64, 128) DLWP_DH2(128, 128) DLWP_DH2(256, 128)
                default: dlwp_set_error("dhconv_apply: contraction width %d has no kernel (64, 128, 256)", K); return DLWP_E_UNSUPPORTED;
            }
        }
#undef DLWP_DH2
    }
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_dhconv_wgrad(const void* const* X, const void* const* gY, int nseg, float* G, int L, int rows, int Cin, int Cout,
                                 int mmax, void* stream) {
    DLWP_REQUIRE(X && gY && G && nseg >= 1 && nseg <= MAXS && aligned16(G), DLWP_E_INVALID, "dhconv_wgrad: %d segments (1..%d) / null pointer", nseg, MAXS);
    DLWP_REQUIRE(dlwp_dhconv_supported(Cin, Cout, L) && rows >= 2 && rows % 2 == 0 && mmax >= 0, DLWP_E_UNSUPPORTED,
                 "dhconv_wgrad: widths %d -> %d, %d degrees, %d rows unsupported", Cin, Cout, L, rows);
    DwDev a{};
    for (int s = 0; s < nseg; ++s) {
        DLWP_REQUIRE(X[s] && gY[s] && aligned16(X[s]) && aligned16(gY[s]), DLWP_E_INVALID, "dhconv_wgrad: segment %d null / unaligned", s);
        a.X[s] = static_cast<const __bf16*>(X[s]);
        a.G[s] = static_cast<const __bf16*>(gY[s]);
    }
    a.out = G; a.nseg = nseg; a.R2 = rows; a.Ci = Cin; a.Co = Cout; a.M = std::max(mmax, 1);
    a.sparse = mmax > 0 && (2 * mmax) % 32 == 0 && rows % (2 * mmax) == 0;
    a.ntn = 2 * Cout / GT;
    a.L = L;
    const size_t lds = (size_t)2 * 2 * GT * KD * sizeof(__bf16);
    if (int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(dhconv_wgrad_kernel), lds, "dhconv_wgrad")) return rc;
    hipLaunchKernelGGL(dhconv_wgrad_kernel, dim3((Cin / GT) * a.ntn * L), dim3(256), lds, (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_dhconv_fold(const float* G, float* gw, int Cin, int Cout, int L, void* stream) {
    DLWP_REQUIRE(G && gw && Cin >= 1 && Cout >= 1 && L >= 1 && Cin <= 65535, DLWP_E_INVALID, "dhconv_fold: bad argument");
    const size_t lds = (size_t)32 * (2 * L + 1) * sizeof(float);
    if (int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(dhconv_fold_kernel), lds, "dhconv_fold")) return rc;
    hipLaunchKernelGGL(dhconv_fold_kernel, dim3(ceil_div(Cout, 32), Cin), dim3(256), lds, (hipStream_t)stream, G, gw, Cin, Cout, L);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

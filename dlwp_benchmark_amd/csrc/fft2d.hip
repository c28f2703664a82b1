// rFFT2 / irFFT2 of real fields, any size, with the butterflies staged in LDS.
// Reference call sites: torch.fft.rfft2(x, dim=(1, 2), norm="ortho") / irfft2 over the MIDDLE dims of a channels-last
// [B, H, W, C] tensor in AFNO2D.forward (src/nsbench/models/fourcastnet/fourcastnet.py:84,123; dlwpbench twin :85,124),
// and rfftn / irfftn(norm="forward") over the last two dims of [B, C, H, W] inside neuralop's SpectralConv (third party,
// SURVEY.md App. A-1).  The pruned-DFT GEMMs of fno_block.hip / afno.hip / afno_tiled.py cover mode-truncated layers on
// small grids; this is the general path: patch-1 AFNO on 128 x 256 or 721 x 1440 (721 = 7 * 103) needs all modes of a grid
// where a dense DFT is quadratic.
//
// MI355X design.  A 2-D transform is two passes over HBM (W axis, then H axis), each an HBM-bound streaming kernel:
//   * one workgroup = one signal position range x IB "inner" lanes that are CONTIGUOUS in memory (channels-last: channels;
//     channels-first: the kept columns), so every global access of a pass is a coalesced row of IB complex numbers;
//   * the workgroup's [N][IB] complex tile lives in LDS (row pitch IB + 1: conflict-free column walks); a mixed-radix
//     Stockham FFT runs on it IN PLACE: per pass every thread accumulates its outputs in registers (one output = R complex
//     multiply-adds against the twiddle table, also in LDS), barrier, writes them back, barrier.  Radices come from the
//     factorisation of N (4, 2, 3, 5, 7, then whatever prime is left: 721 -> 7 x 103 costs 110 multiply-adds per point, still
//     linear in N); no transposes, no global scratch between the butterfly passes;
//   * real transforms pack TWO real signals into one complex one (two adjacent channels channels-last: a float2 load IS the
//     packed sample; two adjacent rows channels-first) and split / merge the Hermitian halves on the way out / in;
//   * normalisation, the Hermitian weights of the adjoint transforms (interior bins count twice) and the
//     real-part selection ride in the store / load stages: no extra pass for the backward transforms.
#include "common.hip.h"
#include "dlwpmi_internal.h"
#include <cmath>
#include <vector>

namespace {

constexpr int MAXRAD = 16;

struct FftAxis {
    int N, nrad, IB, logIB;          // signal length, number of passes, inner lanes per workgroup (power of two)
    int rad[MAXRAD];
    FastDiv dp[MAXRAD], dM[MAXRAD];  // division by p (product of the earlier radices) and by M = p * R
    const float2* tab;               // [N] e^{-2 pi i n / N} (device)
    // last radix R >= 16 (a large prime: 103 of 721): its DFT as a real GEMM on the matrix cores.  amat: [2][2 RP][2 RP] floats
    // (device; forward, inverse), the real image [[Fr, -Fi], [Fi, Fr]] of F[q][r] = e^{sg 2 pi i q r / R} padded to RP = 8-multiple
    const float* amat;
    int rp;
    int sp;                          // compile-time plan (SPlan<sp>) this axis runs on, 0 = the run-time plan
    int sym;                         // a single odd prime radix <= 127 on pass_prime_sym (amat = its cos / sin matrices)
};

__device__ __forceinline__ float2 cmul(float2 a, float2 w) { return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }
// multiply by sg * i  (sg = -1: forward transform's -i, +1: inverse's +i)
__device__ __forceinline__ float2 muli(float2 a, float sg) { return make_float2(-sg * a.y, sg * a.x); }

// DFT of length R in registers, v[q] = sum_r u[r] e^{sg 2 pi i q r / R}  (R = 2, 3, 4, 5)
template <int R>
__device__ __forceinline__ void dft_small(float2 (&u)[R], float sg) {
    if constexpr (R == 2) {
        const float2 a = u[0], b = u[1];
        u[0] = make_float2(a.x + b.x, a.y + b.y);
        u[1] = make_float2(a.x - b.x, a.y - b.y);
    } else if constexpr (R == 4) {
        const float2 s02 = make_float2(u[0].x + u[2].x, u[0].y + u[2].y), d02 = make_float2(u[0].x - u[2].x, u[0].y - u[2].y);
        const float2 s13 = make_float2(u[1].x + u[3].x, u[1].y + u[3].y), d13 = muli(make_float2(u[1].x - u[3].x, u[1].y - u[3].y), sg);
        u[0] = make_float2(s02.x + s13.x, s02.y + s13.y);
        u[1] = make_float2(d02.x + d13.x, d02.y + d13.y);
        u[2] = make_float2(s02.x - s13.x, s02.y - s13.y);
        u[3] = make_float2(d02.x - d13.x, d02.y - d13.y);
    } else if constexpr (R == 3) {
        const float c = -0.5f, s_ = sg * 0.86602540378443864676f;
        const float2 t1 = make_float2(u[1].x + u[2].x, u[1].y + u[2].y), t2 = make_float2(u[1].x - u[2].x, u[1].y - u[2].y);
        const float2 m = make_float2(u[0].x + c * t1.x, u[0].y + c * t1.y), n = make_float2(-s_ * t2.y, s_ * t2.x);   // i s t2
        u[0] = make_float2(u[0].x + t1.x, u[0].y + t1.y);
        u[1] = make_float2(m.x + n.x, m.y + n.y);
        u[2] = make_float2(m.x - n.x, m.y - n.y);
    } else {
        static_assert(R == 5, "register butterflies exist for radix 2, 3, 4, 5");
        const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
        const float s1 = sg * 0.95105651629515357212f, s2 = sg * 0.58778525229247312917f;
        const float2 a14 = make_float2(u[1].x + u[4].x, u[1].y + u[4].y), b14 = make_float2(u[1].x - u[4].x, u[1].y - u[4].y);
        const float2 a23 = make_float2(u[2].x + u[3].x, u[2].y + u[3].y), b23 = make_float2(u[2].x - u[3].x, u[2].y - u[3].y);
        const float2 m1 = make_float2(u[0].x + c1 * a14.x + c2 * a23.x, u[0].y + c1 * a14.y + c2 * a23.y);
        const float2 m2 = make_float2(u[0].x + c2 * a14.x + c1 * a23.x, u[0].y + c2 * a14.y + c1 * a23.y);
        const float2 n1 = make_float2(-(s1 * b14.y + s2 * b23.y), s1 * b14.x + s2 * b23.x);      // i (s1 b14 + s2 b23)
        const float2 n2 = make_float2(-(s2 * b14.y - s1 * b23.y), s2 * b14.x - s1 * b23.x);      // i (s2 b14 - s1 b23)
        u[0] = make_float2(u[0].x + a14.x + a23.x, u[0].y + a14.y + a23.y);
        u[1] = make_float2(m1.x + n1.x, m1.y + n1.y);
        u[4] = make_float2(m1.x - n1.x, m1.y - n1.y);
        u[2] = make_float2(m2.x + n2.x, m2.y + n2.y);
        u[3] = make_float2(m2.x - n2.x, m2.y - n2.y);
    }
}

// e^{2 pi i j / N} for the composite register butterflies (literals: the index is a compile-time constant after unrolling)
template <int N> struct TwC;
template <> struct TwC<6> {
    static __device__ __forceinline__ float2 get(int j) {
        switch (j) {
            case 0: return make_float2(1.f, 0.f);
            case 1: return make_float2(0.50000000000000011f, 0.8660254037844386f);
            case 2: return make_float2(-0.49999999999999978f, 0.86602540378443871f);
            case 3: return make_float2(-1.f, 0.f);
            case 4: return make_float2(-0.50000000000000044f, -0.86602540378443837f);
            case 5: return make_float2(0.50000000000000011f, -0.8660254037844386f);
            default: return make_float2(1.f, 0.f);
        }
    }
};
template <> struct TwC<9> {
    static __device__ __forceinline__ float2 get(int j) {
        switch (j) {
            case 0: return make_float2(1.f, 0.f);
            case 1: return make_float2(0.76604444311897801f, 0.64278760968653925f);
            case 2: return make_float2(0.17364817766693041f, 0.98480775301220802f);
            case 3: return make_float2(-0.49999999999999978f, 0.86602540378443871f);
            case 4: return make_float2(-0.93969262078590832f, 0.34202014332566888f);
            case 5: return make_float2(-0.93969262078590843f, -0.34202014332566866f);
            case 6: return make_float2(-0.50000000000000044f, -0.86602540378443837f);
            case 7: return make_float2(0.17364817766692997f, -0.98480775301220813f);
            case 8: return make_float2(0.76604444311897779f, -0.64278760968653958f);
            default: return make_float2(1.f, 0.f);
        }
    }
};
template <> struct TwC<10> {
    static __device__ __forceinline__ float2 get(int j) {
        switch (j) {
            case 0: return make_float2(1.f, 0.f);
            case 1: return make_float2(0.80901699437494745f, 0.58778525229247314f);
            case 2: return make_float2(0.30901699437494745f, 0.95105651629515353f);
            case 3: return make_float2(-0.30901699437494734f, 0.95105651629515364f);
            case 4: return make_float2(-0.80901699437494734f, 0.58778525229247325f);
            case 5: return make_float2(-1.f, 0.f);
            case 6: return make_float2(-0.80901699437494756f, -0.58778525229247303f);
            case 7: return make_float2(-0.30901699437494756f, -0.95105651629515353f);
            case 8: return make_float2(0.30901699437494723f, -0.95105651629515364f);
            case 9: return make_float2(0.80901699437494734f, -0.58778525229247336f);
            default: return make_float2(1.f, 0.f);
        }
    }
};
template <> struct TwC<12> {
    static __device__ __forceinline__ float2 get(int j) {
        switch (j) {
            case 0: return make_float2(1.f, 0.f);
            case 1: return make_float2(0.86602540378443871f, 0.49999999999999994f);
            case 2: return make_float2(0.50000000000000011f, 0.8660254037844386f);
            case 3: return make_float2(0.f, 1.f);
            case 4: return make_float2(-0.49999999999999978f, 0.86602540378443871f);
            case 5: return make_float2(-0.86602540378443871f, 0.49999999999999994f);
            case 6: return make_float2(-1.f, 0.f);
            case 7: return make_float2(-0.86602540378443882f, -0.49999999999999972f);
            case 8: return make_float2(-0.50000000000000044f, -0.86602540378443837f);
            case 9: return make_float2(0.f, -1.f);
            case 10: return make_float2(0.50000000000000011f, -0.8660254037844386f);
            case 11: return make_float2(0.86602540378443837f, -0.50000000000000044f);
            default: return make_float2(1.f, 0.f);
        }
    }
};
template <> struct TwC<15> {
    static __device__ __forceinline__ float2 get(int j) {
        switch (j) {
            case 0: return make_float2(1.f, 0.f);
            case 1: return make_float2(0.91354545764260087f, 0.40673664307580015f);
            case 2: return make_float2(0.66913060635885824f, 0.74314482547739413f);
            case 3: return make_float2(0.30901699437494745f, 0.95105651629515353f);
            case 4: return make_float2(-0.10452846326765333f, 0.9945218953682734f);
            case 5: return make_float2(-0.49999999999999978f, 0.86602540378443871f);
            case 6: return make_float2(-0.80901699437494734f, 0.58778525229247325f);
            case 7: return make_float2(-0.97814760073380569f, 0.20791169081775931f);
            case 8: return make_float2(-0.97814760073380569f, -0.20791169081775907f);
            case 9: return make_float2(-0.80901699437494756f, -0.58778525229247303f);
            case 10: return make_float2(-0.50000000000000044f, -0.86602540378443837f);
            case 11: return make_float2(-0.10452846326765423f, -0.99452189536827329f);
            case 12: return make_float2(0.30901699437494723f, -0.95105651629515364f);
            case 13: return make_float2(0.66913060635885846f, -0.74314482547739402f);
            case 14: return make_float2(0.91354545764260098f, -0.40673664307580015f);
            default: return make_float2(1.f, 0.f);
        }
    }
};

// DFT of length A * B in registers from the two small ones (Cooley-Tukey inside the register file): n = B n1 + n2, k = k1 + A k2,
//   X[k1 + A k2] = sum_{n2} W_B^{n2 k2} ( W_N^{n2 k1} sum_{n1} x[B n1 + n2] W_A^{n1 k1} ),   W_M = e^{sg 2 pi i / M}
template <int A, int B>
__device__ __forceinline__ void dft_comp(float2 (&u)[A * B], float sg) {
    constexpr int N = A * B;
    float2 y[N];
#pragma unroll
    for (int n2 = 0; n2 < B; ++n2) {
        float2 t[A];
#pragma unroll
        for (int n1 = 0; n1 < A; ++n1) t[n1] = u[B * n1 + n2];
        dft_small<A>(t, sg);
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1) {
            const int m = (n2 * k1) % N;
            if (m == 0) {
                y[k1 * B + n2] = t[k1];
            } else {
                float2 w = TwC<N>::get(m);
                w.y *= sg;
                y[k1 * B + n2] = cmul(t[k1], w);
            }
        }
    }
#pragma unroll
    for (int k1 = 0; k1 < A; ++k1) {
        float2 t[B];
#pragma unroll
        for (int n2 = 0; n2 < B; ++n2) t[n2] = y[k1 * B + n2];
        dft_small<B>(t, sg);
#pragma unroll
        for (int k2 = 0; k2 < B; ++k2) u[k1 + A * k2] = t[k2];
    }
}
template <int R>
__device__ __forceinline__ void dft_reg(float2 (&u)[R], float sg) {
    if constexpr (R <= 5) dft_small<R>(u, sg);
    else if constexpr (R == 6) dft_comp<2, 3>(u, sg);
    else if constexpr (R == 9) dft_comp<3, 3>(u, sg);
    else if constexpr (R == 10) dft_comp<2, 5>(u, sg);
    else if constexpr (R == 12) dft_comp<3, 4>(u, sg);
    else { static_assert(R == 15, "register butterflies: 2 - 6, 9, 10, 12, 15"); dft_comp<3, 5>(u, sg); }
}

// ---- plans fixed at compile time (round 5).  The run-time plan above costs ~27 VALU instructions per point and pass (index
// arithmetic through FastDiv structures, a run-time radix switch inside the pass loop, 60 % of the lanes busy in the partial
// rounds): with the butterfly passes skipped (DLWP_FFT_SKIP_PASSES) rfft2 of 90 x 180 x 768 takes 50.7 us of its 84.4.  For the
// axis lengths of the FourCastNet token grid the whole plan is a template: signal length, inner lanes, thread count and two
// large radices (composite register butterflies) are constants, every index is a shift / constant multiply, the first pass has
// no twiddles, and lanes x butterflies per pass nearly fill the workgroup.
template <int SP> struct SPlan;
template <> struct SPlan<1> { static constexpr int N = 180, LOGIB = 4, NT = 256, R0 = 12, R1 = 15; };     // 15 x 16 = 240, 12 x 16 = 192 items
template <> struct SPlan<2> { static constexpr int N = 90, LOGIB = 5, NT = 320, R0 = 9, R1 = 10; };       // 10 x 32 = 320, 9 x 32 = 288 items
// measured alternatives (DLWP_FFT_SPW / DLWP_FFT_SPH pick a plan by number)
template <> struct SPlan<3> { static constexpr int N = 180, LOGIB = 4, NT = 256, R0 = 15, R1 = 12; };
template <> struct SPlan<4> { static constexpr int N = 90, LOGIB = 5, NT = 320, R0 = 10, R1 = 9; };
template <> struct SPlan<5> { static constexpr int N = 180, LOGIB = 5, NT = 512, R0 = 12, R1 = 15; };
template <> struct SPlan<6> { static constexpr int N = 90, LOGIB = 4, NT = 192, R0 = 9, R1 = 10; };

// one Stockham pass of radix R, P = product of the earlier radices (pass_small with every plan quantity a constant)
template <int N, int LOGIB, int NT, int R, int P>
__device__ __forceinline__ void pass_c(float2* buf, const float2* tabs, float sg) {
    constexpr int IB = 1 << LOGIB, IBP = IB + 1, T = N / R, M = P * R, STEP = N / M, NITEMS = T * IB, NB = (NITEMS + NT - 1) / NT;
    static_assert(N % (P * R) == 0, "radices must divide the signal length");
    const int tid = threadIdx.x;
    float2 u[NB][R];
#pragma unroll
    for (int ub = 0; ub < NB; ++ub) {
        const int e = tid + ub * NT;
        if ((ub + 1) * NT <= NITEMS || e < NITEMS) {
            const int lane = e & (IB - 1), i = e >> LOGIB, blk = i / P, k = i - blk * P;
            const float2* xin = buf + i * IBP + lane;
#pragma unroll
            for (int r = 0; r < R; ++r) u[ub][r] = xin[r * T * IBP];
            if constexpr (P > 1) {
#pragma unroll
                for (int r = 1; r < R; ++r) {
                    float2 w = tabs[r * k * STEP];
                    w.y *= -sg;                                  // table holds e^{-i theta}
                    u[ub][r] = cmul(u[ub][r], w);
                }
            }
            dft_reg<R>(u[ub], sg);
        }
    }
    __syncthreads();
#pragma unroll
    for (int ub = 0; ub < NB; ++ub) {
        const int e = tid + ub * NT;
        if ((ub + 1) * NT <= NITEMS || e < NITEMS) {
            const int lane = e & (IB - 1), i = e >> LOGIB, blk = i / P, k = i - blk * P;
            float2* yout = buf + (blk * M + k) * IBP + lane;
#pragma unroll
            for (int q = 0; q < R; ++q) yout[q * P * IBP] = u[ub][q];
        }
    }
    __syncthreads();
}
template <int SP>
__device__ __forceinline__ void lds_fft_static(float2* buf, const float2* tabs, float sg) {
    using S = SPlan<SP>;
    static_assert(S::R0 * S::R1 == S::N, "two-pass plans");
    pass_c<S::N, S::LOGIB, S::NT, S::R0, 1>(buf, tabs, sg);
    pass_c<S::N, S::LOGIB, S::NT, S::R1, S::R0>(buf, tabs, sg);
}

// One Stockham pass of radix R with the butterflies in registers: work item = (butterfly i < N / R, lane); inputs
// x[i + r N / R] times the twiddles w^(r k) (k = i mod p), outputs y[blk p R + q p + k].  LDS traffic per output: one data read
// plus (R - 1) / R table reads -- against 2 R for the generic pass below.
template <int R, int OUTS, int NT>
__device__ __forceinline__ void pass_small(float2* buf, const float2* tabs, const FftAxis& f, int s, int p, float sg) {
    constexpr int NB = (OUTS + R - 1) / R;
    const int IBP = f.IB + 1, t = f.N / R, M = p * R, step = f.N / M, nitems = t << f.logIB, tid = threadIdx.x;
    float2 u[NB][R];
    int obase[NB];
#pragma unroll
    for (int ub = 0; ub < NB; ++ub) {
        const int e = tid + ub * NT;
        obase[ub] = -1;
        if (e < nitems) {
            const int lane = e & (f.IB - 1), i = e >> f.logIB;
            const int blk = fastdiv(i, f.dp[s]), k = i - blk * p;
            const float2* xin = buf + i * IBP + lane;
#pragma unroll
            for (int r = 0; r < R; ++r) u[ub][r] = xin[r * t * IBP];
#pragma unroll
            for (int r = 1; r < R; ++r) {
                float2 w = tabs[r * k * step];
                w.y *= -sg;                                  // table holds e^{-i theta}
                u[ub][r] = cmul(u[ub][r], w);
            }
            dft_small<R>(u[ub], sg);
            obase[ub] = (blk * M + k) * IBP + lane;
        }
    }
    __syncthreads();
#pragma unroll
    for (int ub = 0; ub < NB; ++ub)
        if (obase[ub] >= 0) {
#pragma unroll
            for (int q = 0; q < R; ++q) buf[obase[ub] + q * p * IBP] = u[ub][q];
        }
    __syncthreads();
}

// Last Stockham pass for a LARGE prime radix R (p = N / R butterflies per lane) on the matrix cores.  The generic pass costs R
// complex multiply-adds per output on the VALU with two LDS reads each (721 = 7 x 103: 110 per point, 2.2 of the 2.4 ms of a
// 721 x 1440 x 64 transform; with this pass the transform takes 1.33 ms, the pass itself now bound by the fp32 MFMA rate:
// 28 GFLOP of real-GEMM work per 721-row pass).  Here: (1) the inputs take their twiddles w^(r k) in place; (2) out[q p + k] = sum_r F[q][r]
// in[k + p r] is the real GEMM  [Yr; Yi] = [[Fr, -Fi], [Fi, Fr]] . [Xr; Xi]  with M = K = 2 RP rows and one column per (k, lane):
// A fragments come from the precomputed matrix in L2 (16 bytes per lane and 16-deep chunk, every wave owns one or two 16-row
// panels and re-uses its fragment for all column tiles), B fragments from the LDS tile; (3) the accumulators overwrite the tile.
template <int NT>
__device__ __forceinline__ void pass_prime_mfma(float2* buf, const float2* tabs, const FftAxis& f, int s, int p, float sg) {
    constexpr int NW = NT / 64, GNT = 4;            // column tiles per group: 32 accumulator registers
    const int R = f.rad[s], RP = f.rp, K2 = 2 * RP, IBP = f.IB + 1, tid = threadIdx.x;
    const int ncols = p << f.logIB, ntn = (ncols + 15) >> 4, ntm = K2 >> 4;
    const int lane = tid & 63, wv = tid >> 6, rl = lane & 15, g = lane >> 4;
    for (int e = tid; e < f.N << f.logIB; e += NT) {
        const int ln = e & (f.IB - 1), idx = e >> f.logIB;
        const int r = fastdiv(idx, f.dp[s]), k = idx - r * p;
        if (r && k) {
            float2 w = tabs[r * k];              // r k < N: the last pass has M = N, step = 1
            w.y *= -sg;                          // table holds e^{-i theta}
            buf[idx * IBP + ln] = cmul(buf[idx * IBP + ln], w);
        }
    }
    const float* A = f.amat + (sg > 0.f ? (long long)K2 * K2 : 0);
    const int mt0 = wv, mt1 = wv + NW;
    const bool has0 = mt0 < ntm, has1 = mt1 < ntm;
    // this wave's one or two 16-row panels of the matrix: fragments come from L2 with the next two chunks' loads in flight
    const float* A0 = A + (long long)(16 * mt0 + rl) * K2 + 4 * g;
    const float* A1 = A + (long long)(16 * (has1 ? mt1 : mt0) + rl) * K2 + 4 * g;
    __syncthreads();                                  // twiddled inputs visible
    // Columns (k, lane) are independent -- column c reads and writes only rows {k + p r} of lane c -- so the tile is updated in
    // place one group of GNT column tiles at a time: compute the group (all waves), barrier, overwrite, barrier.
    for (int nt0 = 0; nt0 < ntn; nt0 += GNT) {
        f32x4 acc[2][GNT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < GNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (has0) {
            f32x4 p0 = *reinterpret_cast<const f32x4*>(A0), p1 = *reinterpret_cast<const f32x4*>(A1);
            f32x4 q0 = p0, q1 = p1;
            if (ntm > 1) { q0 = *reinterpret_cast<const f32x4*>(A0 + 16); q1 = *reinterpret_cast<const f32x4*>(A1 + 16); }
#pragma unroll 1
            for (int kc = 0; kc < ntm; ++kc) {
                const f32x4 a0 = p0, a1 = p1;
                p0 = q0; p1 = q1;
                if (kc + 2 < ntm) {
                    q0 = *reinterpret_cast<const f32x4*>(A0 + 16 * (kc + 2));
                    q1 = *reinterpret_cast<const f32x4*>(A1 + 16 * (kc + 2));
                }
                int roff[4];
                bool im[4], okk[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {     // this lane's four K slots: plane (re | im) and input index r
                    const int kk = 16 * kc + 4 * g + q;
                    im[q] = kk >= RP;
                    const int r = kk - (im[q] ? RP : 0);
                    okk[q] = r < R;
                    roff[q] = p * min(r, R - 1) * IBP;
                }
#pragma unroll
                for (int j = 0; j < GNT; ++j) {
                    const int nt = nt0 + j;
                    if (nt < ntn) {
                        const int n = 16 * nt + rl, nc = min(n, ncols - 1);
                        const int k = nc >> f.logIB, ln = nc & (f.IB - 1);
                        const float* col = reinterpret_cast<const float*>(buf + k * IBP + ln);
                        f32x4 b;
#pragma unroll
                        for (int q = 0; q < 4; ++q) b[q] = (okk[q] && n < ncols) ? col[2 * roff[q] + (im[q] ? 1 : 0)] : 0.f;
                        acc[0][j] = mfma16_chunk(a0, b, acc[0][j]);
                        if (has1) acc[1][j] = mfma16_chunk(a1, b, acc[1][j]);
                    }
                }
            }
        }
        __syncthreads();                              // every wave is done reading this column group
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mt = i ? mt1 : mt0;
            if (mt < ntm) {
#pragma unroll
                for (int j = 0; j < GNT; ++j) {
                    const int n = 16 * (nt0 + j) + rl;
                    if (nt0 + j < ntn && n < ncols) {
                        const int k = n >> f.logIB, ln = n & (f.IB - 1);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const int m = 16 * mt + 4 * g + jj, pl = m >= RP ? 1 : 0, q = m - pl * RP;
                            if (q < R) reinterpret_cast<float*>(buf + (q * p + k) * IBP + ln)[pl] = acc[i][j][jj];
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---- a whole axis of odd prime length N <= 127 as ONE dense pass that uses the symmetries of the DFT matrix (round 5).
// pass_prime_mfma multiplies by the real image [[Fr, -Fi], [Fi, Fr]] of the full N x N matrix: 4 N^2 real multiply-adds per column.
// With h = (N - 1) / 2, e[r] = x[r] + x[N - r], o[r] = x[r] - x[N - r] (r = 1 .. h), e[0] = x[0]:
//     X[q]     = sum_{r <= h} cos(2 pi q r / N) e[r]  +  i sg sum_{r <= h} sin(2 pi q r / N) o[r]  =  C[q] + i sg S[q]
//     X[N - q] =                                                                                      C[q] - i sg S[q]     (q = 0 .. h)
// i.e. two REAL (h + 1) x (h + 1) matrices applied to the re and im planes: (h + 1)^2 * 4 multiply-adds per column, a quarter, and
// the same two matrices serve both directions.  103 (= 721 / 7, the latitude axis of the 721 x 1440 patch grid): the pass went from
// 56.5 k to [see profiles/r05_experiments.md] cycles per workgroup.
// Layout: buf [N][IB + 1] complex, IB = 64 lanes, 512 threads = 8 waves; wave w owns lanes 16 (w & 3) .. + 15 (both planes) and the
// 32 output rows q = 32 (w >> 2) .. + 31; amat = [2][64][64] floats: cos then sin matrix, rows q, columns r, zero beyond h.
template <int NT>
__device__ __forceinline__ void pass_prime_sym(float2* buf, const FftAxis& f, float sg) {
    static_assert(NT == 512, "eight waves: four lane tiles x two row halves");
    const int N = f.N, h = (N - 1) >> 1, IBP = f.IB + 1, tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, rl = lane & 15, g = lane >> 4;
    // fold in place: rows 1 .. h receive e, rows N - r receive o[r]; one thread per (r, lane) pair reads and writes both rows
    for (int e = tid; e < h << 6; e += NT) {
        const int ln = e & 63, r = 1 + (e >> 6);
        const float2 a = buf[r * IBP + ln], b = buf[(N - r) * IBP + ln];
        buf[r * IBP + ln] = make_float2(a.x + b.x, a.y + b.y);
        buf[(N - r) * IBP + ln] = make_float2(a.x - b.x, a.y - b.y);
    }
    __syncthreads();
    const int lt = wv & 3, mh = wv >> 2, ln = 16 * lt + rl;
    const float* Cm = f.amat;
    const float* Sm = f.amat + 64 * 64;
    f32x4 acc[2][2][2];                       // [cos | sin][row tile of the half][re | im]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) acc[a][i][pl] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
        f32x4 ac[2], as_[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 16 * (2 * mh + i) + rl;
            ac[i] = *reinterpret_cast<const f32x4*>(Cm + row * 64 + 16 * kc + 4 * g);
            as_[i] = *reinterpret_cast<const f32x4*>(Sm + row * 64 + 16 * kc + 4 * g);
        }
        f32x4 be[2], bo[2];                   // [re | im] of e[k] and o[k], k = 16 kc + 4 g + q
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 16 * kc + 4 * g + q;
            // rows beyond h meet zero matrix columns: any finite row will do
            const float2 ev = buf[min(k, N - 1) * IBP + ln];
            const float2 ov = buf[(k >= 1 && k <= h ? N - k : 0) * IBP + ln];
            be[0][q] = ev.x; be[1][q] = ev.y;
            bo[0][q] = ov.x; bo[1][q] = ov.y;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                acc[0][i][pl] = mfma16_chunk(ac[i], be[pl], acc[0][i][pl]);
                acc[1][i][pl] = mfma16_chunk(as_[i], bo[pl], acc[1][i][pl]);
            }
    }
    __syncthreads();                          // every wave has read its operands: the tile may be overwritten
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int q = 16 * (2 * mh + i) + 4 * g + jj;
            if (q <= h) {
                const float cx = acc[0][i][0][jj], cy = acc[0][i][1][jj], sx = sg * acc[1][i][0][jj], sy = sg * acc[1][i][1][jj];
                buf[q * IBP + ln] = make_float2(cx - sy, cy + sx);                       // C + i sg S
                if (q) buf[(N - q) * IBP + ln] = make_float2(cx + sy, cy - sx);          // C - i sg S
            }
        }
    __syncthreads();
}

// In-place mixed-radix Stockham FFT of buf[N][IB + 1] (complex, LDS); tabs [N] twiddles in LDS; sg = -1 forward, +1 inverse.
// Radices 2, 3, 4, 5 use register butterflies (pass_small); any other radix (7, 11, ... 103 ...) the generic pass: one output
// per work item, m = blk * M + q * p + k  <-  sum_r in[(blk * p + k) + r * N / R] * w^(r * (k + q * p)),  w = e^{sg 2 pi i / M}.
template <int OUTS, int NT, bool PRIME = false>
__device__ __forceinline__ void lds_fft(float2* buf, const float2* tabs, const FftAxis& f, float sg) {
    const int IBP = f.IB + 1, total = f.N << f.logIB, tid = threadIdx.x;
    int p = 1;
    for (int s = 0; s < f.nrad; ++s) {
        const int R = f.rad[s], t = f.N / R, M = p * R, step = f.N / M;
        if (R == 4) { pass_small<4, OUTS, NT>(buf, tabs, f, s, p, sg); p = M; DLWP_STAMP(16 + s); continue; }
        if (R == 2) { pass_small<2, OUTS, NT>(buf, tabs, f, s, p, sg); p = M; DLWP_STAMP(16 + s); continue; }
        if (R == 3) { pass_small<3, OUTS, NT>(buf, tabs, f, s, p, sg); p = M; DLWP_STAMP(16 + s); continue; }
        if (R == 5) { pass_small<5, OUTS, NT>(buf, tabs, f, s, p, sg); p = M; DLWP_STAMP(16 + s); continue; }
        if constexpr (PRIME) {      // instantiated only for plans with a large last prime: the pass costs 64 accumulator registers
            if constexpr (NT == 512) {
                if (f.sym) { pass_prime_sym<NT>(buf, f, sg); p = M; continue; }
            }
            if (s == f.nrad - 1) { pass_prime_mfma<NT>(buf, tabs, f, s, p, sg); p = M; continue; }
        }
        float2 acc[OUTS];
#pragma unroll
        for (int u = 0; u < OUTS; ++u) {
            const int e = tid + u * NT;
            float ar = 0.f, ai = 0.f;
            if (e < total) {
                const int lane = e & (f.IB - 1), m = e >> f.logIB;
                const int blk = fastdiv(m, f.dM[s]), rem = m - blk * M, q = fastdiv(rem, f.dp[s]), k = rem - q * p;
                const float2* xin = buf + (blk * p + k) * IBP + lane;
                const int cs = (k + q * p) * step, tstride = t * IBP;
                int es = 0;
                for (int r = 0; r < R; ++r) {
                    const float2 v = xin[r * tstride];
                    const float2 w = tabs[es];
                    const float wy = sg * w.y;          // table holds e^{-i...}: sg = -1 keeps it, +1 conjugates
                    ar += v.x * w.x + v.y * wy;         // v * (w.x - i wy)  with wy = sg * w.y and w.y = -sin
                    ai += v.y * w.x - v.x * wy;
                    es += cs;
                    if (es >= f.N) es -= f.N;
                }
            }
            acc[u] = make_float2(ar, ai);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < OUTS; ++u) {
            const int e = tid + u * NT;
            if (e < total) buf[(e >> f.logIB) * IBP + (e & (f.IB - 1))] = acc[u];
        }
        __syncthreads();
        p = M;
    }
}

struct FftIO {
    FftAxis ax;
    const float* in; float* out;
    const float* res;         // channels-last C2R: added to the output (a residual connection around the transform pair), or null
    const float* res2;        // ... a second field added as well (the block's outer skip, dlwp_irfft2_planar2), or null; needs res
    long long outer;          // number of outer units (grid.y)
    int C, H, W, Wc;          // C: channels (CL) / unused (CF)
    long long nrows;          // CF real passes: B * C * H rows
    long long J;              // C2C: inner contiguous length (complex), axis stride = J
    float scale, w_int;       // output scale; weight of the interior (non-DC, non-Nyquist) bins
    float sg;                 // C2C direction
    // C2C: plane == 0: interleaved (re, im) pairs, element (o, n, j) at (o * N + n) * J + j.  Otherwise that side is a planar
    // WINDOW of the spectrum, rows r0 <= n < r1 and inner positions j < Jw: re at (o * (r1 - r0) + n - r0) * Jw + j, im `plane`
    // floats further; read: everything outside the window is zero; written: only the window is
    // bs > 0: the planar side is "block-planar" instead, tokens [.][blk][re | im][bs]: element (t, c) at t * 2 C + (c / bs) * 2 bs +
    // c % bs, im bs floats further (the operand layout of the one-GEMM-per-layer AFNO block MLP, afno_tiled._BlockComplexLinearBP)
    long long plane_in, plane_out, Jw;
    int r0, r1, bs;
    int dbg_skip;             // measurement only (DLWP_FFT_SKIP_PASSES=1): the butterfly passes are skipped, the kernels move data only
    // C2C planar store with a soft-shrink derivative folded in (dlwp_rfft2_planar_masked): the stored component is zeroed where the
    // same element of `mask_p` (the layer's saved pre-activation, the output's layout) has magnitude <= mask_lam
    const float* mask_p;
    float mask_lam;
    int plane_bf16;           // the planar side of a C2C pass (window read or written, and mask_p) is a bf16 array (dlwp_*_planar_ex)
};

__device__ __forceinline__ void load_table(float2* tabs, const FftAxis& f) {
    for (int i = threadIdx.x; i < f.N; i += blockDim.x) tabs[i] = f.tab[i];
}

// ---- W-axis real -> complex.  CF = false: channels-last, lanes = channel pairs of row (b, h) = blockIdx.y;
//      CF = true: channels-first, lanes = pairs of image rows (2 IB consecutive rows per workgroup, blockIdx.y).
template <int OUTS, int NT, bool CF, bool PRIME = false, int SP = 0>
__global__ __launch_bounds__(NT) void fft_r2c_kernel(FftIO a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const FftAxis& f = a.ax;
    const int IBP = f.IB + 1, W = f.N, Wh = W / 2;
    float2* buf = reinterpret_cast<float2*>(smem);   // [W][IBP]
    float2* tabs = buf + W * IBP;                    // [W]
    DLWP_STAMP(0);
    load_table(tabs, f);
    const long long o = blockIdx.y;
    const int c0 = CF ? 0 : blockIdx.x * 2 * f.IB;
    const long long r0 = CF ? o * 2 * f.IB : 0;
    // packed load: z[w] = x1[w] + i x2[w]
    if (!CF) {
        const int nl = min(f.IB, (a.C - c0) / 2);
        const float* src = a.in + o * W * a.C + c0;
        for (int e = threadIdx.x; e < W << f.logIB; e += NT) {
            const int lane = e & (f.IB - 1), w = e >> f.logIB;
            float2 v = make_float2(0.f, 0.f);
            if (lane < nl) v = *reinterpret_cast<const float2*>(src + (long long)w * a.C + 2 * lane);
            buf[w * IBP + lane] = v;
        }
    } else {
        for (int e = threadIdx.x; e < W * 2 * f.IB; e += NT) {      // threads run along w: coalesced rows
            const int rr = e / W, w = e - rr * W;
            const long long row = r0 + rr;
            const float v = row < a.nrows ? a.in[row * W + w] : 0.f;
            float* dst = reinterpret_cast<float*>(&buf[w * IBP + (rr >> 1)]);
            dst[rr & 1] = v;
        }
    }
    __syncthreads();
    DLWP_STAMP(1);
    if constexpr (SP != 0) { if (!a.dbg_skip) lds_fft_static<SP>(buf, tabs, -1.f); }
    else if (!a.dbg_skip) lds_fft<OUTS, NT, PRIME>(buf, tabs, f, -1.f);
    DLWP_STAMP(2);
    // split the two spectra: X1 = (Z_k + conj Z_{W-k}) / 2,  X2 = -i (Z_k - conj Z_{W-k}) / 2
    if (!CF) {
        const int nl = min(f.IB, (a.C - c0) / 2);
        float* dst = a.out + (o * a.Wc * a.C + c0) * 2;
        for (int e = threadIdx.x; e < (Wh + 1) << f.logIB; e += NT) {
            const int lane = e & (f.IB - 1), k = e >> f.logIB;
            if (lane >= nl) continue;
            const float2 zk = buf[k * IBP + lane], zm = buf[(k ? W - k : 0) * IBP + lane];
            const bool edge = k == 0 || 2 * k == W;
            const float s = 0.5f * a.scale * (edge ? 1.f : a.w_int);
            const float4 v = make_float4(s * (zk.x + zm.x), s * (zk.y - zm.y), s * (zk.y + zm.y), s * (zm.x - zk.x));
            *reinterpret_cast<float4*>(dst + ((long long)k * a.C + 2 * lane) * 2) = v;
        }
    } else {
        for (int e = threadIdx.x; e < (Wh + 1) * 2 * f.IB; e += NT) {     // threads run along k
            const int rr = e / (Wh + 1), k = e - rr * (Wh + 1);
            const long long row = r0 + rr;
            if (row >= a.nrows) continue;
            const float2 zk = buf[k * IBP + (rr >> 1)], zm = buf[(k ? W - k : 0) * IBP + (rr >> 1)];
            const bool edge = k == 0 || 2 * k == W;
            const float s = 0.5f * a.scale * (edge ? 1.f : a.w_int);
            const float2 v = (rr & 1) ? make_float2(s * (zk.y + zm.y), s * (zm.x - zk.x)) : make_float2(s * (zk.x + zm.x), s * (zk.y - zm.y));
            reinterpret_cast<float2*>(a.out)[row * a.Wc + k] = v;
        }
    }
    DLWP_STAMP(3);
}

// ---- W-axis complex (Hermitian half) -> real
template <int OUTS, int NT, bool CF, bool PRIME = false, int SP = 0>
__global__ __launch_bounds__(NT) void fft_c2r_kernel(FftIO a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const FftAxis& f = a.ax;
    const int IBP = f.IB + 1, W = f.N, Wh = W / 2;
    float2* buf = reinterpret_cast<float2*>(smem);
    float2* tabs = buf + W * IBP;
    DLWP_STAMP(8);
    load_table(tabs, f);
    const long long o = blockIdx.y;
    const int c0 = CF ? 0 : blockIdx.x * 2 * f.IB;
    const long long r0 = CF ? o * 2 * f.IB : 0;
    // the residual values this thread adds in its stores: loaded first, so that their latency lies under the transform
    float2 resv[OUTS];
    if (!CF && a.res) {
        const int nl = min(f.IB, (a.C - c0) / 2);
        const float* rsrc = a.res + o * W * a.C + c0;
#pragma unroll
        for (int u = 0; u < OUTS; ++u) {
            const int e = threadIdx.x + u * NT, lane = e & (f.IB - 1), w = e >> f.logIB;
            resv[u] = (w < W && lane < nl) ? *reinterpret_cast<const float2*>(rsrc + (long long)w * a.C + 2 * lane) : make_float2(0.f, 0.f);
        }
        if (a.res2) {
            const float* rsrc2 = a.res2 + o * W * a.C + c0;
#pragma unroll
            for (int u = 0; u < OUTS; ++u) {
                const int e = threadIdx.x + u * NT, lane = e & (f.IB - 1), w = e >> f.logIB;
                if (w < W && lane < nl) {
                    const float2 r2 = *reinterpret_cast<const float2*>(rsrc2 + (long long)w * a.C + 2 * lane);
                    resv[u].x += r2.x;
                    resv[u].y += r2.y;
                }
            }
        }
    }
    // merge: Z_k = Y1_k + i Y2_k, Z_{W-k} = conj(Y1_k) + i conj(Y2_k); DC / Nyquist use the real parts only
    if (!CF) {
        const int nl = min(f.IB, (a.C - c0) / 2);
        const float* src = a.in + (o * a.Wc * a.C + c0) * 2;
        for (int e = threadIdx.x; e < (Wh + 1) << f.logIB; e += NT) {
            const int lane = e & (f.IB - 1), k = e >> f.logIB;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane < nl) v = *reinterpret_cast<const float4*>(src + ((long long)k * a.C + 2 * lane) * 2);
            const bool edge = k == 0 || 2 * k == W;
            const float wk = edge ? 1.f : a.w_int;
            float y1x = wk * v.x, y1y = edge ? 0.f : wk * v.y, y2x = wk * v.z, y2y = edge ? 0.f : wk * v.w;
            buf[k * IBP + lane] = make_float2(y1x - y2y, y1y + y2x);
            if (!edge) buf[(W - k) * IBP + lane] = make_float2(y1x + y2y, y2x - y1y);
        }
    } else {
        // zero first (pairs are assembled by two threads each), then accumulate the two rows of every pair
        for (int e = threadIdx.x; e < W * IBP; e += NT) buf[e] = make_float2(0.f, 0.f);
        __syncthreads();
        for (int e = threadIdx.x; e < (Wh + 1) * f.IB; e += NT) {          // one thread builds a whole pair: no races
            const int lane = e / (Wh + 1), k = e - lane * (Wh + 1);
            const long long ra = r0 + 2 * lane, rb = ra + 1;
            const float2 y1 = ra < a.nrows ? reinterpret_cast<const float2*>(a.in)[ra * a.Wc + k] : make_float2(0.f, 0.f);
            const float2 y2 = rb < a.nrows ? reinterpret_cast<const float2*>(a.in)[rb * a.Wc + k] : make_float2(0.f, 0.f);
            const bool edge = k == 0 || 2 * k == W;
            const float wk = edge ? 1.f : a.w_int;
            const float y1x = wk * y1.x, y1y = edge ? 0.f : wk * y1.y, y2x = wk * y2.x, y2y = edge ? 0.f : wk * y2.y;
            buf[k * IBP + lane] = make_float2(y1x - y2y, y1y + y2x);
            if (!edge) buf[(W - k) * IBP + lane] = make_float2(y1x + y2y, y2x - y1y);
        }
    }
    __syncthreads();
    DLWP_STAMP(9);
    if constexpr (SP != 0) { if (!a.dbg_skip) lds_fft_static<SP>(buf, tabs, +1.f); }
    else if (!a.dbg_skip) lds_fft<OUTS, NT, PRIME>(buf, tabs, f, +1.f);
    DLWP_STAMP(10);
    if (!CF) {
        const int nl = min(f.IB, (a.C - c0) / 2);
        float* dst = a.out + o * W * a.C + c0;
#pragma unroll
        for (int u = 0; u < OUTS; ++u) {            // OUTS * NT >= W * IB (shape_of)
            const int e = threadIdx.x + u * NT, lane = e & (f.IB - 1), w = e >> f.logIB;
            if (w >= W || lane >= nl) continue;
            const float2 z = buf[w * IBP + lane];
            float2 v = make_float2(a.scale * z.x, a.scale * z.y);
            if (a.res) { v.x += resv[u].x; v.y += resv[u].y; }
            *reinterpret_cast<float2*>(dst + (long long)w * a.C + 2 * lane) = v;
        }
    } else {
        for (int e = threadIdx.x; e < W * 2 * f.IB; e += NT) {
            const int rr = e / W, w = e - rr * W;
            const long long row = r0 + rr;
            if (row >= a.nrows) continue;
            const float2 z = buf[w * IBP + (rr >> 1)];
            a.out[row * W + w] = a.scale * ((rr & 1) ? z.y : z.x);
        }
    }
    DLWP_STAMP(11);
}

// ---- complex -> complex along an axis of stride J (the H axis of both layouts): element (o, n, j) at ((o * N + n) * J + j)
template <int OUTS, int NT, bool PRIME = false, int SP = 0>
__global__ __launch_bounds__(NT) void fft_c2c_kernel(FftIO a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const FftAxis& f = a.ax;
    const int IBP = f.IB + 1, N = f.N;
    float2* buf = reinterpret_cast<float2*>(smem);
    float2* tabs = buf + N * IBP;
    DLWP_STAMP(4);
    load_table(tabs, f);
    const long long o = blockIdx.y, j0 = (long long)blockIdx.x * f.IB;
    const int nl = (int)min((long long)f.IB, a.J - j0);
    const long long base = o * N * a.J + j0;
    const int R = a.r1 - a.r0;
    float2* dst = reinterpret_cast<float2*>(a.out) + base;
    if (a.plane_in && j0 >= a.Jw) {                   // a column block outside the window: its transform is zero
        for (int e = threadIdx.x; e < N << f.logIB; e += NT) {
            const int lane = e & (f.IB - 1), n = e >> f.logIB;
            if (lane < nl) dst[(long long)n * a.J + lane] = make_float2(0.f, 0.f);
        }
        return;
    }
    // window side: element of row n and this thread's column j = j0 + lane (NT is a multiple of IB: the lane of a thread is fixed)
    const int mylane = threadIdx.x & (f.IB - 1);
    const long long jj = j0 + mylane;
    long long wcol = jj, wim = a.plane_in ? a.plane_in : a.plane_out;
    int wmul = 1;
    if (a.bs) {
        const int c = (int)(jj % a.C);
        wcol = (jj - c) * 2 + (c / a.bs) * 2 * a.bs + c % a.bs;
        wim = a.bs;
        wmul = 2;
    }
    const long long wrow0 = o * R * a.Jw * wmul;
    const float2* src = reinterpret_cast<const float2*>(a.in) + base;
    for (int e = threadIdx.x; e < N << f.logIB; e += NT) {
        const int lane = e & (f.IB - 1), n = e >> f.logIB;
        float2 z = make_float2(0.f, 0.f);
        if (lane < nl) {
            if (a.plane_in) {
                if (n >= a.r0 && n < a.r1 && jj < a.Jw) {
                    const long long i = wrow0 + (long long)(n - a.r0) * a.Jw * wmul + wcol;
                    if (a.plane_bf16) {
                        const __bf16* hin = reinterpret_cast<const __bf16*>(a.in);
                        z = make_float2((float)hin[i], (float)hin[wim + i]);
                    } else {
                        z = make_float2(a.in[i], a.in[wim + i]);
                    }
                }
            } else {
                z = src[(long long)n * a.J + lane];
            }
        }
        buf[n * IBP + lane] = z;
    }
    __syncthreads();
    DLWP_STAMP(5);
    if constexpr (SP == -1) { if (!a.dbg_skip) pass_prime_sym<NT>(buf, f, a.sg); }      // the axis is one folded prime pass: a lean kernel
    else if constexpr (SP != 0) { if (!a.dbg_skip) lds_fft_static<SP>(buf, tabs, a.sg); }
    else if (!a.dbg_skip) lds_fft<OUTS, NT, PRIME>(buf, tabs, f, a.sg);
    DLWP_STAMP(6);
    for (int e = threadIdx.x; e < N << f.logIB; e += NT) {
        const int lane = e & (f.IB - 1), n = e >> f.logIB;
        if (lane < nl) {
            const float2 z = buf[n * IBP + lane];
            if (a.plane_out) {
                if (n >= a.r0 && n < a.r1 && jj < a.Jw) {
                    const long long i = wrow0 + (long long)(n - a.r0) * a.Jw * wmul + wcol;
                    float ox = a.scale * z.x, oy = a.scale * z.y;
                    if (a.plane_bf16) {
                        if (a.mask_p) {
                            const __bf16* hm = reinterpret_cast<const __bf16*>(a.mask_p);
                            if (!(fabsf((float)hm[i]) > a.mask_lam)) ox = 0.f;
                            if (!(fabsf((float)hm[wim + i]) > a.mask_lam)) oy = 0.f;
                        }
                        __bf16* ho = reinterpret_cast<__bf16*>(a.out);
                        ho[i] = (__bf16)ox;
                        ho[wim + i] = (__bf16)oy;
                    } else {
                        if (a.mask_p) {
                            if (!(fabsf(a.mask_p[i]) > a.mask_lam)) ox = 0.f;
                            if (!(fabsf(a.mask_p[wim + i]) > a.mask_lam)) oy = 0.f;
                        }
                        a.out[i] = ox;
                        a.out[wim + i] = oy;
                    }
                }
            } else {
                dst[(long long)n * a.J + lane] = make_float2(a.scale * z.x, a.scale * z.y);
            }
        }
    }
    DLWP_STAMP(7);
}

#ifdef DLWP_STAMPS
}  // namespace
extern "C" int dlwp_debug_stamps_fft(unsigned long long* host_out) {
    DLWP_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dlwp_stamps), sizeof(unsigned long long) * 32));
    return DLWP_OK;
}
namespace {
#endif

int factorise(int N, int* rad) {
    int n = 0;
    auto take = [&](int r) { while (N % r == 0 && n < MAXRAD) { rad[n++] = r; N /= r; } };
    take(4); take(2); take(3); take(5); take(7);
    for (int p = 11; N > 1 && n < MAXRAD; p += 2) take(p);
    return N == 1 ? n : -1;
}

}  // namespace

struct dlwp_fft_plan {
    int H, W;
    FftAxis axW, axH;         // IB chosen per axis from the LDS budget
    float2 *tabW, *tabH;
    float *amatW, *amatH;     // DFT matrices of a large last prime radix (pass_prime_mfma), or nullptr
};

namespace {

// compile-time plan of an axis, or 0: SPlan<1> = a W axis of 180 (channels-last real passes), SPlan<2> = an H axis of 90 -- the
// FourCastNet token grids 90 x 180 and 103 x 180 (DLWP_FFT_STATIC=0: the run-time plan everywhere)
int static_plan_for(int N, bool w_axis) {
    if (dlwp_tune_or("FFT_STATIC", 1) == 0) return 0;
    // defaults 3 / 4 (the larger radix first: it is the pass without twiddles): 48.3 / 51.5 us for rfft2 / irfft2 of 90 x 180 x 768
    // against 50.5 / 53.0 (plans 1 / 2), 51.6 / 56.9 (5 / 2), 49.6 / 55.1 (1 / 6); the run-time plan: 83.5 / 79.2
    if (w_axis && N == SPlan<1>::N) { const int v = dlwp_tune_or("FFT_SPW", 3); return v == 1 || v == 5 ? v : 3; }
    if (!w_axis && N == SPlan<2>::N) { const int v = dlwp_tune_or("FFT_SPH", 4); return v == 2 || v == 6 ? v : 4; }
    return 0;
}

int make_axis(FftAxis& ax, int N, float2** tab_dev, float** amat_dev, int ib_cap, bool w_axis) {
    ax.N = N;
    ax.amat = nullptr;
    ax.rp = 0;
    ax.sp = static_plan_for(N, w_axis);
    switch (ax.sp) {
        case 1: ib_cap = 1 << SPlan<1>::LOGIB; break;
        case 2: ib_cap = 1 << SPlan<2>::LOGIB; break;
        case 3: ib_cap = 1 << SPlan<3>::LOGIB; break;
        case 4: ib_cap = 1 << SPlan<4>::LOGIB; break;
        case 5: ib_cap = 1 << SPlan<5>::LOGIB; break;
        case 6: ib_cap = 1 << SPlan<6>::LOGIB; break;
        default: break;
    }
    ax.nrad = factorise(N, ax.rad);
    DLWP_REQUIRE(ax.nrad > 0, DLWP_E_UNSUPPORTED, "fft: cannot factorise %d into at most %d radices", N, MAXRAD);
    int p = 1;
    for (int s = 0; s < ax.nrad; ++s) {
        ax.dp[s] = make_fastdiv(p);
        ax.dM[s] = make_fastdiv(p * ax.rad[s]);
        p *= ax.rad[s];
    }
    // inner lanes: the widest power of two whose [N][IB + 1] tile plus the table fits ~150 KB of LDS
    int ib = ib_cap;
    // a large prime last radix (103 of 721 / 7 = 103 rows of the 721 x 1440 patch grid) runs on the matrix cores only in the
    // 512-thread shape (N * IB > 4096): widen the lanes until it does (103 x 16 lanes: generic pass, 213 us per planar
    // transform of 103 x 180 x 768; 103 x 64: 149 us)
    if (ax.rad[ax.nrad - 1] >= 16)
        while (N * ib <= 4096 && (size_t)N * (2 * ib + 2) * sizeof(float2) <= 150 * 1024) ib <<= 1;
    while (ib > 1 && (size_t)N * (ib + 2) * sizeof(float2) > 150 * 1024) ib >>= 1;

    DLWP_REQUIRE((size_t)N * (ib + 2) * sizeof(float2) <= 150 * 1024, DLWP_E_UNSUPPORTED, "fft: axis length %d does not fit LDS", N);
    ax.IB = ib;
    ax.logIB = 0;
    while ((1 << ax.logIB) < ib) ++ax.logIB;
    std::vector<float2> tab(N);
    const double PI = 3.14159265358979323846;
    for (int n = 0; n < N; ++n) tab[n] = make_float2((float)cos(2.0 * PI * n / N), (float)-sin(2.0 * PI * n / N));
    DLWP_HIP(hipMalloc(reinterpret_cast<void**>(tab_dev), N * sizeof(float2)));
    DLWP_HIP(hipMemcpy(*tab_dev, tab.data(), N * sizeof(float2), hipMemcpyHostToDevice));
    ax.tab = *tab_dev;
    // a large prime as the last radix: its DFT matrix for the matrix-core pass.  Limits of pass_prime_mfma: at most two 16-row
    // panels per wave (2 RP <= 16 panels with 512 threads: R <= 128)
    const int R = ax.rad[ax.nrad - 1];
    const int RP = (R + 7) / 8 * 8, nthreads = N * ib > 4096 ? 512 : 256;
    ax.sym = 0;
    if (ax.nrad == 1 && R >= 16 && R <= 127 && (R & 1) && ib == 64 && nthreads == 512 && dlwp_tune_or("FFT_PRIME_SYM", 1) != 0) {
        // the whole axis is one odd prime: the folded form (pass_prime_sym) with its two real (h + 1) x (h + 1) matrices
        const int h = (R - 1) / 2;
        std::vector<float> am(2 * 64 * 64, 0.f);
        for (int q = 0; q <= h; ++q)
            for (int r = 0; r <= h; ++r) {
                const double th = 2.0 * PI * (double)(((long long)q * r) % R) / R;
                am[q * 64 + r] = (float)cos(th);
                am[64 * 64 + q * 64 + r] = (float)sin(th);
            }
        DLWP_HIP(hipMalloc(reinterpret_cast<void**>(amat_dev), am.size() * sizeof(float)));
        DLWP_HIP(hipMemcpy(*amat_dev, am.data(), am.size() * sizeof(float), hipMemcpyHostToDevice));
        ax.amat = *amat_dev;
        ax.rp = RP;
        ax.sym = 1;
    } else if (R >= 16 && nthreads == 512 && 2 * RP / 16 <= 2 * (nthreads / 64)) {
        const int K2 = 2 * RP;
        std::vector<float> am(2 * (size_t)K2 * K2, 0.f);
        for (int dir = 0; dir < 2; ++dir) {
            const double sgd = dir ? 1.0 : -1.0;
            for (int m = 0; m < K2; ++m)
                for (int kk = 0; kk < K2; ++kk) {
                    const int po = m / RP, q = m % RP, pi = kk / RP, r = kk % RP;
                    if (q >= R || r >= R) continue;
                    const double th = 2.0 * PI * (double)(((long long)q * r) % R) / R;
                    const double Fr = cos(th), Fi = sgd * sin(th);
                    am[((size_t)dir * K2 + m) * K2 + kk] = (float)(po == 0 ? (pi == 0 ? Fr : -Fi) : (pi == 0 ? Fi : Fr));
                }
        }
        DLWP_HIP(hipMalloc(reinterpret_cast<void**>(amat_dev), am.size() * sizeof(float)));
        DLWP_HIP(hipMemcpy(*amat_dev, am.data(), am.size() * sizeof(float), hipMemcpyHostToDevice));
        ax.amat = *amat_dev;
        ax.rp = RP;
    }
    return DLWP_OK;
}

size_t axis_lds(const FftAxis& ax) { return (size_t)ax.N * (ax.IB + 2) * sizeof(float2); }

// thread count and outputs-per-thread bucket of an axis
struct LaunchShape { int nt, outs; };
LaunchShape shape_of(const FftAxis& ax) {
    const int total = ax.N * ax.IB;
    const int nt = total > dlwp_tune_or("FFT_NT512_FROM", 4096) ? 512 : 256;      // (knob: 512-thread workgroups for smaller tiles too)
    const int need = ceil_div(total, nt);
    const int buckets[] = {2, 4, 8, 16, 24, 32};
    for (int b : buckets)
        if (need <= b) return {nt, b};
    return {nt, -1};
}

#define FFT_DISPATCH(KERNEL_T, EXTRA, sh, grid, lds, stream, io)                                                       \
    do {                                                                                                               \
        int rc__ = DLWP_OK;                                                                                            \
        auto go = [&](auto kern, int nt) {                                                                             \
            if ((rc__ = dlwp_ensure_lds(reinterpret_cast<const void*>(kern), lds, "fft"))) return;                     \
            hipLaunchKernelGGL(kern, grid, dim3(nt), lds, stream, io);                                                 \
        };                                                                                                             \
        if (sh.nt == 256) {                                                                                            \
            switch (sh.outs) {                                                                                         \
                case 2: go(KERNEL_T<2, 256 EXTRA>, 256); break;                                                        \
                case 4: go(KERNEL_T<4, 256 EXTRA>, 256); break;                                                        \
                case 8: go(KERNEL_T<8, 256 EXTRA>, 256); break;                                                        \
                default: go(KERNEL_T<16, 256 EXTRA>, 256); break;                                                      \
            }                                                                                                          \
        } else if (io.ax.amat) {       /* large last prime radix on the matrix cores (pass_prime_mfma) */                \
            switch (sh.outs) {                                                                                         \
                case 16: go(KERNEL_T<16, 512 EXTRA, true>, 512); break;                                                \
                case 24: go(KERNEL_T<24, 512 EXTRA, true>, 512); break;                                                \
                default: go(KERNEL_T<32, 512 EXTRA, true>, 512); break;                                                \
            }                                                                                                          \
        } else {                                                                                                       \
            switch (sh.outs) {                                                                                         \
                case 2: go(KERNEL_T<2, 512 EXTRA>, 512); break;                                                        \
                case 4: go(KERNEL_T<4, 512 EXTRA>, 512); break;                                                        \
                case 8: go(KERNEL_T<8, 512 EXTRA>, 512); break;                                                        \
                case 16: go(KERNEL_T<16, 512 EXTRA>, 512); break;                                                      \
                case 24: go(KERNEL_T<24, 512 EXTRA>, 512); break;                                                      \
                default: go(KERNEL_T<32, 512 EXTRA>, 512); break;                                                      \
            }                                                                                                          \
        }                                                                                                              \
        if (rc__) return rc__;                                                                                         \
    } while (0)
#define COMMA_TRUE , true
#define COMMA_FALSE , false
#define NOTHING

int run_w_real(const dlwp_fft_plan* p, bool to_complex, bool cf, const float* in, float* out, int B, int C, float scale,
               float w_int, hipStream_t stream, const float* res = nullptr, const float* res2 = nullptr) {
    FftIO io{};
    io.res2 = res2;
    io.dbg_skip = dlwp_tune_on("FFT_SKIP_PASSES") ? 1 : 0;
    io.ax = p->axW; io.in = in; io.out = out; io.res = res; io.C = C; io.H = p->H; io.W = p->W; io.Wc = p->W / 2 + 1;
    io.scale = scale; io.w_int = w_int;
    dim3 grid;
    if (cf) {
        io.nrows = (long long)B * C * p->H;
        const long long nblk = (io.nrows + 2 * io.ax.IB - 1) / (2 * io.ax.IB);
        DLWP_REQUIRE(nblk <= 65535LL * 32768, DLWP_E_UNSUPPORTED, "fft: too many rows");
        DLWP_REQUIRE(nblk <= 65535, DLWP_E_UNSUPPORTED, "fft (channels-first): more than 65535 row blocks per call (%lld)", nblk);
        grid = dim3(1, (unsigned)nblk);
    } else {
        DLWP_REQUIRE(C % 2 == 0, DLWP_E_UNSUPPORTED, "fft (channels-last): the channel count must be even (got %d)", C);
        DLWP_REQUIRE((long long)B * p->H <= 65535, DLWP_E_UNSUPPORTED, "fft (channels-last): B * H > 65535 rows per call");
        grid = dim3(ceil_div(C, 2 * io.ax.IB), B * p->H);
    }
    const LaunchShape sh = shape_of(io.ax);
    DLWP_REQUIRE(sh.outs > 0, DLWP_E_UNSUPPORTED, "fft: axis of length %d needs more than 32 outputs per thread", io.ax.N);
    const size_t lds = axis_lds(io.ax);
    // live accounting: B*C*H real rows of W floats on one side, W/2+1 complex bins on the other (+ the optional residual rows)
    const double prows = (double)B * C * p->H;
    dlwp_prof_scope prof(stream, prows * 2.5 * p->W * log2((double)p->W), prows * (4.0 * p->W * (1 + (res ? 1 : 0) + (res2 ? 1 : 0)) + 8.0 * io.Wc),
                         to_complex ? "fft_r2c_kernel" : "fft_c2r_kernel");
    if (io.ax.sp && !cf) {                 // channels-last W axis of 180 on a compile-time plan (OUTS covers N * IB / NT = 11.25)
        int rc = DLWP_OK;
        auto go = [&](auto r2c, auto c2r, int nt) {
            if (to_complex) {
                if ((rc = dlwp_ensure_lds(reinterpret_cast<const void*>(r2c), lds, "fft"))) return;
                hipLaunchKernelGGL(r2c, grid, dim3(nt), lds, stream, io);
            } else {
                if ((rc = dlwp_ensure_lds(reinterpret_cast<const void*>(c2r), lds, "fft"))) return;
                hipLaunchKernelGGL(c2r, grid, dim3(nt), lds, stream, io);
            }
        };
        static_assert(16 * SPlan<1>::NT >= SPlan<1>::N << SPlan<1>::LOGIB && 16 * SPlan<5>::NT >= SPlan<5>::N << SPlan<5>::LOGIB,
                      "the c2r store loop walks OUTS * NT >= N * IB elements");
        switch (io.ax.sp) {
            case 3: go(fft_r2c_kernel<16, SPlan<3>::NT, false, false, 3>, fft_c2r_kernel<16, SPlan<3>::NT, false, false, 3>, SPlan<3>::NT); break;
            case 5: go(fft_r2c_kernel<16, SPlan<5>::NT, false, false, 5>, fft_c2r_kernel<16, SPlan<5>::NT, false, false, 5>, SPlan<5>::NT); break;
            default: go(fft_r2c_kernel<16, SPlan<1>::NT, false, false, 1>, fft_c2r_kernel<16, SPlan<1>::NT, false, false, 1>, SPlan<1>::NT); break;
        }
        if (rc) return rc;
        DLWP_LAUNCH_CHECK();
        return DLWP_OK;
    }
    if (to_complex) {
        if (cf) FFT_DISPATCH(fft_r2c_kernel, COMMA_TRUE, sh, grid, lds, stream, io);
        else FFT_DISPATCH(fft_r2c_kernel, COMMA_FALSE, sh, grid, lds, stream, io);
    } else {
        if (cf) FFT_DISPATCH(fft_c2r_kernel, COMMA_TRUE, sh, grid, lds, stream, io);
        else FFT_DISPATCH(fft_c2r_kernel, COMMA_FALSE, sh, grid, lds, stream, io);
    }
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

int run_h_c2c(const dlwp_fft_plan* p, const float* in, float* out, long long outer, long long J, float sg, float scale,
              hipStream_t stream, long long plane_in = 0, long long plane_out = 0, int r0 = 0, int r1 = 0, long long Jw = 0,
              int C = 0, int bs = 0, const float* mask_p = nullptr, float mask_lam = 0.f, int plane_bf16 = 0) {
    FftIO io{};
    io.mask_p = mask_p; io.mask_lam = mask_lam; io.plane_bf16 = plane_bf16;
    io.dbg_skip = dlwp_tune_on("FFT_SKIP_PASSES") ? 1 : 0;
    io.ax = p->axH; io.in = in; io.out = out; io.J = J; io.sg = sg; io.scale = scale;
    io.plane_in = plane_in; io.plane_out = plane_out; io.r0 = r0; io.r1 = r1; io.Jw = Jw; io.C = C; io.bs = bs;
    if (plane_out) J = Jw;            // only the window's columns are transformed
    DLWP_REQUIRE(outer <= 65535, DLWP_E_UNSUPPORTED, "fft: more than 65535 outer slices per call (%lld)", outer);
    const dim3 grid((unsigned)((J + io.ax.IB - 1) / io.ax.IB), (unsigned)outer);
    const LaunchShape sh = shape_of(io.ax);
    DLWP_REQUIRE(sh.outs > 0, DLWP_E_UNSUPPORTED, "fft: axis of length %d needs more than 32 outputs per thread", io.ax.N);
    const size_t lds = axis_lds(io.ax);
    // live accounting: outer * J complex columns of N points, read and written once (a bf16 plane side moves half the bytes)
    const double pcols = (double)outer * J;
    dlwp_prof_scope prof(stream, pcols * 5.0 * io.ax.N * log2((double)io.ax.N), pcols * io.ax.N * (8.0 + (plane_bf16 ? 4.0 : 8.0)), "fft_c2c_kernel");
    if (io.ax.sym) {                       // the axis is one odd prime: the kernel that holds nothing but the folded pass (the general
                                           // 512-thread kernel with every pass inlined needs 256 VGPRs + scratch: one workgroup per CU)
        auto kern = fft_c2c_kernel<16, 512, false, -1>;
        if (int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(kern), lds, "fft")) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, io);
        DLWP_LAUNCH_CHECK();
        return DLWP_OK;
    }
    if (io.ax.sp) {                        // H axis of 90 on a compile-time plan
        int rc = DLWP_OK;
        auto go = [&](auto kern, int nt) {
            if ((rc = dlwp_ensure_lds(reinterpret_cast<const void*>(kern), lds, "fft"))) return;
            hipLaunchKernelGGL(kern, grid, dim3(nt), lds, stream, io);
        };
        switch (io.ax.sp) {
            case 4: go(fft_c2c_kernel<16, SPlan<4>::NT, false, 4>, SPlan<4>::NT); break;
            case 6: go(fft_c2c_kernel<16, SPlan<6>::NT, false, 6>, SPlan<6>::NT); break;
            default: go(fft_c2c_kernel<16, SPlan<2>::NT, false, 2>, SPlan<2>::NT); break;
        }
        if (rc) return rc;
        DLWP_LAUNCH_CHECK();
        return DLWP_OK;
    }
    FFT_DISPATCH(fft_c2c_kernel, NOTHING, sh, grid, lds, stream, io);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

void norm_scales(int norm, int H, int W, float& sW_f, float& sH_f, float& sW_i, float& sH_i) {
    // forward / inverse scale per axis for torch's norm = "backward" (0), "ortho" (1), "forward" (2)
    if (norm == 1) { sW_f = sW_i = 1.f / sqrtf((float)W); sH_f = sH_i = 1.f / sqrtf((float)H); }
    else if (norm == 2) { sW_f = 1.f / W; sH_f = 1.f / H; sW_i = sH_i = 1.f; }
    else { sW_f = sH_f = 1.f; sW_i = 1.f / W; sH_i = 1.f / H; }
}

}  // namespace

extern "C" int dlwp_fft_plan_create(int H, int W, dlwp_fft_plan** out) {
    DLWP_REQUIRE(out && H > 0 && W > 1, DLWP_E_INVALID, "fft_plan_create: bad shape");
    dlwp_fft_plan* p = new dlwp_fft_plan();
    p->H = H; p->W = W; p->tabW = p->tabH = nullptr; p->amatW = p->amatH = nullptr;
    int rc;
    // inner lanes per workgroup: W axis 8 channel PAIRS = 64 bytes of a channels-last row per w, H axis 16 complex = 128 bytes
    if ((rc = make_axis(p->axW, W, &p->tabW, &p->amatW, dlwp_tune_or("FFT_IBW", 8), true)) ||
        (rc = make_axis(p->axH, H, &p->tabH, &p->amatH, dlwp_tune_or("FFT_IBH", 16), false))) {
        dlwp_fft_plan_destroy(p);
        return rc;
    }
    *out = p;
    return DLWP_OK;
}

extern "C" void dlwp_fft_plan_destroy(dlwp_fft_plan* p) {
    if (!p) return;
    if (p->tabW) (void)hipFree(p->tabW);
    if (p->tabH) (void)hipFree(p->tabH);
    if (p->amatW) (void)hipFree(p->amatW);
    if (p->amatH) (void)hipFree(p->amatH);
    delete p;
}

// layout 0: channels-last x [B][H][W][C] <-> X [B][H][W/2+1][C][2]; layout 1: channels-first x [B][C][H][W] <-> X [B][C][H][W/2+1][2]
extern "C" int dlwp_rfft2(const dlwp_fft_plan* p, const float* x, float* X, int B, int C, int layout, int norm, int adjoint,
                          void* stream_) {
    DLWP_REQUIRE(p && x && X && B > 0 && C > 0 && (layout == 0 || layout == 1) && norm >= 0 && norm <= 2, DLWP_E_INVALID,
                 "rfft2: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    float sWf, sHf, sWi, sHi;
    norm_scales(norm, p->H, p->W, sWf, sHf, sWi, sHi);
    const int Wc = p->W / 2 + 1;
    // plain: X = s_f R x.  adjoint: gX = (irfft2)^H gx = s_i R_2 gx (interior bins weighted twice)
    int rc = run_w_real(p, true, layout == 1, x, X, B, C, adjoint ? sWi : sWf, adjoint ? 2.f : 1.f, stream);
    if (rc) return rc;
    const long long outer = layout == 0 ? B : (long long)B * C, J = layout == 0 ? (long long)Wc * C : Wc;
    return run_h_c2c(p, X, X, outer, J, -1.f, adjoint ? sHi : sHf, stream);
}

// work: scratch of X's size (the H-axis pass must not overwrite the caller's spectrum)
extern "C" int dlwp_irfft2(const dlwp_fft_plan* p, const float* X, float* x, float* work, int B, int C, int layout, int norm,
                           int adjoint, void* stream_) {
    DLWP_REQUIRE(p && x && X && work && B > 0 && C > 0 && (layout == 0 || layout == 1) && norm >= 0 && norm <= 2, DLWP_E_INVALID,
                 "irfft2: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    float sWf, sHf, sWi, sHi;
    norm_scales(norm, p->H, p->W, sWf, sHf, sWi, sHi);
    const int Wc = p->W / 2 + 1;
    const long long outer = layout == 0 ? B : (long long)B * C, J = layout == 0 ? (long long)Wc * C : Wc;
    // plain: x = s_i Q X.  adjoint: gx = (rfft2)^H gX = s_f Q_{1/2} gX (interior bins weighted by one half)
    int rc = run_h_c2c(p, X, work, outer, J, +1.f, adjoint ? sHf : sHi, stream);
    if (rc) return rc;
    return run_w_real(p, false, layout == 1, work, x, B, C, adjoint ? sWf : sWi, adjoint ? 0.5f : 1.f, stream);
}

// Channels-last transforms with a WINDOW of the half spectrum as two planes, X [2 (re | im)][B][r1 - r0][c1][C]: rows r0 <= kh < r1
// and columns kw < c1 of rfft2's [B][H][W/2+1][C] -- the operand of the AFNO mixer's block-diagonal GEMMs on the kept modes
// (afno_tiled._BlockComplexLinear), so that no torch copy or zero fill stands between the transforms and the products
// (reference: torch.fft.rfft2, the slice x[:, total_modes-kept_modes:total_modes+kept_modes, :kept_modes] and irfft2 of the
// zero-initialised o2, src/dlwpbench/models/fourcastnet/fourcastnet.py:85-124).  The H pass of the forward transform only runs
// over the kept columns; the inverse reads zeros outside the window.  adjoint: as dlwp_rfft2 / dlwp_irfft2 (the adjoint of the
// windowed inverse is the windowed forward transform and vice versa).  work: scratch of the FULL half spectrum's size.
// bs > 0: X is block-planar instead, [B][r1 - r0][c1][C / bs][2 (re | im)][bs] (FftIO::bs).  irfft2_planar's residual (field-shaped, or
// null) is added to its output: the skip connection around AFNO2D's transform pair, and in the backward pass the gradient that
// reached the input along that skip.
extern "C" int dlwp_rfft2_planar(const dlwp_fft_plan* p, const float* x, float* X, float* work, int B, int C, int r0, int r1, int c1,
                                 int bs, int norm, int adjoint, void* stream_) {
    return dlwp_rfft2_planar_masked(p, x, X, work, nullptr, 0.f, B, C, r0, r1, c1, bs, norm, adjoint, stream_);
}

extern "C" int dlwp_rfft2_planar_masked(const dlwp_fft_plan* p, const float* x, float* X, float* work, const float* mask, float lam, int B,
                                        int C, int r0, int r1, int c1, int bs, int norm, int adjoint, void* stream_) {
    return dlwp_rfft2_planar_ex(p, x, X, work, mask, lam, B, C, r0, r1, c1, bs, norm, adjoint, 0, stream_);
}

extern "C" int dlwp_rfft2_planar_ex(const dlwp_fft_plan* p, const float* x, void* X_, float* work, const void* mask_, float lam, int B,
                                    int C, int r0, int r1, int c1, int bs, int norm, int adjoint, int flags, void* stream_) {
    float* X = static_cast<float*>(X_);
    const float* mask = static_cast<const float*>(mask_);
    DLWP_REQUIRE(flags == 0 || flags == 1, DLWP_E_INVALID, "rfft2_planar_ex: flags is 0 or 1 (the window X and the mask are bf16 arrays)");
    DLWP_REQUIRE(p && x && X && work && B > 0 && C > 0 && norm >= 0 && norm <= 2, DLWP_E_INVALID, "rfft2_planar: bad argument");
    const int Wc = p->W / 2 + 1;
    DLWP_REQUIRE(0 <= r0 && r0 < r1 && r1 <= p->H && c1 >= 1 && c1 <= Wc, DLWP_E_INVALID, "rfft2_planar: window [%d, %d) x %d outside %d x %d",
                 r0, r1, c1, p->H, Wc);
    DLWP_REQUIRE(bs >= 0 && (bs == 0 || C % bs == 0), DLWP_E_INVALID, "planar fft: channel block %d does not divide C = %d", bs, C);
    hipStream_t stream = (hipStream_t)stream_;
    float sWf, sHf, sWi, sHi;
    norm_scales(norm, p->H, p->W, sWf, sHf, sWi, sHi);
    int rc = run_w_real(p, true, false, x, work, B, C, adjoint ? sWi : sWf, adjoint ? 2.f : 1.f, stream);
    if (rc) return rc;
    const long long J = (long long)Wc * C, Jw = (long long)c1 * C;
    return run_h_c2c(p, work, X, B, J, -1.f, adjoint ? sHi : sHf, stream, 0, (long long)B * (r1 - r0) * Jw, r0, r1, Jw, C, bs, mask, lam, flags);
}

extern "C" int dlwp_irfft2_planar(const dlwp_fft_plan* p, const float* X, float* x, float* work, const float* residual, int B, int C,
                                  int r0, int r1, int c1, int bs, int norm, int adjoint, void* stream_) {
    return dlwp_irfft2_planar2(p, X, x, work, residual, nullptr, B, C, r0, r1, c1, bs, norm, adjoint, stream_);
}

extern "C" int dlwp_irfft2_planar2(const dlwp_fft_plan* p, const float* X, float* x, float* work, const float* residual,
                                   const float* residual2, int B, int C, int r0, int r1, int c1, int bs, int norm, int adjoint,
                                   void* stream_) {
    return dlwp_irfft2_planar_ex(p, X, x, work, residual, residual2, B, C, r0, r1, c1, bs, norm, adjoint, 0, stream_);
}

extern "C" int dlwp_irfft2_planar_ex(const dlwp_fft_plan* p, const void* X_, float* x, float* work, const float* residual,
                                     const float* residual2, int B, int C, int r0, int r1, int c1, int bs, int norm, int adjoint,
                                     int flags, void* stream_) {
    const float* X = static_cast<const float*>(X_);
    DLWP_REQUIRE(flags == 0 || flags == 1, DLWP_E_INVALID, "irfft2_planar_ex: flags is 0 or 1 (the window X is a bf16 array)");
    DLWP_REQUIRE(p && x && X && work && B > 0 && C > 0 && norm >= 0 && norm <= 2, DLWP_E_INVALID, "irfft2_planar: bad argument");
    DLWP_REQUIRE(!residual2 || residual, DLWP_E_INVALID, "irfft2_planar2: a second residual needs the first");
    const int Wc = p->W / 2 + 1;
    DLWP_REQUIRE(0 <= r0 && r0 < r1 && r1 <= p->H && c1 >= 1 && c1 <= Wc, DLWP_E_INVALID, "irfft2_planar: window [%d, %d) x %d outside %d x %d",
                 r0, r1, c1, p->H, Wc);
    DLWP_REQUIRE(bs >= 0 && (bs == 0 || C % bs == 0), DLWP_E_INVALID, "planar fft: channel block %d does not divide C = %d", bs, C);
    hipStream_t stream = (hipStream_t)stream_;
    float sWf, sHf, sWi, sHi;
    norm_scales(norm, p->H, p->W, sWf, sHf, sWi, sHi);
    const long long J = (long long)Wc * C, Jw = (long long)c1 * C;
    int rc = run_h_c2c(p, X, work, B, J, +1.f, adjoint ? sHf : sHi, stream, (long long)B * (r1 - r0) * Jw, 0, r0, r1, Jw, C, bs, nullptr, 0.f,
                       flags);
    if (rc) return rc;
    return run_w_real(p, false, false, work, x, B, C, adjoint ? sWf : sWi, adjoint ? 0.5f : 1.f, stream, residual, residual2);
}

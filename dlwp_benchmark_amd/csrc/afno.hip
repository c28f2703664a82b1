// AFNO2D spectral token mixer (FourCastNet), forward and backward, channels-last [B,H,W,C].
// Reference: AFNO2D.forward src/nsbench/models/fourcastnet/fourcastnet.py:77-126 (dlwpbench twin
// src/dlwpbench/models/fourcastnet/fourcastnet.py:78-127): rfft2(ortho) over (H,W) -> per-mode
// block-diagonal complex 2-layer MLP (ReLU on re/im separately) on the kept-mode window
// rows [tm-km, tm+km) x cols [0, km) with tm = H//2+1, km = int(tm*frac) (computed from H only,
// :92-93 — reproduced exactly) -> zeros elsewhere -> softshrink -> irfft2(ortho) -> + x.
//
// MI355X design (grids whose block spectrum fits LDS: 16x16 nsbench, 32x64 dlwpbench): one workgroup of 16 waves per
// (sample, channel block).  The block's half spectrum [H][c1][bs] stays in LDS for the whole chain
// row DFT -> column DFT -> mixer -> inverse column -> inverse row (+residual): x is read once, y is written once, nothing
// else touches HBM except the kept spectrum saved for the backward pass.  Only the c1 = km columns that survive the mode
// window are ever computed (pruned DFT).  Every stage is a small exact-f32 MFMA product: the DFT passes with transform
// fragments built on the fly from LDS twiddle tables (N <= 64: no stored DFT matrices), the block-diagonal complex MLP on
// real images of the weights.  The backward pass reuses the same chain on gy with the adjoint scalings.
#include "common.hip.h"
#include "dlwpmi_internal.h"
#include <cmath>

namespace {

constexpr int NT = 1024;   // only B * nb workgroups exist (16 at the nsbench shape): the chain is a sequence of thread-parallel
                           // passes, so the workgroup is as wide as the hardware allows (256 threads: 356, 1024: 470 samples/s
                           // on the nsbench AFNONet step; 16 rows per row-pass step instead of 4: 483)
constexpr int ELD = 36;     // row stride of the 32 x 32 weight images (conflict-free row and column fragment reads)
constexpr int TLD = 36;     // row stride of the mixer's [16 modes][32] wave tiles
constexpr int RS = 16;     // image rows staged per row-pass step

struct AfnoDev {
    const float* x;        // fwd: input x; bwd: upstream gradient gy
    const float* res2;     // fwd, optional: a second residual added to the output (the block's double skip)
    float* y;              // fwd: output; bwd: gx
    float2* xsave;         // [B][nb][R][c1][bs] kept spectrum of x (fwd writes, bwd reads)
    const float *w1, *b1, *w2, *b2;   // [2][nb][bs][bs], [2][nb][bs], ...
    float *gw1, *gb1, *gw2, *gb2;     // bwd (atomic accumulate)
    int B, H, W, C, nb, bs, r0, r1, c1;
    int nwm;               // waves that run the mixer (each owns 4 wave tiles in the staging region)
    float lambda;
    FastDiv dbs, dc1, dW, dH;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a*conj(b)

// ---- the four DFT passes as small MFMA products.  S is addressed as floats: Sf[((h c1 + kw) bs + ch) 2 + (re|im)].
// The transform matrices are never stored: a lane builds its A fragment (row m = r, four consecutive k) from the twiddle
// tables twW / twH in LDS.  B fragments (k = 4g+q, column ch = r) come straight from the staged rows / the spectrum;
// padded k read a clamped (finite) element that the zero A entry cancels, padded columns are computed and dropped.
__device__ __forceinline__ int modn(int x, int n, FastDiv d) { return x - fastdiv(x, d) * n; }

// rows: S[h][kw][ch] = ck(kw) sum_w src[b][h][w][blk ch] e^{-2 pi i kw w / W};  per image row a [2 c1 x W] . [W x bs] product
__device__ __forceinline__ void row_pass_fwd(const AfnoDev& a, const float* src, int b, int blk, float* Sf, float* stage,
                                             const float2* twW, float s, bool weight_ck) {
    const int tid = threadIdx.x, bs = a.bs, c1 = a.c1, W = a.W;
    const int lane = tid & 63, wv = tid >> 6, r = lane & 15, g = lane >> 4;
    const int Mt = (2 * c1 + 15) / 16, Kc = (W + 15) / 16, chc = min(r, bs - 1);
    for (int h0 = 0; h0 < a.H; h0 += RS) {
        const int nr = min(RS, a.H - h0);
        if ((bs & 3) == 0 && (a.C & 3) == 0) {
            // 16-byte loads, all of a thread's loads in flight before the first LDS store (a load -> store loop pays one
            // global latency per iteration)
            const int bs4 = bs >> 2, n4 = nr * W * bs4;
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i4 = min(tid + q * NT, n4 - 1), rw = i4 / bs4, c4 = i4 - rw * bs4;
                v[q] = *reinterpret_cast<const float4*>(&src[(((long long)b * a.H + h0) * W + rw) * a.C + blk * bs + 4 * c4]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (tid + q * NT < n4) *reinterpret_cast<float4*>(&stage[4 * (tid + q * NT)]) = v[q];
            for (int i4 = tid + 4 * NT; i4 < n4; i4 += NT) {
                const int rw = i4 / bs4, c4 = i4 - rw * bs4;
                *reinterpret_cast<float4*>(&stage[4 * i4]) =
                    *reinterpret_cast<const float4*>(&src[(((long long)b * a.H + h0) * W + rw) * a.C + blk * bs + 4 * c4]);
            }
        } else {
            for (int idx = tid; idx < nr * W * bs; idx += NT) {
                const int rw = fastdiv(idx, a.dbs), ch = idx - rw * bs;       // rw = r*W + w
                stage[idx] = src[(((long long)b * a.H + h0) * W + rw) * a.C + blk * bs + ch];
            }
        }
        __syncthreads();
        for (int u = wv; u < nr * Mt; u += NT / 64) {
            const int hr = u / Mt, mt = u - hr * Mt;
            const int m = 16 * mt + r, kw = m >> 1, ri = m & 1;            // A row of this lane
            const float ck = kw >= c1 ? 0.f : ((weight_ck && !(kw == 0 || (W % 2 == 0 && kw == W / 2))) ? 2.f * s : s);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int kc = 0; kc < Kc; ++kc) {
                f32x4 a4, b4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int w = 16 * kc + 4 * g + q, wc = min(w, W - 1);
                    const float2 t = twW[modn(kw * wc, W, a.dW)];
                    a4[q] = w < W ? ck * (ri ? -t.y : t.x) : 0.f;
                    b4[q] = stage[(hr * W + wc) * bs + chc];
                }
                acc = mfma16_chunk(a4, b4, acc);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int mo = 16 * mt + 4 * g + j, kwo = mo >> 1;
                if (kwo < c1 && r < bs) Sf[(((h0 + hr) * c1 + kwo) * bs + r) * 2 + (mo & 1)] = acc[j];
            }
        }
        __syncthreads();
    }
}

// columns, in place: S[kh][kw][ch] = sum_h S[h][kw][ch] e^{-/+ 2 pi i kh h / H};  per kept column kw a real
// [2H x 2H] . [2H x bs] product; the wave that owns the column reads all of it before writing any of it back
template <bool INVERSE>
__device__ __forceinline__ void col_pass(const AfnoDev& a, float* Sf, const float2* twH) {
    const int tid = threadIdx.x, bs = a.bs, c1 = a.c1, H = a.H;
    const int lane = tid & 63, wv = tid >> 6, r = lane & 15, g = lane >> 4;
    const int Mt = (2 * H + 15) / 16, chc = min(r, bs - 1);              // Mt <= 8 (H <= 64)
    for (int kw = wv; kw < c1; kw += NT / 64) {
        f32x4 acc[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < Mt; ++kc) {
            f32x4 b4;
            int hq[4], riq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 16 * kc + 4 * g + q;
                hq[q] = k >> 1; riq[q] = k & 1;
                b4[q] = Sf[((min(hq[q], H - 1) * c1 + kw) * bs + chc) * 2 + riq[q]];
            }
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                if (mt < Mt) {
                    const int m = 16 * mt + r, kh = m >> 1, ro = m & 1;
                    f32x4 a4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float2 t = twH[modn(min(kh, H - 1) * min(hq[q], H - 1), H, a.dH)];
                        // forward (e^{-i th}): re = a c + b s, im = b c - a s;  inverse (e^{+i th}): re = a c - b s, im = a s + b c
                        const float sn = INVERSE ? -t.y : t.y;
                        const float v = ro == 0 ? (riq[q] == 0 ? t.x : sn) : (riq[q] == 0 ? -sn : t.x);
                        a4[q] = (kh < H && hq[q] < H) ? v : 0.f;
                    }
                    acc[mt] = mfma16_chunk(a4, b4, acc[mt]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            if (mt < Mt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int mo = 16 * mt + 4 * g + j, kho = mo >> 1;
                    if (kho < H && r < bs) Sf[((kho * c1 + kw) * bs + r) * 2 + (mo & 1)] = acc[mt][j];
                }
            }
        }
    }
    __syncthreads();
}

// rows inverse: dst[b][h][w][ch] = res[b][h][w][ch] + s sum_kw c(kw) Re(S[h][kw][ch] e^{+2 pi i kw w / W});  per image row
// a [W x 2 c1] . [2 c1 x bs] product
__device__ __forceinline__ void row_pass_inv(const AfnoDev& a, float* dst, const float* res, const float* res2, int b, int blk,
                                             const float* Sf, const float2* twW, float s, bool weight_ck) {
    const int tid = threadIdx.x, bs = a.bs, c1 = a.c1, W = a.W;
    const int lane = tid & 63, wv = tid >> 6, r = lane & 15, g = lane >> 4;
    const int Mt = (W + 15) / 16, Kc = (2 * c1 + 15) / 16, chc = min(r, bs - 1);
    for (int u = wv; u < a.H * Mt; u += NT / 64) {
        const int h = u / Mt, mt = u - h * Mt;
        const int wr = 16 * mt + r, wrc = min(wr, W - 1);                   // A row of this lane: output column w
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < Kc; ++kc) {
            f32x4 a4, b4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 16 * kc + 4 * g + q, kw = k >> 1, ri = k & 1, kwc = min(kw, c1 - 1);
                const float2 t = twW[modn(kwc * wrc, W, a.dW)];
                const float ck = (weight_ck && !(kw == 0 || (W % 2 == 0 && kw == W / 2))) ? 2.f : 1.f;
                a4[q] = (kw < c1 && wr < W) ? s * ck * (ri ? -t.y : t.x) : 0.f;     // Re((x + i y)(c + i s)) = x c - y s
                b4[q] = Sf[((h * c1 + kwc) * bs + chc) * 2 + ri];
            }
            acc = mfma16_chunk(a4, b4, acc);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int wo = 16 * mt + 4 * g + j;
            if (wo < W && r < bs) {
                const long long gi = (((long long)b * a.H + h) * W + wo) * a.C + blk * bs + r;
                dst[gi] = res[gi] + acc[j] + (res2 ? res2[gi] : 0.f);
            }
        }
    }
}

__device__ __forceinline__ float softshrink(float v, float l) { return v > l ? v - l : (v < -l ? v + l : 0.f); }

template <bool BWD>
__global__ __launch_bounds__(NT) void afno2d_kernel(AfnoDev a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bs = a.bs, c1 = a.c1, H = a.H, W = a.W;
    float2* S = reinterpret_cast<float2*>(smem);                 // [H][c1][bs]
    float2* twW = S + H * c1 * bs;                               // [W]  e^{+2 pi i n / W}
    float2* twH = twW + W;                                       // [H]
    float* E1 = reinterpret_cast<float*>(twH + H);               // [32][ELD] real image of W1: rows (i, re|im), cols (o, re|im)
    float* E2 = E1 + 32 * ELD;                                   // same for W2
    float* bsm = E2 + 32 * ELD;                                  // [2][32] biases of layer 1 / 2 in (o, re|im) order
    float* stage = bsm + 64;                                     // [RS][W][bs] row staging; reused as the mixer's wave tiles
    const int tid = threadIdx.x;
    const int b = blockIdx.x / a.nb, blk = blockIdx.x % a.nb;
    const float s = rsqrtf((float)(H * W));
    DLWP_STAMP(0);

    for (int i = tid; i < W; i += NT) { float sn, cs; sincospif(2.f * i / W, &sn, &cs); twW[i] = make_float2(cs, sn); }
    for (int i = tid; i < H; i += NT) { float sn, cs; sincospif(2.f * i / H, &sn, &cs); twH[i] = make_float2(cs, sn); }
    // complex [bs x bs] weights as real [2 bs x 2 bs] images, zero-padded to 32 x 32:
    //   (xr, xi) . [[Wr, Wi], [-Wi, Wr]] = (Re(x W), Im(x W));  the transposed image is the map g -> g . conj(W)^T
    for (int e = tid; e < 32 * 32; e += NT) {
        const int kr = e >> 5, nc = e & 31, i = kr >> 1, ri = kr & 1, o = nc >> 1, ro = nc & 1;
        float v1 = 0.f, v2 = 0.f;
        if (i < bs && o < bs) {
            const int plane = ri ^ ro;                            // same parts -> real plane, mixed -> imaginary plane
            const float sg = (ri == 1 && ro == 0) ? -1.f : 1.f;
            v1 = sg * a.w1[(plane * a.nb + blk) * bs * bs + i * bs + o];
            v2 = sg * a.w2[(plane * a.nb + blk) * bs * bs + i * bs + o];
        }
        E1[kr * ELD + nc] = v1;
        E2[kr * ELD + nc] = v2;
    }
    for (int e = tid; e < 64; e += NT) {
        const int layer = e >> 5, nc = e & 31, o = nc >> 1, ro = nc & 1;
        const float* bp = layer ? a.b2 : a.b1;
        bsm[e] = o < bs ? bp[(ro * a.nb + blk) * bs + o] : 0.f;
    }
    __syncthreads();

    // forward transform of x (fwd) / adjoint of the inverse transform applied to gy (bwd)
    DLWP_STAMP(1);
    float* Sf = reinterpret_cast<float*>(S);
    row_pass_fwd(a, a.x, b, blk, Sf, stage, twW, s, BWD);
    DLWP_STAMP(2);
    col_pass<false>(a, Sf, twH);
    DLWP_STAMP(3);

    // ---- per-mode mixer on the kept window, on the matrix cores.  The kept modes are the contiguous rows
    // [r0 c1, r1 c1) of S, each 2 bs floats (re, im interleaved) = one row of a [modes x 2 bs] real matrix; a wave takes
    // tiles of 16 modes through the whole two-layer chain (and its backward) on wave-private LDS tiles: no workgroup
    // barrier inside the mixer.  D layout of a 16 x 16 accumulator tile: lane (r, g) holds rows (modes) 4g+j, column r.
    const int R = a.r1 - a.r0, nmodes = R * c1, K2 = 2 * bs, ntiles = (nmodes + 15) / 16;
    const int lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    float* Srow0 = reinterpret_cast<float*>(S + a.r0 * c1 * bs);          // [nmodes][K2]
    float* xs_base = reinterpret_cast<float*>(a.xsave + ((long long)(b * a.nb + blk) * nmodes) * bs);   // [nmodes][K2]
    float* wt = stage + w * (4 * 16 * TLD);                                 // this wave's tiles
    float* Xt = wt, *O1t = wt + 16 * TLD, *G2t = wt + 2 * 16 * TLD, *GZt = wt + 3 * 16 * TLD;
    f32x4 gw1e[2][2], gw2e[2][2];                                           // expanded weight-gradient tiles [(i,ri)][(o,ro)]
    float gb1p[2] = {0.f, 0.f}, gb2p[2] = {0.f, 0.f};                       // bias-gradient partials of column n = r + 16 nt
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) { gw1e[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; gw2e[mt][nt] = gw1e[mt][nt]; }

    // tile[16][32] . E (or E^T) -> two 16 x 16 accumulator tiles; bias: added per column (nullptr: none)
    auto tile_gemm = [&](const float* tile, const float* E, bool transposed, const float* bias, f32x4 (&acc)[2]) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float bv = bias ? bias[16 * nt + r] : 0.f;
            acc[nt] = f32x4{bv, bv, bv, bv};
        }
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(&tile[r * TLD + 16 * kc + 4 * g]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x4 b4;
                if (transposed) {
                    b4 = *reinterpret_cast<const f32x4*>(&E[(16 * nt + r) * ELD + 16 * kc + 4 * g]);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) b4[q] = E[(16 * kc + 4 * g + q) * ELD + 16 * nt + r];
                }
                acc[nt] = mfma16_chunk(a4, b4, acc[nt]);
            }
        }
    };
    auto put_tile = [&](float* tile, const f32x4 (&v)[2]) {             // accumulator layout -> [mode][32] tile
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) tile[(4 * g + j) * TLD + 16 * nt + r] = v[nt][j];
        __builtin_amdgcn_wave_barrier();
    };
    // G[(i,ri)][(o,ro)] += sum_modes A[mode][(i,ri)] B[mode][(o,ro)]
    auto outer_acc = [&](const float* At, const float* Bt, f32x4 (&G)[2][2]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 a4;
#pragma unroll
            for (int q = 0; q < 4; ++q) a4[q] = At[(4 * g + q) * TLD + 16 * mt + r];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x4 b4;
#pragma unroll
                for (int q = 0; q < 4; ++q) b4[q] = Bt[(4 * g + q) * TLD + 16 * nt + r];
                G[mt][nt] = mfma16_chunk(a4, b4, G[mt][nt]);
            }
        }
    };

    if (w < a.nwm) {
        for (int t = w; t < ntiles; t += a.nwm) {
            const int m0 = 16 * t;
            // X tile: forward from the spectrum (and saved for the backward pass), backward from the saved copy
            {
                const float* src = BWD ? xs_base : Srow0;
                __builtin_amdgcn_wave_barrier();
                float xv[8];          // all eight loads in flight first (the backward source is global memory)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int e = lane + 64 * q, mr = e >> 5, k = e & 31;
                    const bool ok = m0 + mr < nmodes && k < K2;
                    const float v = src[ok ? (long long)(m0 + mr) * K2 + k : 0];
                    xv[q] = ok ? v : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int e = lane + 64 * q, mr = e >> 5, k = e & 31;
                    if (!BWD && m0 + mr < nmodes && k < K2) xs_base[(long long)(m0 + mr) * K2 + k] = xv[q];
                    Xt[mr * TLD + k] = xv[q];
                }
                __builtin_amdgcn_wave_barrier();
            }
            f32x4 z1[2], z2[2], o1[2];
            tile_gemm(Xt, E1, false, bsm, z1);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) o1[nt][j] = fmaxf(z1[nt][j], 0.f);
            put_tile(O1t, o1);
            tile_gemm(O1t, E2, false, bsm + 32, z2);
            if (!BWD) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int mr = 4 * g + j, n = 16 * nt + r;
                        if (m0 + mr < nmodes && n < K2) Srow0[(long long)(m0 + mr) * K2 + n] = softshrink(z2[nt][j], a.lambda);
                    }
            } else {
                // softshrink backward on the incoming spectrum gradient, then back through layer 2 and the ReLU
                f32x4 g2[2], go1[2], gz[2], gx[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int mr = 4 * g + j, n = 16 * nt + r;
                        const bool ok = m0 + mr < nmodes && n < K2;
                        const float gf = ok ? Srow0[(long long)(m0 + mr) * K2 + n] : 0.f;
                        g2[nt][j] = fabsf(z2[nt][j]) > a.lambda ? gf : 0.f;
                        gb2p[nt] += g2[nt][j];
                    }
                put_tile(G2t, g2);
                tile_gemm(G2t, E2, true, nullptr, go1);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        gz[nt][j] = z1[nt][j] > 0.f ? go1[nt][j] : 0.f;
                        gb1p[nt] += gz[nt][j];
                    }
                put_tile(GZt, gz);
                tile_gemm(GZt, E1, true, nullptr, gx);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int mr = 4 * g + j, n = 16 * nt + r;
                        if (m0 + mr < nmodes && n < K2) Srow0[(long long)(m0 + mr) * K2 + n] = gx[nt][j];
                    }
                outer_acc(O1t, G2t, gw2e);       // gW2 += conj(o1)^T g2, gW1 += conj(x)^T gz  (in the real image)
                outer_acc(Xt, GZt, gw1e);
            }
        }
    }
    __syncthreads();
    DLWP_STAMP(4);
    // zero everything outside the kept row window (columns >= c1 are never formed)
    for (int o = tid; o < H * c1 * bs; o += NT) {
        const int h = o / (c1 * bs);
        if (h < a.r0 || h >= a.r1) S[o] = make_float2(0.f, 0.f);
    }
    __syncthreads();
    if (BWD) {
        if (w < a.nwm) {
            // fold the real image of the weight gradients back into complex planes and add to the parameters' gradients:
            // G[(i,ri)][(o,ro)]: gWr[i][o] = G[(i,0)][(o,0)] + G[(i,1)][(o,1)], gWi[i][o] = G[(i,0)][(o,1)] - G[(i,1)][(o,0)].
            // Lane (r, g) holds rows (i,ri) = 4g+j of column n = (o,ro) = 16 nt + r: ri = j & 1, the partner column is in lane r^1.
            auto fold = [&](const f32x4 (&G)[2][2], float* gw) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int jp = 0; jp < 2; ++jp) {
                            const float g0 = G[mt][nt][2 * jp], g1 = G[mt][nt][2 * jp + 1];      // rows (i, re), (i, im)
                            const float p0 = __shfl_xor(g0, 1), p1 = __shfl_xor(g1, 1);          // partner column o, other part
                            const int i = (16 * mt + 4 * g + 2 * jp) >> 1, n = 16 * nt + r, o = n >> 1, ro = n & 1;
                            // ro == 0: this lane has (o,re): real = g0 + partner's g1 ; ro == 1: imag = g0 - partner's g1
                            const float val = ro == 0 ? g0 + p1 : g0 - p1;
                            (void)p0;
                            if (i < bs && o < bs) atomic_add_f32(&gw[(ro * a.nb + blk) * bs * bs + i * bs + o], val);
                        }
            };
            fold(gw1e, a.gw1);
            fold(gw2e, a.gw2);
            // bias gradients: sum the four lane groups (modes) of each column, one atomic per (wave, column)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float v1 = gb1p[nt], v2 = gb2p[nt];
                v1 += __shfl_xor(v1, 16); v1 += __shfl_xor(v1, 32);
                v2 += __shfl_xor(v2, 16); v2 += __shfl_xor(v2, 32);
                const int n = 16 * nt + r, o = n >> 1, ro = n & 1;
                if (g == 0 && o < bs) {
                    atomic_add_f32(&a.gb1[(ro * a.nb + blk) * bs + o], v1);
                    atomic_add_f32(&a.gb2[(ro * a.nb + blk) * bs + o], v2);
                }
            }
        }
    }
    DLWP_STAMP(5);
    col_pass<true>(a, Sf, twH);
    DLWP_STAMP(6);
    // inverse rows + residual: fwd = irfft (interior columns doubled); bwd = adjoint of rfft (no doubling)
    row_pass_inv(a, a.y, a.x, BWD ? nullptr : a.res2, b, blk, Sf, twW, s, !BWD);
    DLWP_STAMP(7);
}

constexpr size_t AFNO_LDS_LIMIT = 156 * 1024;
// floats in front of the staging region, and the staging region itself (row staging, later nwm x 4 wave tiles of the mixer)
size_t afno_fixed_floats(int H, int W, int bs, int c1) { return (size_t)2 * H * c1 * bs + 2 * (W + H) + 2 * 32 * ELD + 64; }
int afno_mixer_waves(int H, int W, int bs, int c1, int ntiles) {
    const size_t room = AFNO_LDS_LIMIT / sizeof(float) - afno_fixed_floats(H, W, bs, c1);
    int nwm = (int)(room / (4 * 16 * TLD));
    if (nwm > NT / 64) nwm = NT / 64;
    if (nwm > ntiles) nwm = ntiles;
    return nwm;
}
size_t afno_lds_bytes(int H, int W, int bs, int c1, int nwm) {
    const size_t stage = (size_t)RS * W * bs, tiles = (size_t)nwm * 4 * 16 * TLD;
    return sizeof(float) * (afno_fixed_floats(H, W, bs, c1) + (stage > tiles ? stage : tiles));
}

int afno_setup(AfnoDev& a, int B, int H, int W, int C, int nb, float frac, const char* who) {
    DLWP_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && nb > 0 && C % nb == 0, DLWP_E_INVALID, "%s: bad shape", who);
    a.B = B; a.H = H; a.W = W; a.C = C; a.nb = nb; a.bs = C / nb;
    const int total = H / 2 + 1, kept = (int)(total * frac);     // fourcastnet.py:92-93
    a.r0 = total - kept < 0 ? 0 : total - kept;
    a.r1 = total + kept > H ? H : total + kept;
    a.c1 = kept < W / 2 + 1 ? kept : W / 2 + 1;
    DLWP_REQUIRE(kept > 0, DLWP_E_INVALID, "%s: hard_thresholding_fraction keeps no mode", who);
    DLWP_REQUIRE(a.bs <= 16, DLWP_E_UNSUPPORTED, "%s: block size %d > 16: use the tiled (batched GEMM) path", who, a.bs);
    DLWP_REQUIRE(H <= 64 && W <= 1024, DLWP_E_UNSUPPORTED, "%s: grid %dx%d: the LDS-resident AFNO kernel takes H <= 64", who, H, W);
    a.dbs = make_fastdiv(a.bs); a.dc1 = make_fastdiv(a.c1); a.dW = make_fastdiv(W); a.dH = make_fastdiv(H);
    a.nwm = afno_mixer_waves(H, W, a.bs, a.c1, ((a.r1 - a.r0) * a.c1 + 15) / 16);
    DLWP_REQUIRE(a.nwm >= 1 && afno_lds_bytes(H, W, a.bs, a.c1, a.nwm) <= AFNO_LDS_LIMIT, DLWP_E_UNSUPPORTED,
                 "%s: grid %dx%d with block size %d does not fit the LDS-resident AFNO kernel: use the tiled path", who, H, W, a.bs);
    return DLWP_OK;
}

}  // namespace

extern "C" long long dlwp_afno2d_save_elems(int B, int H, int W, int C, int nb, float frac) {
    AfnoDev a{};
    if (afno_setup(a, B, H, W, C, nb, frac, "afno2d")) return -1;
    return (long long)B * nb * (a.r1 - a.r0) * a.c1 * a.bs * 2;
}

extern "C" int dlwp_afno2d_fwd_res(const float* x, const float* residual, const float* w1, const float* b1, const float* w2,
                                   const float* b2, float* y, float* xsave, int B, int H, int W, int C, int nb,
                                   float sparsity_threshold, float hard_thresholding_fraction, void* stream) {
    DLWP_REQUIRE(x && w1 && b1 && w2 && b2 && y && xsave, DLWP_E_INVALID, "afno2d_fwd: NULL argument");
    AfnoDev a{};
    int rc = afno_setup(a, B, H, W, C, nb, hard_thresholding_fraction, "afno2d_fwd");
    if (rc) return rc;
    a.x = x; a.res2 = residual; a.y = y; a.xsave = reinterpret_cast<float2*>(xsave); a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2;
    a.lambda = sparsity_threshold;
    const size_t lds = afno_lds_bytes(H, W, a.bs, a.c1, a.nwm);
    if ((rc = dlwp_ensure_lds(reinterpret_cast<const void*>(afno2d_kernel<false>), lds, "afno2d_fwd"))) return rc;
    hipLaunchKernelGGL(afno2d_kernel<false>, dim3(B * nb), dim3(NT), lds, (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_afno2d_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                               float* y, float* xsave, int B, int H, int W, int C, int nb, float sparsity_threshold,
                               float hard_thresholding_fraction, void* stream) {
    return dlwp_afno2d_fwd_res(x, nullptr, w1, b1, w2, b2, y, xsave, B, H, W, C, nb, sparsity_threshold,
                               hard_thresholding_fraction, stream);
}

extern "C" int dlwp_afno2d_bwd(const float* gy, const float* xsave, const float* w1, const float* b1, const float* w2,
                               const float* b2, float* gx, float* gw1, float* gb1, float* gw2, float* gb2, int B, int H,
                               int W, int C, int nb, float sparsity_threshold, float hard_thresholding_fraction,
                               void* stream) {
    DLWP_REQUIRE(gy && xsave && w1 && b1 && w2 && b2 && gx && gw1 && gb1 && gw2 && gb2, DLWP_E_INVALID,
                 "afno2d_bwd: NULL argument");
    AfnoDev a{};
    int rc = afno_setup(a, B, H, W, C, nb, hard_thresholding_fraction, "afno2d_bwd");
    if (rc) return rc;
    a.x = gy; a.y = gx; a.xsave = const_cast<float2*>(reinterpret_cast<const float2*>(xsave));
    a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.gw1 = gw1; a.gb1 = gb1; a.gw2 = gw2; a.gb2 = gb2;
    a.lambda = sparsity_threshold;
    const size_t lds = afno_lds_bytes(H, W, a.bs, a.c1, a.nwm);
    if ((rc = dlwp_ensure_lds(reinterpret_cast<const void*>(afno2d_kernel<true>), lds, "afno2d_bwd"))) return rc;
    hipLaunchKernelGGL(afno2d_kernel<true>, dim3(B * nb), dim3(NT), lds, (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// General-grid AFNO2D ("tiled" path, python: afno_tiled.py): when the block spectrum does not fit LDS (e.g. the
// FourCastNet-scale 90x180 token grid with 768 channels, or 720x1440 at patch 1) the transforms run as strided-batched
// MFMA GEMMs against DFT tables and the per-mode block-diagonal complex MLP as batched GEMMs over the channel blocks.
// For that, the complex block weights w [2][nb][bs][bs] (re, im planes, AFNO2D.w1 / .w2 fourcastnet.py:70-75) are
// expanded to the four real matrices of  [Or | Oi] = [Xr | Xi] . [[Wr, Wi], [-Wi, Wr]]:
//     wq[ri][ro][blk][i][o],  ri = input plane, ro = output plane
namespace {

__global__ __launch_bounds__(256) void afno_wq_expand_kernel(const float* __restrict__ w, float* __restrict__ wq, long long n) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const float wr = w[e], wi = w[n + e];
        wq[e] = wr;              // [0][0]
        wq[n + e] = wi;          // [0][1]
        wq[2 * n + e] = -wi;     // [1][0]
        wq[3 * n + e] = wr;      // [1][1]
    }
}

__global__ __launch_bounds__(256) void afno_wq_fold_kernel(const float* __restrict__ gq, float* __restrict__ gw, long long n) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        gw[e] += gq[e] + gq[3 * n + e];
        gw[n + e] += gq[n + e] - gq[2 * n + e];
    }
}

// block-planar image (tokens [T][blk][re | im][bs]): wq [blk][ri][i][ro][o] -- one real 2 bs_in x 2 bs_out matrix per channel block,
// so the complex block MLP is ONE batched real GEMM per layer -- and the bias in the same order, bq [blk][ro][o]
__global__ __launch_bounds__(256) void afno_wq_expand_bp_kernel(const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ wq,
                                                                float* __restrict__ bq, int nb, int bsi, int bso) {
    const long long n = (long long)nb * bsi * bso;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int o = (int)(e % bso), i = (int)((e / bso) % bsi), blk = (int)(e / ((long long)bso * bsi));
        const float wr = w[e], wi = w[n + e];
        float* q = wq + (long long)blk * 4 * bsi * bso;
        q[((0 * bsi + i) * 2 + 0) * bso + o] = wr;
        q[((0 * bsi + i) * 2 + 1) * bso + o] = wi;
        q[((1 * bsi + i) * 2 + 0) * bso + o] = -wi;
        q[((1 * bsi + i) * 2 + 1) * bso + o] = wr;
        if (e < 2LL * nb * bso) {                        // bias [ro][blk][o] -> [blk][ro][o]
            const int oo = (int)(e % bso), bb = (int)((e / bso) % nb), ro = (int)(e / ((long long)bso * nb));
            bq[((long long)bb * 2 + ro) * bso + oo] = b[e];
        }
    }
}

__global__ __launch_bounds__(256) void afno_wq_fold_bp_kernel(const float* __restrict__ gq, const float* __restrict__ gbq, float* __restrict__ gw,
                                                              float* __restrict__ gb, int nb, int bsi, int bso) {
    const long long n = (long long)nb * bsi * bso;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int o = (int)(e % bso), i = (int)((e / bso) % bsi), blk = (int)(e / ((long long)bso * bsi));
        const float* q = gq + (long long)blk * 4 * bsi * bso;
        gw[e] += q[((0 * bsi + i) * 2 + 0) * bso + o] + q[((1 * bsi + i) * 2 + 1) * bso + o];
        gw[n + e] += q[((0 * bsi + i) * 2 + 1) * bso + o] - q[((1 * bsi + i) * 2 + 0) * bso + o];
        if (e < 2LL * nb * bso) {
            const int oo = (int)(e % bso), bb = (int)((e / bso) % nb), ro = (int)(e / ((long long)bso * nb));
            gb[e] += gbq[((long long)bb * 2 + ro) * bso + oo];
        }
    }
}

}  // namespace

extern "C" int dlwp_afno_wq_expand_bp(const float* w, const float* b, float* wq, float* bq, int nb, int bs_in, int bs_out, void* stream) {
    DLWP_REQUIRE(w && b && wq && bq && nb > 0 && bs_in > 0 && bs_out > 0, DLWP_E_INVALID, "afno_wq_expand_bp: bad argument");
    DLWP_REQUIRE(bs_in >= 2, DLWP_E_UNSUPPORTED, "afno_wq_expand_bp: block size 1");      // the bias rides on the first 2 nb bs_out elements
    const long long n = (long long)nb * bs_in * bs_out;
    hipLaunchKernelGGL(afno_wq_expand_bp_kernel, dim3((int)std::min<long long>((n + 255) / 256, 2048)), dim3(256), 0,
                       (hipStream_t)stream, w, b, wq, bq, nb, bs_in, bs_out);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_afno_wq_fold_bp(const float* gq, const float* gbq, float* gw, float* gb, int nb, int bs_in, int bs_out, void* stream) {
    DLWP_REQUIRE(gq && gbq && gw && gb && nb > 0 && bs_in >= 2 && bs_out > 0, DLWP_E_INVALID, "afno_wq_fold_bp: bad argument");
    const long long n = (long long)nb * bs_in * bs_out;
    hipLaunchKernelGGL(afno_wq_fold_bp_kernel, dim3((int)std::min<long long>((n + 255) / 256, 2048)), dim3(256), 0,
                       (hipStream_t)stream, gq, gbq, gw, gb, nb, bs_in, bs_out);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_afno_wq_expand(const float* w, float* wq, int nb, int bs_in, int bs_out, void* stream) {
    DLWP_REQUIRE(w && wq && nb > 0 && bs_in > 0 && bs_out > 0, DLWP_E_INVALID, "afno_wq_expand: bad argument");
    const long long n = (long long)nb * bs_in * bs_out;
    hipLaunchKernelGGL(afno_wq_expand_kernel, dim3((int)std::min<long long>((n + 255) / 256, 2048)), dim3(256), 0,
                       (hipStream_t)stream, w, wq, n);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_afno_wq_fold(const float* gq, float* gw, int nb, int bs_in, int bs_out, void* stream) {
    DLWP_REQUIRE(gq && gw && nb > 0 && bs_in > 0 && bs_out > 0, DLWP_E_INVALID, "afno_wq_fold: bad argument");
    const long long n = (long long)nb * bs_in * bs_out;
    hipLaunchKernelGGL(afno_wq_fold_kernel, dim3((int)std::min<long long>((n + 255) / 256, 2048)), dim3(256), 0,
                       (hipStream_t)stream, gq, gw, n);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

#ifdef DLWP_STAMPS
extern "C" int dlwp_debug_stamps_afno(unsigned long long* host_out) {
    DLWP_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dlwp_stamps), sizeof(unsigned long long) * 32));
    return DLWP_OK;
}
#endif

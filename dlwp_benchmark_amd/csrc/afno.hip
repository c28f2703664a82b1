// AFNO2D spectral token mixer (FourCastNet), forward and backward, channels-last [B,H,W,C].
// Reference: AFNO2D.forward src/nsbench/models/fourcastnet/fourcastnet.py:77-126 (dlwpbench twin
// src/dlwpbench/models/fourcastnet/fourcastnet.py:78-127): rfft2(ortho) over (H,W) -> per-mode
// block-diagonal complex 2-layer MLP (ReLU on re/im separately) on the kept-mode window
// rows [tm-km, tm+km) x cols [0, km) with tm = H//2+1, km = int(tm*frac) (computed from H only,
// :92-93 — reproduced exactly) -> zeros elsewhere -> softshrink -> irfft2(ortho) -> + x.
//
// MI355X design (round 1, grids whose block spectrum fits LDS: 16x16 nsbench, 32x64 dlwpbench):
// one workgroup per (sample, channel block).  The block's half spectrum [H][c1][bs] stays in LDS for
// the whole chain  row DFT -> column DFT -> mixer -> inverse column -> inverse row (+residual):
// x is read once, y is written once, nothing else touches HBM except the kept spectrum saved for the
// backward pass.  Only the c1 = km columns that survive the mode window are ever computed (pruned
// DFT); DFTs are direct O(N^2) sums with LDS twiddle tables (N <= 64).  The backward pass reuses the
// same chain on gy with the adjoint scalings.
#include "common.cuh"
#include "dlwpmi_internal.h"
#include <cmath>

namespace {

constexpr int NT = 1024;   // only B * nb workgroups exist (16 at the nsbench shape): the chain is a sequence of thread-parallel
                           // passes, so the workgroup is as wide as the hardware allows (256 threads: 356, 1024: 470 samples/s
                           // on the nsbench AFNONet step; 16 rows per row-pass step instead of 4: 483)
constexpr int MAXQ = 10;   // column-pass outputs per thread held in registers
constexpr int RS = 16;     // image rows staged per row-pass step

struct AfnoDev {
    const float* x;        // fwd: input x; bwd: upstream gradient gy
    float* y;              // fwd: output; bwd: gx
    float2* xsave;         // [B][nb][R][c1][bs] kept spectrum of x (fwd writes, bwd reads)
    const float *w1, *b1, *w2, *b2;   // [2][nb][bs][bs], [2][nb][bs], ...
    float *gw1, *gb1, *gw2, *gb2;     // bwd (atomic accumulate)
    int B, H, W, C, nb, bs, r0, r1, c1;
    float lambda;
    FastDiv dbs, dc1;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a*conj(b)

// rows: S[h][kw][ch] = scale(kw) * sum_w src[b][h][w][blk ch] * e^{-2 pi i kw w / W}
__device__ __forceinline__ void row_pass_fwd(const AfnoDev& a, const float* src, int b, int blk, float2* S, float* stage,
                                             const float2* twW, float s, bool weight_ck) {
    const int tid = threadIdx.x, bs = a.bs, c1 = a.c1, W = a.W;
    for (int h0 = 0; h0 < a.H; h0 += RS) {
        const int nr = min(RS, a.H - h0);
        for (int idx = tid; idx < nr * W * bs; idx += NT) {
            const int ch = idx % bs, rw = idx / bs;           // rw = r*W + w
            const int r = rw / W, w = rw - r * W;
            stage[idx] = src[(((long long)b * a.H + h0 + r) * W + w) * a.C + blk * bs + ch];
        }
        __syncthreads();
        for (int o = tid; o < nr * c1 * bs; o += NT) {
            const int ch = fastdiv(o, a.dbs) , t2 = ch;       // o = (r*c1 + kw)*bs + ch
            const int chn = o - t2 * bs;
            const int r = fastdiv(t2, a.dc1), kw = t2 - r * c1;
            float re = 0.f, im = 0.f;
            int ti = 0;
            const float* row = stage + (r * W) * bs + chn;
            for (int w = 0; w < W; ++w) {
                const float v = row[w * bs];
                const float2 t = twW[ti];
                re += v * t.x;
                im -= v * t.y;                                  // e^{-i theta}
                ti += kw;
                if (ti >= W) ti -= W;
            }
            const float ck = (weight_ck && !(kw == 0 || (W % 2 == 0 && kw == W / 2))) ? 2.f * s : s;
            S[((h0 + r) * c1 + kw) * bs + chn] = make_float2(re * ck, im * ck);
        }
        __syncthreads();
    }
}

// columns, in place: S[kh][kw][ch] = sum_h S[h][kw][ch] * e^{-/+ 2 pi i kh h / H}
template <bool INVERSE>
__device__ __forceinline__ void col_pass(const AfnoDev& a, float2* S, const float2* twH) {
    const int tid = threadIdx.x, H = a.H, per_row = a.c1 * a.bs, total = H * per_row;
    float2 acc[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int o = tid + q * NT;
        acc[q] = make_float2(0.f, 0.f);
        if (o < total) {
            const int kh = o / per_row, rest = o - kh * per_row;
            float re = 0.f, im = 0.f;
            int ti = 0;
            for (int h = 0; h < H; ++h) {
                const float2 v = S[h * per_row + rest];
                float2 t = twH[ti];
                if (!INVERSE) t.y = -t.y;                       // forward: e^{-i theta}
                re += v.x * t.x - v.y * t.y;
                im += v.x * t.y + v.y * t.x;
                ti += kh;
                if (ti >= H) ti -= H;
            }
            acc[q] = make_float2(re, im);
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int o = tid + q * NT;
        if (o < total) S[o] = acc[q];
    }
    __syncthreads();
}

// rows inverse: dst[b][h][w][ch] = res[b][h][w][ch] + s * sum_kw c(kw) Re(S[h][kw][ch] e^{+2 pi i kw w / W})
__device__ __forceinline__ void row_pass_inv(const AfnoDev& a, float* dst, const float* res, int b, int blk, const float2* S,
                                             const float2* twW, float s, bool weight_ck) {
    const int tid = threadIdx.x, bs = a.bs, c1 = a.c1, W = a.W;
    for (int o = tid; o < a.H * W * bs; o += NT) {
        const int ch = o % bs, hw = o / bs, h = hw / W, w = hw - h * W;
        float accv = 0.f;
        int ti = 0;
        const float2* sp = S + (h * c1) * bs + ch;
        for (int kw = 0; kw < c1; ++kw) {
            const float2 v = sp[kw * bs];
            const float2 t = twW[ti];
            const float ck = (weight_ck && !(kw == 0 || (W % 2 == 0 && kw == W / 2))) ? 2.f : 1.f;
            accv += ck * (v.x * t.x - v.y * t.y);               // Re(v e^{+i theta})
            ti += w;
            if (ti >= W) ti -= W;
        }
        const long long g = (((long long)b * a.H + h) * W + w) * a.C + blk * bs + ch;
        dst[g] = res[g] + s * accv;
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
    float* w1r = reinterpret_cast<float*>(twH + H);              // [bs][bs] each: w1 re/im, w2 re/im
    float* w1i = w1r + bs * bs;
    float* w2r = w1i + bs * bs;
    float* w2i = w2r + bs * bs;
    float* bsm = w2i + bs * bs;                                  // b1 re, b1 im, b2 re, b2 im  [4][bs]
    float* stage = bsm + 4 * bs;                                 // [RS][W][bs] row staging; reused as mixer scratch
    const int tid = threadIdx.x;
    const int b = blockIdx.x / a.nb, blk = blockIdx.x % a.nb;
    const float s = rsqrtf((float)(H * W));
    DLWP_STAMP(0);

    for (int i = tid; i < W; i += NT) { float sn, cs; sincospif(2.f * i / W, &sn, &cs); twW[i] = make_float2(cs, sn); }
    for (int i = tid; i < H; i += NT) { float sn, cs; sincospif(2.f * i / H, &sn, &cs); twH[i] = make_float2(cs, sn); }
    for (int i = tid; i < bs * bs; i += NT) {
        w1r[i] = a.w1[(0 * a.nb + blk) * bs * bs + i];
        w1i[i] = a.w1[(1 * a.nb + blk) * bs * bs + i];
        w2r[i] = a.w2[(0 * a.nb + blk) * bs * bs + i];
        w2i[i] = a.w2[(1 * a.nb + blk) * bs * bs + i];
    }
    for (int i = tid; i < bs; i += NT) {
        bsm[i] = a.b1[(0 * a.nb + blk) * bs + i];
        bsm[bs + i] = a.b1[(1 * a.nb + blk) * bs + i];
        bsm[2 * bs + i] = a.b2[(0 * a.nb + blk) * bs + i];
        bsm[3 * bs + i] = a.b2[(1 * a.nb + blk) * bs + i];
    }
    __syncthreads();

    // forward transform of x (fwd) / adjoint of the inverse transform applied to gy (bwd)
    DLWP_STAMP(1);
    row_pass_fwd(a, a.x, b, blk, S, stage, twW, s, BWD);
    DLWP_STAMP(2);
    col_pass<false>(a, S, twH);
    DLWP_STAMP(3);

    // ---- per-mode mixer on the kept window; chunks of MC = NT / bs modes, one thread per (mode, channel)
    const int R = a.r1 - a.r0, nmodes = R * c1, MC = NT / bs;
    float2* o1c = reinterpret_cast<float2*>(stage);             // [MC][bs]
    float2* g2c = o1c + MC * bs;                                 // [MC][bs]  (bwd)
    float2* z1c = g2c + MC * bs;                                 // [MC][bs]  (bwd)
    float2* xc = z1c + MC * bs;                                  // [MC][bs]  (bwd) saved spectrum chunk
    const int ml = tid / bs, ch = tid - ml * bs;                 // local mode, channel
    float2* xs_base = a.xsave + ((long long)(b * a.nb + blk) * nmodes) * bs;
    float gw1a_r = 0.f, gw1a_i = 0.f, gw2a_r = 0.f, gw2a_i = 0.f, gb1r = 0.f, gb1i = 0.f, gb2r = 0.f, gb2i = 0.f;
    // weight gradients: thread = ((i,o) pair, part); the NT / (bs*bs) parts split the modes of a chunk between them (with
    // one part, 256 of the 1024 threads walked 64 modes each: a third of the backward mixer)
    const int npart = NT / (bs * bs), wpart = tid / (bs * bs), wpair = tid - wpart * bs * bs;
    const int pi_ = wpair / bs, po_ = wpair - pi_ * bs;
    for (int m0 = 0; m0 < nmodes; m0 += MC) {
        const int m = m0 + ml;
        const bool valid = ml < MC && m < nmodes;
        const int row = a.r0 + (valid ? m / c1 : 0), col = valid ? m % c1 : 0;
        float2* sp = S + (row * c1 + col) * bs;
        if (!BWD) {
            float2 xv = make_float2(0.f, 0.f);
            if (valid) { xv = sp[ch]; xs_base[(long long)m * bs + ch] = xv; }
            __syncthreads();                                      // everyone has read its x before o1 overwrites nothing yet
            // layer 1: z1[o] = sum_i x[i] W1[i][o] + b1[o]; o1 = relu(re), relu(im)
            float2 z = make_float2(bsm[ch], bsm[bs + ch]);
            if (valid)
                for (int i = 0; i < bs; ++i) {
                    const float2 v = sp[i];
                    const float wr = w1r[i * bs + ch], wi = w1i[i * bs + ch];
                    z.x += v.x * wr - v.y * wi;
                    z.y += v.y * wr + v.x * wi;
                }
            if (ml < MC) o1c[ml * bs + ch] = make_float2(fmaxf(z.x, 0.f), fmaxf(z.y, 0.f));
            __syncthreads();
            float2 o2 = make_float2(bsm[2 * bs + ch], bsm[3 * bs + ch]);
            if (valid) {
                for (int i = 0; i < bs; ++i) {
                    const float2 v = o1c[ml * bs + i];
                    const float wr = w2r[i * bs + ch], wi = w2i[i * bs + ch];
                    o2.x += v.x * wr - v.y * wi;
                    o2.y += v.y * wr + v.x * wi;
                }
                sp[ch] = make_float2(softshrink(o2.x, a.lambda), softshrink(o2.y, a.lambda));
            }
            __syncthreads();
        } else {
            // recompute layer 1 / layer 2 pre-activations from the saved spectrum
            float2 xv = make_float2(0.f, 0.f), gF = xv;
            if (valid) { xv = xs_base[(long long)m * bs + ch]; gF = sp[ch]; }
            if (ml < MC) xc[ml * bs + ch] = xv;
            __syncthreads();
            float2 z = make_float2(bsm[ch], bsm[bs + ch]);
            if (valid)
                for (int i = 0; i < bs; ++i) {
                    const float2 v = xc[ml * bs + i];
                    const float wr = w1r[i * bs + ch], wi = w1i[i * bs + ch];
                    z.x += v.x * wr - v.y * wi;
                    z.y += v.y * wr + v.x * wi;
                }
            const float2 o1 = make_float2(fmaxf(z.x, 0.f), fmaxf(z.y, 0.f));
            if (ml < MC) o1c[ml * bs + ch] = valid ? o1 : make_float2(0.f, 0.f);
            __syncthreads();
            float2 o2 = make_float2(bsm[2 * bs + ch], bsm[3 * bs + ch]);
            if (valid)
                for (int i = 0; i < bs; ++i) {
                    const float2 v = o1c[ml * bs + i];
                    const float wr = w2r[i * bs + ch], wi = w2i[i * bs + ch];
                    o2.x += v.x * wr - v.y * wi;
                    o2.y += v.y * wr + v.x * wi;
                }
            // softshrink backward (re and im independently)
            float2 g2 = make_float2(0.f, 0.f);
            if (valid) g2 = make_float2(fabsf(o2.x) > a.lambda ? gF.x : 0.f, fabsf(o2.y) > a.lambda ? gF.y : 0.f);
            if (ml < MC) g2c[ml * bs + ch] = g2;
            gb2r += g2.x; gb2i += g2.y;
            __syncthreads();
            // gO1[i=ch] = sum_o g2[o] conj(W2[i][o]); ReLU backward on re/im independently
            float2 g1 = make_float2(0.f, 0.f);
            if (valid)
                for (int o = 0; o < bs; ++o) {
                    const float2 gv = g2c[ml * bs + o];
                    const float wr = w2r[ch * bs + o], wi = w2i[ch * bs + o];
                    g1.x += gv.x * wr + gv.y * wi;
                    g1.y += gv.y * wr - gv.x * wi;
                }
            const float2 gz = make_float2(z.x > 0.f ? g1.x : 0.f, z.y > 0.f ? g1.y : 0.f);
            if (ml < MC) z1c[ml * bs + ch] = valid ? gz : make_float2(0.f, 0.f);
            gb1r += valid ? gz.x : 0.f; gb1i += valid ? gz.y : 0.f;
            __syncthreads();
            // gX[i=ch] = sum_o gz[o] conj(W1[i][o])  -> back into the spectrum buffer
            if (valid) {
                float2 gx = make_float2(0.f, 0.f);
                for (int o = 0; o < bs; ++o) {
                    const float2 gv = z1c[ml * bs + o];
                    const float wr = w1r[ch * bs + o], wi = w1i[ch * bs + o];
                    gx.x += gv.x * wr + gv.y * wi;
                    gx.y += gv.y * wr - gv.x * wi;
                }
                sp[ch] = gx;
            }
            // weight gradients for the (i,o) pair this thread owns: gW2 += conj(o1[i]) g2[o]; gW1 += conj(x[i]) gz[o]
            if (wpart < npart) {
                const int mc = min(MC, nmodes - m0);
                for (int q = wpart; q < mc; q += npart) {
                    const float2 a1 = o1c[q * bs + pi_], g2v = g2c[q * bs + po_];
                    gw2a_r += a1.x * g2v.x + a1.y * g2v.y;
                    gw2a_i += a1.x * g2v.y - a1.y * g2v.x;
                    const float2 xq = xc[q * bs + pi_], gzv = z1c[q * bs + po_];
                    gw1a_r += xq.x * gzv.x + xq.y * gzv.y;
                    gw1a_i += xq.x * gzv.y - xq.y * gzv.x;
                }
            }
            __syncthreads();
        }
    }
    DLWP_STAMP(4);
    // zero everything outside the kept row window (columns >= c1 are never formed)
    for (int o = tid; o < H * c1 * bs; o += NT) {
        const int h = o / (c1 * bs);
        if (h < a.r0 || h >= a.r1) S[o] = make_float2(0.f, 0.f);
    }
    __syncthreads();
    if (BWD) {
        if (wpart < npart) {
            const int wofs = (blk * bs + pi_) * bs + po_;
            atomic_add_f32(&a.gw1[wofs], gw1a_r);
            atomic_add_f32(&a.gw1[a.nb * bs * bs + wofs], gw1a_i);
            atomic_add_f32(&a.gw2[wofs], gw2a_r);
            atomic_add_f32(&a.gw2[a.nb * bs * bs + wofs], gw2a_i);
        }
        // bias gradients: reduce over the threads that share a channel (stride bs) through LDS
        float* red = stage;                                       // [4][NT]
        red[tid] = gb1r; red[NT + tid] = gb1i; red[2 * NT + tid] = gb2r; red[3 * NT + tid] = gb2i;
        __syncthreads();
        if (tid < bs) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            for (int q = tid; q < NT; q += bs) { s0 += red[q]; s1 += red[NT + q]; s2 += red[2 * NT + q]; s3 += red[3 * NT + q]; }
            atomic_add_f32(&a.gb1[blk * bs + tid], s0);
            atomic_add_f32(&a.gb1[a.nb * bs + blk * bs + tid], s1);
            atomic_add_f32(&a.gb2[blk * bs + tid], s2);
            atomic_add_f32(&a.gb2[a.nb * bs + blk * bs + tid], s3);
        }
        __syncthreads();
    }
    DLWP_STAMP(5);
    col_pass<true>(a, S, twH);
    DLWP_STAMP(6);
    // inverse rows + residual: fwd = irfft (interior columns doubled); bwd = adjoint of rfft (no doubling)
    row_pass_inv(a, a.y, a.x, b, blk, S, twW, s, !BWD);
    DLWP_STAMP(7);
}

size_t afno_lds_bytes(int H, int W, int bs, int c1) {
    const size_t stage = (size_t)RS * W * bs;
    size_t scratch = (size_t)4 * (NT / bs) * bs * 2;  // mixer chunks (float2)
    if (scratch < (size_t)4 * NT) scratch = 4 * NT;
    return sizeof(float) * ((size_t)2 * H * c1 * bs + 2 * (W + H) + 4 * bs * bs + 4 * bs + (stage > scratch ? stage : scratch));
}

int afno_setup(AfnoDev& a, int B, int H, int W, int C, int nb, float frac, const char* who) {
    DLWP_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && nb > 0 && C % nb == 0, DLWP_E_INVALID, "%s: bad shape", who);
    a.B = B; a.H = H; a.W = W; a.C = C; a.nb = nb; a.bs = C / nb;
    const int total = H / 2 + 1, kept = (int)(total * frac);     // fourcastnet.py:92-93
    a.r0 = total - kept < 0 ? 0 : total - kept;
    a.r1 = total + kept > H ? H : total + kept;
    a.c1 = kept < W / 2 + 1 ? kept : W / 2 + 1;
    DLWP_REQUIRE(kept > 0, DLWP_E_INVALID, "%s: hard_thresholding_fraction keeps no mode", who);
    DLWP_REQUIRE(a.bs <= 64 && NT % a.bs == 0 && a.bs * a.bs <= NT * 16, DLWP_E_UNSUPPORTED,
                 "%s: block size %d unsupported (must divide 256, <= 64)", who, a.bs);
    DLWP_REQUIRE(a.bs * a.bs <= NT, DLWP_E_UNSUPPORTED, "%s: block size %d > 16 not supported yet", who, a.bs);
    DLWP_REQUIRE(H * a.c1 * a.bs <= MAXQ * NT, DLWP_E_UNSUPPORTED,
                 "%s: grid %dx%d with block size %d exceeds the LDS-resident AFNO kernel (tiled FFT path not built yet)",
                 who, H, W, a.bs);
    a.dbs = make_fastdiv(a.bs); a.dc1 = make_fastdiv(a.c1);
    return DLWP_OK;
}

}  // namespace

extern "C" long long dlwp_afno2d_save_elems(int B, int H, int W, int C, int nb, float frac) {
    AfnoDev a{};
    if (afno_setup(a, B, H, W, C, nb, frac, "afno2d")) return -1;
    return (long long)B * nb * (a.r1 - a.r0) * a.c1 * a.bs * 2;
}

extern "C" int dlwp_afno2d_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                               float* y, float* xsave, int B, int H, int W, int C, int nb, float sparsity_threshold,
                               float hard_thresholding_fraction, void* stream) {
    DLWP_REQUIRE(x && w1 && b1 && w2 && b2 && y && xsave, DLWP_E_INVALID, "afno2d_fwd: NULL argument");
    AfnoDev a{};
    int rc = afno_setup(a, B, H, W, C, nb, hard_thresholding_fraction, "afno2d_fwd");
    if (rc) return rc;
    a.x = x; a.y = y; a.xsave = reinterpret_cast<float2*>(xsave); a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2;
    a.lambda = sparsity_threshold;
    const size_t lds = afno_lds_bytes(H, W, a.bs, a.c1);
    if ((rc = dlwp_ensure_lds(reinterpret_cast<const void*>(afno2d_kernel<false>), lds, "afno2d_fwd"))) return rc;
    hipLaunchKernelGGL(afno2d_kernel<false>, dim3(B * nb), dim3(NT), lds, (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
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
    const size_t lds = afno_lds_bytes(H, W, a.bs, a.c1);
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

}  // namespace

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

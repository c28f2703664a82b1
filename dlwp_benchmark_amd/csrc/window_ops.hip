// Window partition / reverse for shifted-window attention as single gather kernels.
// Reference (each of these is a separate full-tensor copy there): SwinTransformerBlock.forward
// src/nsbench/models/swintransformer/swin_transformer.py:213-250 (F.pad -> torch.roll -> window_partition ... window_reverse ->
// torch.roll -> crop), EarthSpecificBlock.forward src/dlwpbench/models/panguweather/panguweather.py:283-317 (ZeroPad3d ->
// roll -> window_partition ... window_reverse -> roll -> crop3d).
//
// tokens x [B][D0][D1][D2][C]  <->  windows [B * nW][N = w0 w1 w2][C]
// padded canvas P_d = f_d + D_d + back pad (a multiple of w_d); rolled[i] = padded[(i + s_d) mod P_d] (torch.roll by -s_d);
// window (i0,i1,i2) sits at index i0*sw0 + i1*sw1 + i2*sw2 inside a sample's nW windows (Swin: row-major; Pangu:
// longitude-major, panguweather utils window_partition).  Padding per axis: constant zero or circular.
//   gather : windows[...] = rolled padded x           (forward of partition; also the adjoint of reverse)
//   scatter: x[b][q] = windows at the rolled position of q + f   (forward of reverse + crop; with `dup` it sums every padded
//            copy of q, which is the adjoint of a circularly padded gather)
#include "common.hip.h"
#include "dlwpmi_internal.h"

namespace {

struct WinDev {
    const float* src;
    const float* res;      // scatter only, optional: a residual in token layout added to the result (the block's skip connection)
    const float* fill;     // gather only, optional [C]: value of the padded positions instead of zero (a Linear layer applied
                           // BEFORE the padding puts its bias there: Linear(0) = bias, see dlwp_window_gather_fill)
    float* dst;
    int B, C, D[3], P[3], f[3], s[3], w[3], nw[3], circ[3];
    long long sw[3];
    int nW, N, dup;
    // round 5 (dlwp_window_gather_ex / dlwp_window_scatter_ex): src / dst are bf16 arrays (same layouts, 2-byte elements); scale
    // [B]: the result of sample b is multiplied by scale[b] before the residual joins (stochastic depth of the branch, DropPath)
    int src_bf16, dst_bf16;
    const float* scale;
    FastDiv dC4, dN, dnW, dw12, dw2, dnw12, dnw2, dD2, dD1, dD0, dwd[3];   // umulhi divisions (index spaces < 2^31 / 256 per step)
};

__device__ __forceinline__ int pmod(int a, int m) { a %= m; return a < 0 ? a + m : a; }

typedef __bf16 wbf16x4 __attribute__((ext_vector_type(4)));
// four consecutive channels at element offset `o` of an fp32 or bf16 array
__device__ __forceinline__ f32x4 win_load4(const float* base, long long o, int bf16) {
    if (bf16) {
        const wbf16x4 h = *reinterpret_cast<const wbf16x4*>(reinterpret_cast<const __bf16*>(base) + o);
        return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    }
    return *reinterpret_cast<const f32x4*>(base + o);
}
__device__ __forceinline__ void win_store4(float* base, long long o, int bf16, const f32x4& v) {
    if (bf16) *reinterpret_cast<wbf16x4*>(reinterpret_cast<__bf16*>(base) + o) = wbf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    else *reinterpret_cast<f32x4*>(base + o) = v;
}

// one thread per (output token, 4 channels)
__global__ __launch_bounds__(256) void win_gather_kernel(WinDev a) {
    const int C4 = a.C >> 2;
    const long long total = (long long)a.B * a.nW * a.N * C4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int tok = fastdiv((int)e, a.dC4);
        const int c4 = (int)e - tok * C4;
        const int bw = fastdiv(tok, a.dN);
        const int n = tok - bw * a.N;
        const int b = fastdiv(bw, a.dnW), wlin = bw - b * a.nW;      // canonical (i0,i1,i2) row-major
        const int i0 = fastdiv(wlin, a.dnw12), r1 = wlin - i0 * a.nw[1] * a.nw[2], i1 = fastdiv(r1, a.dnw2), i2 = r1 - i1 * a.nw[2];
        const int j0 = fastdiv(n, a.dw12), r2 = n - j0 * a.w[1] * a.w[2], j1 = fastdiv(r2, a.dw2), j2 = r2 - j1 * a.w[2];
        const int ii[3] = {i0, i1, i2}, jj[3] = {j0, j1, j2};
        int q[3];
        bool ok = true;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int p = pmod(ii[d] * a.w[d] + jj[d] + a.s[d], a.P[d]);
            int v = p - a.f[d];
            if (a.circ[d]) v = pmod(v, a.D[d]);
            else ok = ok && v >= 0 && v < a.D[d];
            q[d] = v;
        }
        f32x4 val = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok) val = win_load4(a.src, ((((long long)b * a.D[0] + q[0]) * a.D[1] + q[1]) * a.D[2] + q[2]) * a.C + 4 * c4, a.src_bf16);
        else if (a.fill) val = *reinterpret_cast<const f32x4*>(a.fill + 4 * c4);
        if (a.scale) {
            const float sc = a.scale[b];
#pragma unroll
            for (int k = 0; k < 4; ++k) val[k] *= sc;
        }
        const long long wout = (long long)b * a.nW + i0 * a.sw[0] + i1 * a.sw[1] + i2 * a.sw[2];
        win_store4(a.dst, (wout * a.N + n) * a.C + 4 * c4, a.dst_bf16, val);
        (void)tok;
    }
}

// out[c] += sum over the PADDED positions of windows[.][.][c], channels c >= c_lo: the adjoint of the `fill` of
// win_gather_kernel (gradient of the bias that stands at the padded positions).  A workgroup takes 256 window tokens at a time:
// every thread classifies ONE token (the index walk of win_gather_kernel, once per token instead of once per channel quad) and
// appends the padded ones to a list in LDS; then thread = (list lane, channel quad) streams the listed rows.
__global__ __launch_bounds__(256) void win_pad_colsum_kernel(WinDev a, float* out, int c_lo) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [lanes][4 cw] partial sums | list[256] | count
    const int q_lo = c_lo >> 2, C4 = (a.C >> 2) - q_lo;              // channel quads to sum
    const int cw = C4 < 256 ? C4 : 256;                              // quads handled per thread column
    const int lanes = 256 / cw;                                      // list lanes per workgroup
    const int nq = (C4 + cw - 1) / cw;                               // quads per thread (<= 4: at most 4096 channels)
    int* list = reinterpret_cast<int*>(red + (size_t)lanes * cw * 4 * nq);
    int* count = list + 256;
    const int cq = threadIdx.x % cw, tl = threadIdx.x / cw;
    const long long ntok = (long long)a.B * a.nW * a.N;
    f32x4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long long base = (long long)blockIdx.x * 256; base < ntok; base += (long long)gridDim.x * 256) {
        if (threadIdx.x == 0) *count = 0;
        __syncthreads();
        const long long tok = base + threadIdx.x;
        if (tok < ntok) {
            const int bw = fastdiv((int)tok, a.dN), n = (int)tok - bw * a.N;
            const int b = fastdiv(bw, a.dnW), wlin = bw - b * a.nW;
            const int i0 = fastdiv(wlin, a.dnw12), r1 = wlin - i0 * a.nw[1] * a.nw[2], i1 = fastdiv(r1, a.dnw2), i2 = r1 - i1 * a.nw[2];
            const int j0 = fastdiv(n, a.dw12), r2 = n - j0 * a.w[1] * a.w[2], j1 = fastdiv(r2, a.dw2), j2 = r2 - j1 * a.w[2];
            const int ii[3] = {i0, i1, i2}, jj[3] = {j0, j1, j2};
            bool ok = true;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int v = pmod(ii[d] * a.w[d] + jj[d] + a.s[d], a.P[d]) - a.f[d];
                if (!a.circ[d]) ok = ok && v >= 0 && v < a.D[d];
            }
            if (!ok) {
                const long long wout = (long long)b * a.nW + i0 * a.sw[0] + i1 * a.sw[1] + i2 * a.sw[2];
                list[atomicAdd(count, 1)] = (int)(wout * a.N + n);          // row index (< 2^31: win_setup)
            }
        }
        __syncthreads();
        const int np = *count;
        if (tl < lanes) {
            // eight listed rows in flight per thread (a read-then-add loop pays the L2 latency once per row)
            for (int e0 = tl; e0 < np; e0 += 8 * lanes) {
                const float* rows[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int e = e0 + t * lanes;
                    rows[t] = a.src + (long long)list[e < np ? e : e0] * a.C + 4 * q_lo;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c4 = cq + u * cw;
                    if (u < nq && c4 < C4) {
                        f32x4 v[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) v[t] = *reinterpret_cast<const f32x4*>(rows[t] + 4 * c4);
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            if (e0 + t * lanes < np) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) acc[u][k] += v[t][k];
                            }
                    }
                }
            }
        }
        __syncthreads();
    }
    if (tl < lanes) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < nq)
#pragma unroll
                for (int k = 0; k < 4; ++k) red[((tl * nq + u) * cw + cq) * 4 + k] = acc[u][k];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 4 * cw * nq; e += 256) {
        const int u = e / (4 * cw), r = e - u * 4 * cw;                 // quad group u, element r = 4 cq + k
        const int c = 4 * (u * cw) + r;
        if (c < 4 * C4) {
            float t = 0.f;
            for (int l = 0; l < lanes; ++l) t += red[(l * nq + u) * cw * 4 + r];
            if (t != 0.f) atomic_add_f32(&out[c_lo + c], t);
        }
    }
}

__device__ __forceinline__ f32x4 win_read(const WinDev& a, int b, const int (&p)[3], int c4) {
    int i[3], j[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int r = pmod(p[d] - a.s[d], a.P[d]);
        i[d] = fastdiv(r, a.dwd[d]);
        j[d] = r - i[d] * a.w[d];
    }
    const long long wout = (long long)b * a.nW + i[0] * a.sw[0] + i[1] * a.sw[1] + i[2] * a.sw[2];
    const int n = (j[0] * a.w[1] + j[1]) * a.w[2] + j[2];
    return win_load4(a.src, (wout * a.N + n) * a.C + 4 * c4, a.src_bf16);
}

__global__ __launch_bounds__(256) void win_scatter_kernel(WinDev a) {
    const int C4 = a.C >> 2;
    const long long total = (long long)a.B * a.D[0] * a.D[1] * a.D[2] * C4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int tok = fastdiv((int)e, a.dC4);
        const int c4 = (int)e - tok * C4;
        const int t1 = fastdiv(tok, a.dD2);
        const int q2 = tok - t1 * a.D[2];
        const int t0 = fastdiv(t1, a.dD1);
        const int q1 = t1 - t0 * a.D[1];
        const int b = fastdiv(t0, a.dD0), q0 = t0 - b * a.D[0];
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!a.dup) {
            const int p[3] = {q0 + a.f[0], q1 + a.f[1], q2 + a.f[2]};
            acc = win_read(a, b, p, c4);
        } else {
            // every padded position that holds a copy of (q0,q1,q2): p_d = q_d + f_d + k D_d inside [0, P_d), circular axes only
            for (int p0 = q0 + a.f[0] - (a.circ[0] ? ((q0 + a.f[0]) / a.D[0]) * a.D[0] : 0); p0 < a.P[0]; p0 += a.D[0]) {
                for (int p1 = q1 + a.f[1] - (a.circ[1] ? ((q1 + a.f[1]) / a.D[1]) * a.D[1] : 0); p1 < a.P[1]; p1 += a.D[1]) {
                    for (int p2 = q2 + a.f[2] - (a.circ[2] ? ((q2 + a.f[2]) / a.D[2]) * a.D[2] : 0); p2 < a.P[2]; p2 += a.D[2]) {
                        const int p[3] = {p0, p1, p2};
                        const f32x4 v = win_read(a, b, p, c4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[k] += v[k];
                        if (!a.circ[2]) break;
                    }
                    if (!a.circ[1]) break;
                }
                if (!a.circ[0]) break;
            }
        }
        if (a.scale) {
            const float sc = a.scale[b];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] *= sc;
        }
        if (a.res) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(a.res + (long long)tok * a.C + 4 * c4);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += r[k];
        }
        win_store4(a.dst, (long long)tok * a.C + 4 * c4, a.dst_bf16, acc);
    }
}

// Patch merging gather (PatchMerging.forward, src/nsbench/models/swintransformer/swin_transformer.py:291-312; dlwpbench twin):
// out[b][i][j][q C + c] = x[b][2 i + (q & 1)][2 j + (q >> 1)][c], zero beyond an odd H / W -- the reference's pad + four strided
// slices + cat.  BWD is the adjoint: gx[b][h][w][c] = g[b][h / 2][w / 2][((w & 1) * 2 + (h & 1)) C + c].
template <bool BWD, int VEC>
__global__ __launch_bounds__(256) void patch_merge_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int H, int W,
                                                          int C) {
    const int H2 = (H + 1) / 2, W2 = (W + 1) / 2, CV = C / VEC;
    const long long total = BWD ? (long long)B * H * W * CV : (long long)B * H2 * W2 * 4 * CV;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % CV) * VEC;
        long long r = e / CV;
        int h, w, q;
        if (BWD) {
            w = (int)(r % W); r /= W;
            h = (int)(r % H); r /= H;
            q = (w & 1) * 2 + (h & 1);
        } else {
            q = (int)(r & 3); r >>= 2;
            const int j = (int)(r % W2); r /= W2;
            const int i = (int)(r % H2); r /= H2;
            h = 2 * i + (q & 1); w = 2 * j + (q >> 1);
        }
        const int b = (int)r;
        const long long xo = (((long long)b * H + h) * W + w) * C + c;
        const long long mo = ((((long long)b * H2 + h / 2) * W2 + w / 2) * 4 + q) * C + c;
        if (BWD) {
            if (VEC == 4) *reinterpret_cast<f32x4*>(dst + xo) = *reinterpret_cast<const f32x4*>(src + mo);
            else dst[xo] = src[mo];
        } else {
            const bool ok = h < H && w < W;
            if (VEC == 4) *reinterpret_cast<f32x4*>(dst + mo) = ok ? *reinterpret_cast<const f32x4*>(src + xo) : f32x4{0.f, 0.f, 0.f, 0.f};
            else dst[mo] = ok ? src[xo] : 0.f;
        }
    }
}

// Transposed convolution with kernel == stride on channels-last tokens (Swin U-decoder: nn.ConvTranspose2d(k, stride k) + GELU,
// src/nsbench/models/swintransformer/swin_transformer.py:580-588): the GEMM y[m][(o, i, j)] = x[m] . W[c][(o, i, j)] leaves the
// matrix cores in (pixel, o, i, j) order; this kernel adds the bias, applies the activation and interleaves the k x k blocks
// into the up-sampled token grid out[b][h k + i][w k + j][coff + o] (row pitch ctot: the result can land inside the concat
// buffer of the next decoder level).  One thread per (pixel m, channel o): its k*k inputs are contiguous, every store is
// coalesced over o.  BWD: gy = gout * act'(y + bias) back in GEMM order, bias gradient summed per workgroup in LDS.
template <bool BWD>
__global__ __launch_bounds__(256) void upconv_shuffle_kernel(const float* __restrict__ y, const float* __restrict__ bias,
                                                             const float* __restrict__ gout, float* __restrict__ dst,
                                                             float* gbias, int B, int H, int W, int O, int kh, int kw, int ctot,
                                                             int coff, int act) {
    __shared__ float red[256];                // BWD: per-thread bias-gradient partials, [pixel slot][channel]
    const int kk = kh * kw, Ho = H * kh, Wo = W * kw;
    const long long Mtot = (long long)B * H * W;
    // a thread keeps ONE channel for all its pixels (its bias-gradient partial stays in a register): OT channels x PPB pixels
    // per workgroup step, consecutive threads on consecutive channels (coalesced)
    const int OT = O < 256 ? O : 256, PPB = 256 / OT;
    const int ol = threadIdx.x % OT, ms = threadIdx.x / OT;
    for (int ob = 0; ob < O; ob += OT) {          // one pass unless O > 256; the bound is uniform (barriers inside)
        const int o = ob + ol;
        const bool ov = o < O;
        const float bo = (bias && ov) ? bias[o] : 0.f;
        float gsum = 0.f;
        if (ms < PPB && ov) {
            for (long long m = (long long)blockIdx.x * PPB + ms; m < Mtot; m += (long long)gridDim.x * PPB) {
                const int w = (int)(m % W);
                const long long r = m / W;
                const int h = (int)(r % H), b = (int)(r / H);
                const long long yo = m * (long long)O * kk + (long long)o * kk;
                for (int ij = 0; ij < kk; ++ij) {
                    const int i = ij / kw, j = ij - i * kw;
                    const long long po = (((long long)b * Ho + h * kh + i) * Wo + w * kw + j) * ctot + coff + o;
                    const float z = y[yo + ij] + bo;
                    if (BWD) {
                        const float g = gout[po] * (act == 1 ? gelu_grad_f(z) : 1.f);
                        dst[yo + ij] = g;
                        gsum += g;
                    } else {
                        dst[po] = act == 1 ? gelu_f(z) : z;
                    }
                }
            }
        }
        if (BWD && gbias) {
            red[threadIdx.x] = gsum;
            __syncthreads();
            if (threadIdx.x < OT && ov) {         // here ms == 0: o = ob + threadIdx.x
                float t = 0.f;
                for (int q = 0; q < PPB; ++q) t += red[q * OT + threadIdx.x];
                if (t != 0.f) atomic_add_f32(&gbias[o], t);
            }
            __syncthreads();
        }
    }
}

int win_setup(WinDev& a, const float* src, float* dst, int B, int C, const int* D, const int* P, const int* f, const int* s,
              const int* w, const long long* sw, const int* circ, const char* who) {
    DLWP_REQUIRE(src && dst && D && P && f && s && w && sw && circ && B > 0 && C > 0, DLWP_E_INVALID, "%s: bad argument", who);
    DLWP_REQUIRE(C % 4 == 0, DLWP_E_UNSUPPORTED, "%s: channel count %d must be a multiple of 4", who, C);
    a.src = src; a.dst = dst; a.B = B; a.C = C; a.nW = 1; a.N = 1;
    for (int d = 0; d < 3; ++d) {
        DLWP_REQUIRE(D[d] > 0 && w[d] > 0 && P[d] >= D[d] + f[d] && P[d] % w[d] == 0 && f[d] >= 0, DLWP_E_INVALID,
                     "%s: axis %d: size %d, front pad %d, padded %d, window %d are inconsistent", who, d, D[d], f[d], P[d], w[d]);
        a.D[d] = D[d]; a.P[d] = P[d]; a.f[d] = f[d]; a.s[d] = ((s[d] % P[d]) + P[d]) % P[d]; a.w[d] = w[d]; a.nw[d] = P[d] / w[d];
        a.sw[d] = sw[d]; a.circ[d] = circ[d] != 0;
        a.nW *= a.nw[d]; a.N *= w[d];
        a.dwd[d] = make_fastdiv(w[d]);
    }
    const long long toks_w = (long long)B * a.nW * a.N, toks_x = (long long)B * D[0] * D[1] * D[2], C4 = C / 4;
    const long long total = std::max(toks_w, toks_x) * C4;
    // the umulhi divisions are exact while numerator * divisor < 2^32 (common.hip.h FastDiv)
    const long long lim = 1ll << 32;
    const bool fits = total < (1ll << 31) && total * C4 < lim && toks_w * a.N < lim && (long long)B * a.nW * a.nW < lim &&
                      toks_x * D[2] < lim && (long long)B * D[0] * D[1] * D[1] < lim && (long long)B * D[0] * D[0] < lim &&
                      (long long)a.N * w[1] * w[2] < lim && (long long)a.nW * a.nw[1] * a.nw[2] < lim;
    DLWP_REQUIRE(fits, DLWP_E_UNSUPPORTED, "%s: %lld vector elements exceed the 32-bit index arithmetic of this kernel; split the batch", who, total);
    a.dC4 = make_fastdiv(C / 4); a.dN = make_fastdiv(a.N); a.dnW = make_fastdiv(a.nW);
    a.dw12 = make_fastdiv(w[1] * w[2]); a.dw2 = make_fastdiv(w[2]);
    a.dnw12 = make_fastdiv(a.nw[1] * a.nw[2]); a.dnw2 = make_fastdiv(a.nw[2]);
    a.dD2 = make_fastdiv(D[2]); a.dD1 = make_fastdiv(D[1]); a.dD0 = make_fastdiv(D[0]);
    return DLWP_OK;
}

int grid_for(long long n) { return (int)std::min<long long>((n + 255) / 256, 8192); }

}  // namespace

extern "C" int dlwp_window_gather(const float* x, float* windows, int B, int C, const int* dims, const int* padded,
                                  const int* front, const int* shift, const int* window, const long long* wstride,
                                  const int* circular, void* stream) {
    WinDev a{};
    int rc = win_setup(a, x, windows, B, C, dims, padded, front, shift, window, wstride, circular, "window_gather");
    if (rc) return rc;
    hipLaunchKernelGGL(win_gather_kernel, dim3(grid_for((long long)B * a.nW * a.N * (C / 4))), dim3(256), 0, (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

// dlwp_window_gather / dlwp_window_gather_fill with storage flags (bit 0: x is a bf16 array, bit 1: windows is one) and an optional
// per-sample scale [B] applied to the gathered values (the adjoint of dlwp_window_scatter_ex's scale)
extern "C" int dlwp_window_gather_ex(const void* x, const float* fill, const float* scale, void* windows, int B, int C, const int* dims,
                                     const int* padded, const int* front, const int* shift, const int* window,
                                     const long long* wstride, const int* circular, int flags, void* stream) {
    WinDev a{};
    int rc = win_setup(a, (const float*)x, (float*)windows, B, C, dims, padded, front, shift, window, wstride, circular, "window_gather_ex");
    if (rc) return rc;
    DLWP_REQUIRE(flags >= 0 && flags < 4, DLWP_E_INVALID, "window_gather_ex: flags is a mask of 1 (x bf16) | 2 (windows bf16)");
    DLWP_REQUIRE(!fill || (reinterpret_cast<uintptr_t>(fill) & 15) == 0, DLWP_E_INVALID, "window_gather_ex: fill must be 16-byte aligned");
    a.fill = fill;
    a.scale = scale;
    a.src_bf16 = flags & 1;
    a.dst_bf16 = (flags >> 1) & 1;
    hipLaunchKernelGGL(win_gather_kernel, dim3(grid_for((long long)B * a.nW * a.N * (C / 4))), dim3(256), 0, (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

// partition of a tensor that a token-wise Linear layer has ALREADY been applied to: the padded positions hold `fill` (= that
// layer's bias) instead of zero.  Linear commutes with the gather, so "pad, then Linear over every window token" (reference:
// panguweather.py:283-292, EarthAttention3D.qkv on the padded windows) equals "Linear over the real tokens, then pad with the
// bias" -- and the GEMM runs on the real tokens only (Pangu 128 x 256, window (2,7,7): 32,768 instead of 68,894 rows).
extern "C" int dlwp_window_gather_fill(const float* x, const float* fill, float* windows, int B, int C, const int* dims,
                                       const int* padded, const int* front, const int* shift, const int* window,
                                       const long long* wstride, const int* circular, void* stream) {
    WinDev a{};
    int rc = win_setup(a, x, windows, B, C, dims, padded, front, shift, window, wstride, circular, "window_gather_fill");
    if (rc) return rc;
    DLWP_REQUIRE(!fill || (reinterpret_cast<uintptr_t>(fill) & 15) == 0, DLWP_E_INVALID, "window_gather_fill: fill must be 16-byte aligned");
    a.fill = fill;
    hipLaunchKernelGGL(win_gather_kernel, dim3(grid_for((long long)B * a.nW * a.N * (C / 4))), dim3(256), 0, (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

// gfill[c] += sum of g_windows[.][.][c] over the padded (non-circular) positions, channels c >= c_lo (a multiple of 4): gradient
// of dlwp_window_gather_fill's fill.  c_lo lets the caller skip channels whose padded rows are known to be zero (the query third of
// a qkv gradient whose padded rows were not computed, dlwp_window_attn_bwd_qrange).
extern "C" int dlwp_window_pad_colsum(const float* g_windows, float* gfill, int B, int C, const int* dims, const int* padded,
                                      const int* front, const int* shift, const int* window, const long long* wstride,
                                      const int* circular, int c_lo, void* stream) {
    WinDev a{};
    int rc = win_setup(a, g_windows, gfill, B, C, dims, padded, front, shift, window, wstride, circular, "window_pad_colsum");
    if (rc) return rc;
    DLWP_REQUIRE(c_lo >= 0 && c_lo < C && c_lo % 4 == 0, DLWP_E_INVALID, "window_pad_colsum: c_lo must be a multiple of 4 inside [0, C)");
    const int C4 = (C - c_lo) / 4, cw = std::min(C4, 256), lanes = 256 / cw, nq = (C4 + cw - 1) / cw;
    DLWP_REQUIRE(nq <= 4, DLWP_E_UNSUPPORTED, "window_pad_colsum: at most 4096 channels");
    const long long ntok = (long long)B * a.nW * a.N;
    const int grid = (int)std::max<long long>(1, std::min<long long>((ntok + 255) / 256, 1024));
    const size_t lds = sizeof(float) * (size_t)lanes * cw * 4 * nq + sizeof(int) * 260;
    hipLaunchKernelGGL(win_pad_colsum_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, a, gfill, c_lo);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_window_scatter(const float* windows, float* x, int B, int C, const int* dims, const int* padded,
                                   const int* front, const int* shift, const int* window, const long long* wstride,
                                   const int* circular, int sum_copies, void* stream) {
    return dlwp_window_scatter_add(windows, nullptr, x, B, C, dims, padded, front, shift, window, wstride, circular, sum_copies,
                                   stream);
}

extern "C" int dlwp_upconv_shuffle(const float* y, const float* bias, const float* gout, float* dst, float* gbias, int B, int H,
                                   int W, int O, int kh, int kw, int ctot, int coff, int act, int backward, void* stream) {
    DLWP_REQUIRE(y && dst && B > 0 && H > 0 && W > 0 && O > 0 && kh > 0 && kw > 0 && ctot >= coff + O && coff >= 0 &&
                     (act == 0 || act == 1) && (!backward || gout),
                 DLWP_E_INVALID, "upconv_shuffle: bad argument");
    const long long total = (long long)B * H * W * O;
    const dim3 grid(std::min(grid_for(total), 2048));
    if (backward)
        hipLaunchKernelGGL(upconv_shuffle_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, y, bias, gout, dst,
                           gbias, B, H, W, O, kh, kw, ctot, coff, act);
    else
        hipLaunchKernelGGL(upconv_shuffle_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, y, bias, gout, dst, gbias, B, H,
                           W, O, kh, kw, ctot, coff, act);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_patch_merge(const float* src, float* dst, int B, int H, int W, int C, int backward, void* stream) {
    DLWP_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && C > 0, DLWP_E_INVALID, "patch_merge: bad argument");
    const bool vec = C % 4 == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0;
    const long long H2 = (H + 1) / 2, W2 = (W + 1) / 2;
    const long long total = (backward ? (long long)B * H * W : (long long)B * H2 * W2 * 4) * (vec ? C / 4 : C);
    const dim3 grid(grid_for(total));
    if (backward) {
        if (vec) hipLaunchKernelGGL((patch_merge_kernel<true, 4>), grid, dim3(256), 0, (hipStream_t)stream, src, dst, B, H, W, C);
        else hipLaunchKernelGGL((patch_merge_kernel<true, 1>), grid, dim3(256), 0, (hipStream_t)stream, src, dst, B, H, W, C);
    } else {
        if (vec) hipLaunchKernelGGL((patch_merge_kernel<false, 4>), grid, dim3(256), 0, (hipStream_t)stream, src, dst, B, H, W, C);
        else hipLaunchKernelGGL((patch_merge_kernel<false, 1>), grid, dim3(256), 0, (hipStream_t)stream, src, dst, B, H, W, C);
    }
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

// dlwp_window_scatter_add with storage flags (bit 0: windows is a bf16 array, bit 1: x is one; the residual stays fp32) and an optional
// per-sample scale [B]: x[b] = residual[b] + scale[b] * scatter(windows)[b] -- reverse + stochastic depth + skip connection in one pass
extern "C" int dlwp_window_scatter_ex(const void* windows, const float* residual, const float* scale, void* x, int B, int C,
                                      const int* dims, const int* padded, const int* front, const int* shift, const int* window,
                                      const long long* wstride, const int* circular, int sum_copies, int flags, void* stream) {
    WinDev a{};
    int rc = win_setup(a, (const float*)windows, (float*)x, B, C, dims, padded, front, shift, window, wstride, circular, "window_scatter_ex");
    if (rc) return rc;
    DLWP_REQUIRE(flags >= 0 && flags < 4, DLWP_E_INVALID, "window_scatter_ex: flags is a mask of 1 (windows bf16) | 2 (x bf16)");
    a.res = residual;
    a.scale = scale;
    a.dup = sum_copies != 0;
    a.src_bf16 = flags & 1;
    a.dst_bf16 = (flags >> 1) & 1;
    hipLaunchKernelGGL(win_scatter_kernel, dim3(grid_for((long long)B * dims[0] * dims[1] * dims[2] * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_window_scatter_add(const float* windows, const float* residual, float* x, int B, int C, const int* dims,
                                       const int* padded, const int* front, const int* shift, const int* window,
                                       const long long* wstride, const int* circular, int sum_copies, void* stream) {
    WinDev a{};
    int rc = win_setup(a, windows, x, B, C, dims, padded, front, shift, window, wstride, circular, "window_scatter");
    if (rc) return rc;
    a.res = residual;
    a.dup = sum_copies != 0;
    hipLaunchKernelGGL(win_scatter_kernel, dim3(grid_for((long long)B * dims[0] * dims[1] * dims[2] * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

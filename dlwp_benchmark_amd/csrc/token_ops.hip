// The GEMM family of the token-level layers of the AFNO / Swin / Pangu / SFNO blocks: MFMA GEMM with fused bias / GELU / residual
// epilogue (nn.Linear, patch embedding, heads, 1x1 convolutions), fp32 or bf16 operands, batched and grouped forms, the LDS-DMA
// bf16 kernels and the weight-gradient products.  (LayerNorm, activation backward, column sums and the bf16 cast live in
// norm_ops.hip since round 4.)
// Reference call sites (file:line under /root/reference/src/nsbench/models):
//   nn.Linear      fourcastnet/fourcastnet.py:44-46,233 (Mlp.fc1/fc2, head); swintransformer/
//                  swin_transformer.py:35-39,131,153,274 (Mlp, qkv, proj, PatchMerging.reduction)
//
// GEMM: C[M,N] = epilogue(op(A)[M,K] . op(B)[K,N]), row-major operands with leading dimensions and
// transpose flags so that the three products of a Linear layer (y = x W^T, gx = gy W, gW = gy^T x)
// use one kernel family.  64x64 output tile per workgroup, K-step 32, four waves of 32x32 (2x2 MFMA
// 16x16x4 f32 blocks, permuted-k operand order).  Operand tiles are fetched with 16-byte loads along
// whichever dimension is contiguous in memory, kept in the matching LDS layout ([row][k] -> one b128
// fragment read, [k][row] -> four b32 reads), and double-buffered: the global loads of K-step i+1 are in
// flight while step i runs on the matrix cores, one barrier per step.  Reductions over the token
// dimension (gW: K = tokens, few output tiles) are split along K across workgroups (grid.z) and
// combined with float atomics into a zeroed C.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include "common.hip.h"
#include "dlwpmi_internal.h"

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;    // [row][k] layout: 16-byte aligned rows
// tile edge 64 * T (T = 1: 64 x 64 per workgroup, T = 2: 128 x 128 for the large products)
template <int T> struct Tile {
    static constexpr int ROWS = 64 * T;
    static constexpr int LDR = ROWS + 4;                                   // [k][row] layout
    static constexpr int FLOATS = ROWS * LDK > BK * LDR ? ROWS * LDK : BK * LDR;
};

struct GemmDev {
    const float *A, *B, *bias, *residual;
    float *C, *preact, *rowsum;
    int M, N, K, lda, ldb, ldc, act, accumulate, kchunk, splits;
    // strided-batched mode (nbatch > 1, no split-K): batch z = z1 * nb2 + z2, operand offsets z1 * s?1 + z2 * s?2
    int nbatch, nb2, res_pre;
    int ntn, ntm;              // column / row tile counts (1-D tile grid, see the XCD-aware order in the kernel)
    int vec_epi;               // epilogue through an LDS tile with 16-byte global accesses (alignment checked on the host)
    long long sA1, sA2, sB1, sB2, sC1, sC2, sR1, sR2, sBi1, sBi2;
    float act_param;           // soft-shrink threshold (act == 3)
    // channels-first extensions (the wide FNO path, fno_wide.hip: C[out channel][pixel] = W . X per sample)
    int bias_row;              // bias indexed by the output ROW m (1x1 convolution on [C, H*W] fields) instead of the column
    int act_b;                 // GELU applied to the B operand when its tile is committed to LDS (weight gradient of a layer
                               // whose input was activated on load: gW = g . gelu(x)^T without a stored gelu(x))
    int atomic_out;            // every workgroup ADDS its tile with float atomics (batches that share one output matrix)
    // storage types (dlwp_gemm_mixed): bit 0 A, bit 1 B, bit 2 C and preact, bit 3 residual hold bf16 in memory (the pointers
    // above are then __bf16*, leading dimensions and batch strides stay in elements).  Accumulation and epilogue run in fp32.
    int dt;
    float* slab;               // gemm_glds_tn_kernel: [K slices][M][N] partial products (plain stores; gemm_slab_reduce_kernel adds them)
    int xcd_splitk;            // gemm_kernel, nbatch == 1, splits % 8 == 0: the tiles of one K slice run on one XCD (see gemm_body)
    // per-sample scale of the product before the residual joins (dlwp_gemm_rowscale): C = (A.B + bias) * row_scale[m / scale_rows]
    // + residual -- stochastic depth (DropPath) of a residual branch inside the epilogue of the branch's last product
    const float* row_scale = nullptr;
    int scale_rows = 1;
    int wide_epi = 0;          // gemm_p8_kernel: the register epilogue in 128-byte rows (p8_rows8; needs N % 8 == 0 and 16-byte aligned rows)
};
constexpr int DT_A = 1, DT_B = 2, DT_C = 4, DT_R = 8;

// live accounting (prof.hip): algorithmic work of one product launch -- every operand read once, every output written once
static inline double gemm_prof_flops(const GemmDev& a) { return 2.0 * a.M * a.N * (double)a.K * a.nbatch; }
static inline double gemm_prof_bytes(const GemmDev& a) {
    const double nbA = (a.nbatch > 1 && a.sA1 == 0 && a.sA2 == 0) ? 1 : a.nbatch, nbB = (a.nbatch > 1 && a.sB1 == 0 && a.sB2 == 0) ? 1 : a.nbatch;
    const double nbC = (a.nbatch > 1 && a.sC1 == 0 && a.sC2 == 0) ? 1 : a.nbatch;
    const double eA = (a.dt & DT_A) ? 2 : 4, eB = (a.dt & DT_B) ? 2 : 4, eC = (a.dt & DT_C) ? 2 : 4, eR = (a.dt & DT_R) ? 2 : 4;
    const double mn = (double)a.M * a.N;
    return nbA * eA * a.M * (double)a.K + nbB * eB * a.N * (double)a.K + nbC * mn * eC * (1 + (a.preact ? 1 : 0) + (a.accumulate ? 1 : 0)) +
           (a.residual ? nbC * mn * eR : 0.0) + (a.bias ? 4.0 * (a.bias_row ? a.M : a.N) : 0.0);
}
// dlwp_prof_enable(2): the row names carry the product's shape and epilogue (one row per distinct product of the step)
struct GemmProfTag { char s[96]; };
static inline GemmProfTag gemm_prof_tag(const GemmDev& a) {
    GemmProfTag t; t.s[0] = 0;
    if (dlwp_prof_detail())
        snprintf(t.s, sizeof t.s, " M%d N%d K%d x%d dt%d%s%s%s%s%s", a.M, a.N, a.K, a.nbatch, a.dt, a.bias ? " bias" : "", a.act ? " act" : "",
                 a.preact ? " pre" : "", a.residual ? " res" : "", a.accumulate ? " acc" : "");
    return t;
}

// bf16-operand mode (dlwp_set_gemm_precision(1)): operands are rounded to bf16 when a tile is committed to LDS and
// multiplied by v_mfma_f32_16x16x32_bf16 (16x the fp32 MFMA rate) with fp32 accumulation -- the arithmetic of the
// reference's bf16-autocast runs (BASELINE configs C3-C5); tensors in HBM stay fp32.  LDS tiles are [row][k] bf16.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int BKB = 32;        // K-step of the bf16-operand mode.  A 64-deep step (fewer barriers, twice the bytes in flight
                               // per workgroup) was measured: 136 instead of 104 registers drop the occupancy from 4 to 3
                               // waves per SIMD and the net effect is -13 % ... +10 % depending on the shape (8192x512x256:
                               // 14.1 -> 16.2 us, 8192x256x512: 15.1 -> 13.6 us), so the 32-deep step stays
// K-step of a bf16-operand kernel: 64 only for the 128 x 128 weight-gradient product of two bf16 arrays (both operands
// row-contiguous, K = tokens, split-K): 16200-deep 3072 x 768: 300 -> 410 TFLOP/s.  Everywhere else the doubled LDS image
// halves the workgroups per CU and loses (16200 x 3072 x 768 forward: 479 -> 392; profiles/r02_gemm_bench.txt).
constexpr int gemm_bf_kstep(int s16m, int T, bool akc, bool bkc) { return s16m == 3 && T == 2 && !akc && !bkc ? 64 : BKB; }

// epilogue activations: 0 none, 1 GELU (erf), 2 ReLU, 3 soft-shrink(lambda); 4 is the backward form
// v * GELU'(aux) with aux read through the `residual` view (the saved pre-activation): the product g W of a Linear layer
// then leaves the kernel already multiplied by the derivative of the activation that fed it (no separate gelu_bwd pass)
constexpr int ACT_GELU_GRAD_MUL = 4;
// 5 / 6: the same for ReLU and soft-shrink(lambda) (the AFNO block MLP's activations): v * [aux > 0], v * [|aux| > lambda]
constexpr int ACT_RELU_GRAD_MUL = 5, ACT_SHRINK_GRAD_MUL = 6;
// 7 / 8 (round 5): GELU whose `preact` output receives the DERIVATIVE GELU'(v) instead of v (evaluated with the activation: three
// more instructions), and the matching backward form v * aux -- the layer's backward product then multiplies by a stored factor
// instead of evaluating an exponential and a reciprocal per element (its epilogue was VALU-bound on that: 131 vs 101 us for the
// FourCastNet product, profiles/r04_gemm_epilogue_ab.txt)
constexpr int ACT_GELU_STORE_D = 7, ACT_MUL = 8;
__host__ __device__ __forceinline__ bool act_is_grad_mul(int act) { return (act >= ACT_GELU_GRAD_MUL && act <= ACT_SHRINK_GRAD_MUL) || act == ACT_MUL; }
__device__ __forceinline__ float act_grad_mul(float v, float aux, int act, float lam) {
    if (act == ACT_MUL) return v * aux;
    if (act == ACT_GELU_GRAD_MUL) return v * gelu_grad_f(aux);
    if (act == ACT_RELU_GRAD_MUL) return aux > 0.f ? v : 0.f;
    return (act == ACT_SHRINK_GRAD_MUL && fabsf(aux) > lam) ? v : 0.f;
}
__device__ __forceinline__ float apply_act(float v, int act, float lam) {
    if (act == 1) return gelu_f(v);
    if (act == 2) return fmaxf(v, 0.f);
    if (act == 3) return v > lam ? v - lam : (v < -lam ? v + lam : 0.f);
    return v;
}

// One (64 T) x 32 operand tile.  KC: k is the contiguous dimension in memory (element (row, k) at
// base[row * ld + k]); otherwise the row index is contiguous (base[k * ld + row]).
// S16: the operand is a bf16 array, fully aligned (bf16-operand mode only): one 16-byte load carries 8 elements and goes to
// the LDS image as it is -- half the load instructions of the fp32 source, no conversion.
template <bool KC, bool VEC, int T, int BKT, bool S16 = false>
struct TileIO {
    static constexpr int ROWS = Tile<T>::ROWS, LDR = Tile<T>::LDR, KM = BKT / 32;
    static constexpr int LDRB = ROWS + 8;
    static constexpr int LDKT = BKT + 8;      // bf16 [row][k] image: row pitch for this K-step (16-byte aligned fragments)
    float v[S16 ? 1 : 8 * T * KM];
    bf16x8 h[S16 ? T * KM : 1];
    static __device__ __forceinline__ void coords16(int f, int& row, int& k) {
        if (KC) { row = f / (BKT / 8); k = 8 * (f % (BKT / 8)); }
        else { row = 8 * (f % (8 * T)); k = f / (8 * T); }
    }
    __device__ __forceinline__ void load16(const float* __restrict__ base, int ld, int row0, int nrows, int k0, int kend) {
        const __bf16* __restrict__ hb = reinterpret_cast<const __bf16*>(base);
#pragma unroll
        for (int q = 0; q < T * KM; ++q) {
            int row, k;
            coords16(threadIdx.x + 256 * q, row, k);
            const bool ok = row0 + row < nrows && k0 + k < kend;        // rows / k come in whole groups of 8 (host check)
            const long long off = KC ? (long long)(row0 + row) * ld + k0 + k : (long long)(k0 + k) * ld + row0 + row;
            bf16x8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (__bf16)0.f;
            if (ok) t = *reinterpret_cast<const bf16x8*>(hb + off);
            h[q] = t;
        }
    }
    __device__ __forceinline__ void store16(__bf16* S) const {
#pragma unroll
        for (int q = 0; q < T * KM; ++q) {
            int row, k;
            coords16(threadIdx.x + 256 * q, row, k);
            *reinterpret_cast<bf16x8*>(&S[KC ? row * LDKT + k : k * LDRB + row]) = h[q];
        }
    }
    static __device__ __forceinline__ void coords(int f, int& row, int& k) {
        if (VEC) {
            if (KC) { row = f / (BKT / 4); k = 4 * (f % (BKT / 4)); }
            else { row = 4 * (f % (16 * T)); k = f / (16 * T); }
        } else {
            if (KC) { row = f / BKT; k = f % BKT; }
            else { row = f % ROWS; k = f / ROWS; }
        }
    }
    // bf: the operand is stored as bf16 (same coordinates, 8-byte instead of 16-byte vector loads, widened in registers)
    __device__ __forceinline__ void load(const float* __restrict__ base, int ld, int row0, int nrows, int k0, int kend, bool bf) {
        const int tid = threadIdx.x;
        const __bf16* __restrict__ hb = reinterpret_cast<const __bf16*>(base);
        if (VEC) {
#pragma unroll
            for (int q = 0; q < 2 * T * KM; ++q) {
                int row, k;
                coords(tid + 256 * q, row, k);
                const bool ok = row0 + row < nrows && k0 + k < kend;
                const long long off = KC ? (long long)(row0 + row) * ld + k0 + k : (long long)(k0 + k) * ld + row0 + row;
                f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    if (bf) {
                        const bf16x4 hv = *reinterpret_cast<const bf16x4*>(hb + off);
#pragma unroll
                        for (int s = 0; s < 4; ++s) t[s] = (float)hv[s];
                    } else {
                        t = *reinterpret_cast<const f32x4*>(base + off);
                    }
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) v[4 * q + s] = t[s];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8 * T * KM; ++q) {
                int row, k;
                coords(tid + 256 * q, row, k);
                const bool ok = row0 + row < nrows && k0 + k < kend;
                const long long off = KC ? (long long)(row0 + row) * ld + k0 + k : (long long)(k0 + k) * ld + row0 + row;
                v[q] = ok ? (bf ? (float)hb[off] : base[off]) : 0.f;
            }
        }
    }
    __device__ __forceinline__ void apply_gelu() {
#pragma unroll
        for (int q = 0; q < 8 * T * KM; ++q) v[q] = gelu_f(v[q]);      // gelu(0) = 0: out-of-range zeros stay zero
    }
    __device__ __forceinline__ void store(float* S) const {
        static_assert(BKT == BK, "the fp32 LDS images are laid out for the 32-deep K-step");
        const int tid = threadIdx.x;
        if (VEC) {
#pragma unroll
            for (int q = 0; q < 2 * T * KM; ++q) {
                int row, k;
                coords(tid + 256 * q, row, k);
                *reinterpret_cast<f32x4*>(&S[KC ? row * LDK + k : k * LDR + row]) =
                    f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8 * T * KM; ++q) {
                int row, k;
                coords(tid + 256 * q, row, k);
                S[KC ? row * LDK + k : k * LDR + row] = v[q];
            }
        }
    }
    // bf16 images: [row][k] (row pitch LDKB) when k is contiguous in memory, [k][row] (row pitch LDRB) otherwise, so
    // that every float4 fetched from HBM becomes one packed 8-byte LDS write either way
    __device__ __forceinline__ void store_bf16(__bf16* S) const {
        const int tid = threadIdx.x;
        if (VEC) {
#pragma unroll
            for (int q = 0; q < 2 * T * KM; ++q) {
                int row, k;
                coords(tid + 256 * q, row, k);
                *reinterpret_cast<bf16x4*>(&S[KC ? row * LDKT + k : k * LDRB + row]) =
                    bf16x4{(__bf16)v[4 * q], (__bf16)v[4 * q + 1], (__bf16)v[4 * q + 2], (__bf16)v[4 * q + 3]};
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8 * T * KM; ++q) {
                int row, k;
                coords(tid + 256 * q, row, k);
                S[KC ? row * LDKT + k : k * LDRB + row] = (__bf16)v[q];
            }
        }
    }
    // bf16 MFMA fragment of chunk c: lane (r, g) holds row rb + r, k = 32c + 8g .. 32c + 8g+7 of the K-step.  From the [k][row] image
    // it comes through two hardware transpose reads (ds_read_b64_tr_b16: per 16-lane group a 4 x 16 block, lane 4q+p
    // addresses block row q / columns 4p.., lane i receives column i; tools/micro/tr_read.hip) -- EXEC is all ones here.
    static __device__ __forceinline__ bf16x8 frag_bf16(const __bf16* S, int rb, int c, int r, int g) {
        if (KC) return *reinterpret_cast<const bf16x8*>(&S[(rb + r) * LDKT + 32 * c + 8 * g]);
        typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
        const __bf16* p0 = &S[(32 * c + 8 * g + (r >> 2)) * LDRB + rb + 4 * (r & 3)];
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * LDRB));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
    // MFMA fragment of chunk c (k = 16c + 4g + s) for the 16 rows starting at rb
    static __device__ __forceinline__ f32x4 frag(const float* S, int rb, int c, int r, int g) {
        if (KC) return *reinterpret_cast<const f32x4*>(&S[(rb + r) * LDK + 16 * c + 4 * g]);
        f32x4 o;
#pragma unroll
        for (int s = 0; s < 4; ++s) o[s] = S[(16 * c + 4 * g + s) * LDR + rb + r];
        return o;
    }
};

// (lds_barrier(), common.hip.h: __syncthreads() would wait for every global store of the previous pass -- stamps: 34 k cycles for the
// four passes of a 256 x 256 tile, as long as twelve K-tiles)
// Store loop of an LDS-staged epilogue: NROWS tile rows per thread (row = tid / C4 + RPP * i, four consecutive columns each), in
// groups of G rows: every LDS read and every residual / accumulate load of a group is issued before the first use, the options
// are tested OUTSIDE the row loops (workgroup-uniform branches around whole loops), stores last.  The first form of this loop tested
// every option per row: 35 branches and a serialised load -> use -> store chain per row -- 34 k cycles for a 256 x 256 tile against
// 14.6 k for a bias-only loop (stamps, tools/probe_stamps_glds.py).
// G values of four consecutive columns n .. n + 3 in rows m[0..G): the epilogue arithmetic and the stores
template <int G>
__device__ __forceinline__ void epilogue_group(const GemmDev& a, f32x4 (&v)[G], const int (&m)[G], int n, const f32x4& bv) {
    const bool col_ok = n < a.N;
    auto put = [&](float* dst, long long o, const f32x4& val) {
        if (a.dt & DT_C)
            *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(dst) + o) = bf16x4{(__bf16)val[0], (__bf16)val[1], (__bf16)val[2], (__bf16)val[3]};
        else
            *reinterpret_cast<f32x4*>(dst + o) = val;
    };
    f32x4 rv[G];
    long long o[G];
    int mrow[G];
    bool ok[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        ok[i] = col_ok && m[i] < a.M;
        o[i] = ok[i] ? (long long)m[i] * a.ldc + n : 0;            // masked rows read element 0 (valid memory) and store nothing
        rv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        mrow[i] = ok[i] ? m[i] : 0;
    }
    if (a.residual) {
        if (a.dt & DT_R) {
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const bf16x4 hv = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(a.residual) + o[i]);
                rv[i] = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
            }
        } else {
#pragma unroll
            for (int i = 0; i < G; ++i) rv[i] = *reinterpret_cast<const f32x4*>(a.residual + o[i]);
        }
    }
    if (a.bias && a.bias_row) {                     // one bias value per output ROW (channels-first products)
        float br[G];
#pragma unroll
        for (int i = 0; i < G; ++i) br[i] = a.bias[mrow[i]];
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[i][k] += br[i];
    } else {
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[i][k] += bv[k];
    }
    if (act_is_grad_mul(a.act)) {
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[i][k] = act_grad_mul(v[i][k], rv[i][k], a.act, a.act_param);
    } else {
        if (a.res_pre) {
#pragma unroll
            for (int i = 0; i < G; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[i][k] += rv[i][k];
        }
        if (a.act == ACT_GELU_STORE_D) {
#pragma unroll
            for (int i = 0; i < G; ++i) {
                f32x4 d;
                gelu_both4(v[i], v[i], d);
                if (ok[i] && a.preact) put(a.preact, o[i], d);
            }
        } else if (a.preact) {
#pragma unroll
            for (int i = 0; i < G; ++i)
                if (ok[i]) put(a.preact, o[i], v[i]);
        }
        if (a.act && a.act != ACT_GELU_STORE_D) {
#pragma unroll
            for (int i = 0; i < G; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[i][k] = apply_act(v[i][k], a.act, a.act_param);
        }
        if (a.row_scale) {
            float sc[G];
#pragma unroll
            for (int i = 0; i < G; ++i) sc[i] = a.row_scale[mrow[i] / a.scale_rows];
#pragma unroll
            for (int i = 0; i < G; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[i][k] *= sc[i];
        }
        if (!a.res_pre && a.residual) {
#pragma unroll
            for (int i = 0; i < G; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[i][k] += rv[i][k];
        }
    }
    if (a.accumulate) {
        f32x4 cv[G];
#pragma unroll
        for (int i = 0; i < G; ++i) cv[i] = *reinterpret_cast<const f32x4*>(a.C + o[i]);
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[i][k] += cv[i][k];
    }
#pragma unroll
    for (int i = 0; i < G; ++i)
        if (ok[i]) put(a.C, o[i], v[i]);
}

// The same arithmetic on G row pieces of EIGHT consecutive columns n .. n + 7 (v[i][0]: columns n .. n + 3, v[i][1]: n + 4 .. n + 7): one
// 16-byte piece per lane where the tensor is bf16, two adjacent ones where it is fp32 -- the 256 x 256 kernel's register epilogue
// (round 6) after its cross-lane transposition, where eight lanes cover 128 bytes of one output row.
// `fill(v)` produces the G pieces AFTER the residual loads have been issued: the cross-lane transposition of the accumulators then runs
// under the latency of those loads (in the training step the stored derivative / residual comes from HBM, not from a cache: with two
// pieces per load -> use -> store round the gh = (g W2) * GELU' product took 187 us in the C5 step against 116 us back to back)
template <int G, class Fill>
__device__ __forceinline__ void epilogue_group8(const GemmDev& a, f32x4 (&v)[G][2], const int (&m)[G], int n, const f32x4 (&bv)[2], Fill fill) {
    const bool col_ok = n < a.N;                       // N % 8 == 0 on this path: a piece is whole or absent
    auto get = [&](const float* src, int dtbit, long long o, f32x4 (&out)[2]) {
        if (a.dt & dtbit) {
            const bf16x8 hv = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(src) + o);
            out[0] = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
            out[1] = f32x4{(float)hv[4], (float)hv[5], (float)hv[6], (float)hv[7]};
        } else {
            out[0] = *reinterpret_cast<const f32x4*>(src + o);
            out[1] = *reinterpret_cast<const f32x4*>(src + o + 4);
        }
    };
    auto put = [&](float* dst, long long o, const f32x4 (&val)[2]) {
        if (a.dt & DT_C) {
            *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(dst) + o) =
                bf16x8{(__bf16)val[0][0], (__bf16)val[0][1], (__bf16)val[0][2], (__bf16)val[0][3],
                       (__bf16)val[1][0], (__bf16)val[1][1], (__bf16)val[1][2], (__bf16)val[1][3]};
        } else {
            *reinterpret_cast<f32x4*>(dst + o) = val[0];
            *reinterpret_cast<f32x4*>(dst + o + 4) = val[1];
        }
    };
    f32x4 rv[G][2];
    long long o[G];
    int mrow[G];
    bool ok[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        ok[i] = col_ok && m[i] < a.M;
        o[i] = ok[i] ? (long long)m[i] * a.ldc + n : 0;            // masked rows read element 0 (valid memory) and store nothing
        rv[i][0] = rv[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        mrow[i] = ok[i] ? m[i] : 0;
    }
    if (a.residual) {
#pragma unroll
        for (int i = 0; i < G; ++i) get(a.residual, DT_R, o[i], rv[i]);
    }
    fill(v);
    if (a.bias && a.bias_row) {
        float br[G];
#pragma unroll
        for (int i = 0; i < G; ++i) br[i] = a.bias[mrow[i]];
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[i][h][k] += br[i];
    } else {
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[i][h][k] += bv[h][k];
    }
    if (act_is_grad_mul(a.act)) {
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[i][h][k] = act_grad_mul(v[i][h][k], rv[i][h][k], a.act, a.act_param);
    } else {
        if (a.res_pre) {
#pragma unroll
            for (int i = 0; i < G; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[i][h][k] += rv[i][h][k];
        }
        if (a.act == ACT_GELU_STORE_D) {
#pragma unroll
            for (int i = 0; i < G; ++i) {
                f32x4 d[2];
                gelu_both4(v[i][0], v[i][0], d[0]);
                gelu_both4(v[i][1], v[i][1], d[1]);
                if (ok[i] && a.preact) put(a.preact, o[i], d);
            }
        } else if (a.preact) {
#pragma unroll
            for (int i = 0; i < G; ++i)
                if (ok[i]) put(a.preact, o[i], v[i]);
        }
        if (a.act && a.act != ACT_GELU_STORE_D) {
#pragma unroll
            for (int i = 0; i < G; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[i][h][k] = apply_act(v[i][h][k], a.act, a.act_param);
        }
        if (a.row_scale) {
            float sc[G];
#pragma unroll
            for (int i = 0; i < G; ++i) sc[i] = a.row_scale[mrow[i] / a.scale_rows];
#pragma unroll
            for (int i = 0; i < G; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[i][h][k] *= sc[i];
        }
        if (!a.res_pre && a.residual) {
#pragma unroll
            for (int i = 0; i < G; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[i][h][k] += rv[i][h][k];
        }
    }
    if (a.accumulate) {                                 // (fp32 output only: checked on the host)
        f32x4 cv[G][2];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            cv[i][0] = *reinterpret_cast<const f32x4*>(a.C + o[i]);
            cv[i][1] = *reinterpret_cast<const f32x4*>(a.C + o[i] + 4);
        }
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[i][h][k] += cv[i][h][k];
    }
#pragma unroll
    for (int i = 0; i < G; ++i)
        if (ok[i]) put(a.C, o[i], v[i]);
}

// Cross-lane transposition of one 16-row block of the 256 x 256 kernel's C^T accumulators (lane (r, g) holds, for each of the wave's
// four 16-column blocks J, columns 16 J + 4 g .. + 3 of row r) into ROW pieces of eight consecutive columns, arranged so that the
// eight lanes of a row cover its 64 columns (128 bytes of bf16) in ONE store instruction:
//   step 1, v_permlane16_swap_b32 on (block 2 p, block 2 p + 1): lanes of even g end up with columns 4 g .. 4 g + 7 of block 2 p,
//           lanes of odd g with columns 4 (g - 1) .. 4 g + 3 of block 2 p + 1 -- a piece X_p starting at column 32 p + 16 (g & 1) + 4 (g & ~1);
//   step 2, two masked DPP moves (row_ror:8) per register: lanes r < 8 trade X_1 for row r + 8's X_0 -- afterwards
//           ya = (row r & 7, piece r >> 3), yb = (row 8 + (r & 7), piece r >> 3) with piece columns 32 (r >> 3) + 16 (g & 1) + 4 (g & ~1).
// 8 + 16 cross-lane instructions per block, no selects.  (round 6; stamps: the 8-byte / 32-bytes-per-row stores of the plain
// register epilogue took 25 k cycles per 256 x 256 bf16 tile, 43 % of a K = 768 tile)
#ifndef P8_WIDE_NI
#define P8_WIDE_NI 1
#endif
__device__ __forceinline__ void p8_rows8(const f32x4 (&accI)[4], f32x4 (&ya)[2], f32x4 (&yb)[2]) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef int i4 __attribute__((ext_vector_type(4)));
    // (whole-vector bit casts and constant element indices: with `v[q]` inside an unrolled loop hipcc (ROCm 7.2) kept only the q = 0
    // exchange and stored it to every element -- tools/micro/p8_rows8_check.hip)
    const u4 a0 = __builtin_bit_cast(u4, accI[0]), a1 = __builtin_bit_cast(u4, accI[1]);
    const u4 a2 = __builtin_bit_cast(u4, accI[2]), a3 = __builtin_bit_cast(u4, accI[3]);
#define P8_SWAP(A, B, q) const u2 s_##A##_##q = __builtin_amdgcn_permlane16_swap(A.q, B.q, false, false)
    P8_SWAP(a0, a1, x); P8_SWAP(a0, a1, y); P8_SWAP(a0, a1, z); P8_SWAP(a0, a1, w);
    P8_SWAP(a2, a3, x); P8_SWAP(a2, a3, y); P8_SWAP(a2, a3, z); P8_SWAP(a2, a3, w);
#undef P8_SWAP
    // X_p = {lo, hi}: lo = first operand after the swap, hi = second
    const i4 x0lo = i4{(int)s_a0_x.x, (int)s_a0_y.x, (int)s_a0_z.x, (int)s_a0_w.x}, x0hi = i4{(int)s_a0_x.y, (int)s_a0_y.y, (int)s_a0_z.y, (int)s_a0_w.y};
    const i4 x1lo = i4{(int)s_a2_x.x, (int)s_a2_y.x, (int)s_a2_z.x, (int)s_a2_w.x}, x1hi = i4{(int)s_a2_x.y, (int)s_a2_y.y, (int)s_a2_z.y, (int)s_a2_w.y};
    // ya: lanes r >= 8 (banks 2, 3) take X_1 of lane r - 8, lanes r < 8 keep their X_0;  yb: lanes r < 8 take X_0 of lane r + 8, the others keep X_1
#define P8_DPP(OLD, SRC, e, MASK) __builtin_amdgcn_update_dpp(OLD.e, SRC.e, 0x128, 0xf, MASK, false)
    ya[0] = __builtin_bit_cast(f32x4, (i4{P8_DPP(x0lo, x1lo, x, 0xc), P8_DPP(x0lo, x1lo, y, 0xc), P8_DPP(x0lo, x1lo, z, 0xc), P8_DPP(x0lo, x1lo, w, 0xc)}));
    ya[1] = __builtin_bit_cast(f32x4, (i4{P8_DPP(x0hi, x1hi, x, 0xc), P8_DPP(x0hi, x1hi, y, 0xc), P8_DPP(x0hi, x1hi, z, 0xc), P8_DPP(x0hi, x1hi, w, 0xc)}));
    yb[0] = __builtin_bit_cast(f32x4, (i4{P8_DPP(x1lo, x0lo, x, 0x3), P8_DPP(x1lo, x0lo, y, 0x3), P8_DPP(x1lo, x0lo, z, 0x3), P8_DPP(x1lo, x0lo, w, 0x3)}));
    yb[1] = __builtin_bit_cast(f32x4, (i4{P8_DPP(x1hi, x0hi, x, 0x3), P8_DPP(x1hi, x0hi, y, 0x3), P8_DPP(x1hi, x0hi, z, 0x3), P8_DPP(x1hi, x0hi, w, 0x3)}));
#undef P8_DPP
}

#ifndef GLDS_EPI_G
#define GLDS_EPI_G 2          // rows per thread in flight in the 128 x 128 LDS-DMA kernel's epilogue (8 rows per thread and half tile).  4 was measured
#endif                        // in round 6 (114 -> 168 VGPRs at three waves per SIMD, 7 - 10 spilled): gh 164 -> 158 us, fc1 142 -> 147 us in the C5 step: no gain
template <int NROWS, int RPP, int C4, int LDE, int G>      // G rows in flight per thread (registers: ~14 G)
__device__ __forceinline__ void epilogue_rows(const GemmDev& a, const float* tile, int tid, int m_base, int n, const f32x4& bv) {
    static_assert(NROWS % G == 0, "rows per thread come in whole groups");
    const int c4 = tid % C4;
#pragma unroll
    for (int i0 = 0; i0 < NROWS; i0 += G) {
        f32x4 v[G];
        int m[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int row = tid / C4 + RPP * (i0 + i);
            m[i] = m_base + row;
            v[i] = *reinterpret_cast<const f32x4*>(&tile[row * LDE + 4 * c4]);
        }
        epilogue_group<G>(a, v, m, n, bv);
    }
}

// bias gradient by-product: sum over k of op(A) rows (see dlwp_gemm's rowsum)
template <int T>
__device__ __forceinline__ void gemm_rowsum_flush(const GemmDev& a, const float (&rsum)[2 * T], int m0, int n0, int wm, int w, int r, int g) {
    if (!a.rowsum || n0 != 0 || (w & 1)) return;
#pragma unroll
    for (int i = 0; i < 2 * T; ++i) {
        float v = rsum[i];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        const int m = m0 + wm + 16 * i + r;
        if (g == 0 && m < a.M) atomic_add_f32(&a.rowsum[m], v);
    }
}

// VEC bit 0: 16-byte loads for the A tile, bit 1: for the B tile (alignment checked per operand on the host)
// S16M bit 0 / 1: operand A / B is an aligned bf16 array read through TileIO's 16-byte path (BF kernels only)
// (the kernel body: workgroup bx of ntiles_grid tiles, slice / batch bz; gemm_kernel launches one product, gemm_group_kernel several)
template <bool AKC, bool BKC, int VEC, int T, bool BF, int S16M = 0>
__device__ __forceinline__ void gemm_body(GemmDev a, const int bx, const int bz, const int ntiles_grid) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    constexpr int BMN = Tile<T>::ROWS, NT16 = 2 * T;   // 16-row MFMA tiles per wave and direction
    // K-step: 32, or 64 when both operands are bf16 arrays (S16M == 3: four staging registers per operand and K-step instead
    // of sixteen, so the deeper step no longer costs occupancy -- and it halves the barriers per K)
    constexpr int BKT = BF ? gemm_bf_kstep(S16M, T, AKC, BKC) : BK;
    // tile size in floats (bf16: two elements per float; the larger of the [row][k] and [k][row] images)
    constexpr int TF = BF ? (BMN * (BKT + 8) > BKT * (BMN + 8) ? BMN * (BKT + 8) : BKT * (BMN + 8)) / 2 : Tile<T>::FLOATS;
    float* As = gsm;                 // [2][TF]
    float* Bs = gsm + 2 * TF;        // [2][TF]
    const int lane = lane_id(), w = wave_id(), r = lane & 15, g = lane >> 4;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so consecutive ids never
    // share an L2.  Remap so that every XCD walks a contiguous range of tiles, row-panel by row-panel: the A panel that the
    // ntn column tiles of one row panel share is then fetched into ONE L2 instead of eight.
    int tile_id = bx, bzz = bz;
    if (a.nbatch == 1 && a.splits > 1 && a.splits % 8 == 0 && a.xcd_splitk) {
        // split-K products with few tiles (weight gradients: K = tokens): the tiles of ONE K slice read the same token rows, so a
        // slice's tiles run on one XCD, back to back (round 5; as in wgrad_multi_kernel).  The contiguous-range order below needs
        // gridDim.x % 8 == 0 to mean anything; with 12 tiles x 43 slices every tile fetched its own copy of the rows across the fabric
        const int id = bx + ntiles_grid * bz, xcd = id & 7, j = id >> 3;
        tile_id = j % ntiles_grid;
        bzz = (j / ntiles_grid) * 8 + xcd;
    } else {
        const int nt = ntiles_grid, full = (nt / 8) * 8;
        if (tile_id < full) tile_id = (tile_id % 8) * (nt / 8) + tile_id / 8;
    }
    // inside an XCD's range: groups of 8 row panels, column tile by column tile, so that the ~128 workgroups an XCD runs
    // at a time form a compact block (8 A panels x 16 B panels in flight) instead of one long row or column
    constexpr int GM = 8;
    const int grp = tile_id / (GM * a.ntn), within = tile_id - grp * GM * a.ntn;
    const int rows_in = min(GM, a.ntm - grp * GM);
    const int nt_ = within / rows_in, mt = grp * GM + (within - nt_ * rows_in);
    const int m0 = mt * BMN, n0 = nt_ * BMN;
    int zs = bzz;
    if (a.nbatch > 1) {
        const int zb = zs / a.splits;                 // batch index; zs % splits = K split within the batch
        zs -= zb * a.splits;
        const int z1 = zb / a.nb2, z2 = zb - z1 * a.nb2;
        // element offsets; a bf16 operand advances by half the bytes
        auto adv = [](const float* p, long long off, bool bf) {
            return bf ? reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(p) + off) : p + off;
        };
        a.A = adv(a.A, z1 * a.sA1 + z2 * a.sA2, a.dt & DT_A);
        a.B = adv(a.B, z1 * a.sB1 + z2 * a.sB2, a.dt & DT_B);
        const long long oc = z1 * a.sC1 + z2 * a.sC2;
        a.C = const_cast<float*>(adv(a.C, oc, a.dt & DT_C));
        if (a.preact) a.preact = const_cast<float*>(adv(a.preact, oc, a.dt & DT_C));
        if (a.residual) a.residual = adv(a.residual, z1 * a.sR1 + z2 * a.sR2, a.dt & DT_R);
        if (a.bias) a.bias += z1 * a.sBi1 + z2 * a.sBi2;
    }
    const int kbeg = zs * a.kchunk, kend = min(a.K, kbeg + a.kchunk);
    const int wm = (w >> 1) * 32 * T, wn = (w & 1) * 32 * T;
    f32x4 acc[NT16][NT16];
#pragma unroll
    for (int i = 0; i < NT16; ++i)
#pragma unroll
        for (int j = 0; j < NT16; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float rsum[NT16];                // sum over k of op(A) rows wm + 16 i + r (this lane's k slots)
#pragma unroll
    for (int i = 0; i < NT16; ++i) rsum[i] = 0.f;
    constexpr bool VA = (VEC & 1) != 0, VB = (VEC & 2) != 0;
    constexpr bool SA = BF && (S16M & 1), SB = BF && (S16M & 2);
    TileIO<AKC, VA, T, BKT, SA> ta;
    TileIO<BKC, VB, T, BKT, SB> tb;
    const int nk = (kend - kbeg + BKT - 1) / BKT;
    DLWP_STAMP(0);
    const bool abf = a.dt & DT_A, bbf = a.dt & DT_B;
    if constexpr (SA) ta.load16(a.A, a.lda, m0, a.M, kbeg, kend); else ta.load(a.A, a.lda, m0, a.M, kbeg, kend, abf);
    if constexpr (SB) tb.load16(a.B, a.ldb, n0, a.N, kbeg, kend); else tb.load(a.B, a.ldb, n0, a.N, kbeg, kend, bbf);
    if constexpr (!SB) { if (a.act_b) tb.apply_gelu(); }
    if constexpr (BF) {
        if constexpr (SA) ta.store16(reinterpret_cast<__bf16*>(As)); else ta.store_bf16(reinterpret_cast<__bf16*>(As));
        if constexpr (SB) tb.store16(reinterpret_cast<__bf16*>(Bs)); else tb.store_bf16(reinterpret_cast<__bf16*>(Bs));
    } else {
        ta.store(As);
        tb.store(Bs);
    }
    __syncthreads();
    DLWP_STAMP(1);
    for (int it = 0; it < nk; ++it) {
        const int cur = it & 1;
        if (it == 1) DLWP_STAMP(2);
        if (it == 2) DLWP_STAMP(3);
        if (it + 1 < nk) {
            if constexpr (SA) ta.load16(a.A, a.lda, m0, a.M, kbeg + (it + 1) * BKT, kend);
            else ta.load(a.A, a.lda, m0, a.M, kbeg + (it + 1) * BKT, kend, abf);
            if constexpr (SB) tb.load16(a.B, a.ldb, n0, a.N, kbeg + (it + 1) * BKT, kend);
            else tb.load(a.B, a.ldb, n0, a.N, kbeg + (it + 1) * BKT, kend, bbf);
        }
        if constexpr (BF) {
#pragma unroll
            for (int c = 0; c < BKT / 32; ++c) {
                bf16x8 af[NT16], bf[NT16];
#pragma unroll
                for (int i = 0; i < NT16; ++i) af[i] = TileIO<AKC, VA, T, BKT>::frag_bf16(reinterpret_cast<const __bf16*>(As + cur * TF), wm + 16 * i, c, r, g);
#pragma unroll
                for (int j = 0; j < NT16; ++j) bf[j] = TileIO<BKC, VB, T, BKT>::frag_bf16(reinterpret_cast<const __bf16*>(Bs + cur * TF), wn + 16 * j, c, r, g);
#pragma unroll
                for (int i = 0; i < NT16; ++i) {
                    if (a.rowsum) {
#pragma unroll
                        for (int s = 0; s < 8; ++s) rsum[i] += (float)af[i][s];
                    }
#pragma unroll
                    for (int j = 0; j < NT16; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
                }
            }
        } else
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x4 af[NT16], bf[NT16];
#pragma unroll
            for (int i = 0; i < NT16; ++i) af[i] = TileIO<AKC, VA, T, BKT>::frag(As + cur * TF, wm + 16 * i, c, r, g);
#pragma unroll
            for (int j = 0; j < NT16; ++j) bf[j] = TileIO<BKC, VB, T, BKT>::frag(Bs + cur * TF, wn + 16 * j, c, r, g);
#pragma unroll
            for (int i = 0; i < NT16; ++i) {
                rsum[i] += (af[i][0] + af[i][1]) + (af[i][2] + af[i][3]);
#pragma unroll
                for (int j = 0; j < NT16; ++j) acc[i][j] = mfma16_chunk(af[i], bf[j], acc[i][j]);
            }
        }
        if (it + 1 < nk) {
            if constexpr (!SB) { if (a.act_b) tb.apply_gelu(); }
            if constexpr (BF) {
                if constexpr (SA) ta.store16(reinterpret_cast<__bf16*>(As + (cur ^ 1) * TF));
                else ta.store_bf16(reinterpret_cast<__bf16*>(As + (cur ^ 1) * TF));
                if constexpr (SB) tb.store16(reinterpret_cast<__bf16*>(Bs + (cur ^ 1) * TF));
                else tb.store_bf16(reinterpret_cast<__bf16*>(Bs + (cur ^ 1) * TF));
            } else {
                ta.store(As + (cur ^ 1) * TF);
                tb.store(Bs + (cur ^ 1) * TF);
            }
        }
        __syncthreads();
    }
    DLWP_STAMP(4);
    gemm_rowsum_flush<T>(a, rsum, m0, n0, wm, w, r, g);
    {
        if (a.vec_epi) {
            // The accumulator layout gives each lane 4 rows x 1 column: stored directly, a wave-instruction touches four
            // 64-byte row segments (issue-bound, ~2 TB/s on the MLP epilogues; 2-byte scalar stores with a bf16 output).  Through
            // an LDS tile every global access of the epilogue (C, residual, pre-activation) becomes a 16-byte (bf16: 8-byte) one
            // on contiguous rows.  The 128 x 128 tile goes through the (dead) operand buffers in two halves of 64 rows.
            constexpr int EC = 64 * T, LDE = EC + 4, C4 = EC / 4, RPP = 256 / C4, NPASS = 64 / RPP;
            float* tile = gsm;                          // the operand buffers are dead (loop ended with a barrier)
            const int tid = threadIdx.x, c4 = tid % C4;
            const int n = n0 + 4 * c4;
            f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.bias && !a.bias_row && n < a.N) bv = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
            for (int half = 0; half < T; ++half) {
                if (half) lds_barrier();                // the previous half has been read out
                if (wm / 64 == half) {
                    const int wl = wm - 64 * half;
#pragma unroll
                    for (int i = 0; i < NT16; ++i)
#pragma unroll
                        for (int j = 0; j < NT16; ++j)
#pragma unroll
                            for (int q = 0; q < 4; ++q) tile[(wl + i * 16 + 4 * g + q) * LDE + wn + j * 16 + r] = acc[i][j][q];
                }
                lds_barrier();
                epilogue_rows<NPASS, RPP, C4, LDE, 2>(a, tile, tid, m0 + 64 * half, n, bv);
            }       // half
            DLWP_STAMP(5);
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < NT16; ++i)
#pragma unroll
        for (int j = 0; j < NT16; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + wm + i * 16 + 4 * g + q, n = n0 + wn + j * 16 + r;
                if (m < a.M && n < a.N) {
                    const long long o = (long long)m * a.ldc + n;
                    float v = acc[i][j][q];
                    if (a.splits > 1 || a.atomic_out) { atomic_add_f32(&a.C[o], v); continue; }   // C zeroed (or accumulate)
                    if (a.bias) v += a.bias[a.bias_row ? m : n];
                    const float rres = !a.residual ? 0.f
                                       : (a.dt & DT_R) ? (float)reinterpret_cast<const __bf16*>(a.residual)[o] : a.residual[o];
                    auto put1 = [&](float* dst, float val) {
                        if (a.dt & DT_C) reinterpret_cast<__bf16*>(dst)[o] = (__bf16)val;
                        else dst[o] = val;
                    };
                    if (act_is_grad_mul(a.act)) {
                        v = act_grad_mul(v, rres, a.act, a.act_param);
                    } else {
                        if (a.res_pre) v += rres;
                        if (a.act == ACT_GELU_STORE_D) {
                            float gv, dv;
                            gelu_both(v, gv, dv);
                            if (a.preact) put1(a.preact, dv);
                            v = gv;
                        } else {
                            if (a.preact) put1(a.preact, v);
                            v = apply_act(v, a.act, a.act_param);
                        }
                        if (a.row_scale) v *= a.row_scale[m / a.scale_rows];
                        if (!a.res_pre) v += rres;
                    }
                    put1(a.C, a.accumulate ? a.C[o] + v : v);
                }
            }
}

template <bool AKC, bool BKC, int VEC, int T, bool BF, int S16M = 0>
__global__ __launch_bounds__(256) void gemm_kernel(GemmDev a) {
    gemm_body<AKC, BKC, VEC, T, BF, S16M>(a, blockIdx.x, blockIdx.z, gridDim.x);
}

// up to three independent products in ONE launch (blockIdx.y picks the product): the weight gradients of a token MLP are three
// small, latency-bound split-K products (SFNO C3: 512 workgroups of ~20 us each on a 1024-slot chip) that do not depend on each other
struct GemmGroup { GemmDev g[3]; int s16m[3]; };      // s16m: bit 0 / 1 = operand A / B is an aligned bf16 array (16-byte loads, no widening)
template <bool AKC, bool BKC, int VEC, int T, bool BF, int S16M = 0>
__global__ __launch_bounds__(256) void gemm_group_kernel(GemmGroup gg) {
    const GemmDev a = gg.g[blockIdx.y];
    const int nt = a.ntn * a.ntm;
    if ((int)blockIdx.x >= nt || (long long)blockIdx.z >= a.nbatch * a.splits) return;
    if constexpr (BF) {
        switch (gg.s16m[blockIdx.y]) {
            case 3: gemm_body<AKC, BKC, VEC, T, true, 3>(a, blockIdx.x, blockIdx.z, nt); return;
            case 2: gemm_body<AKC, BKC, VEC, T, true, 2>(a, blockIdx.x, blockIdx.z, nt); return;
            case 1: gemm_body<AKC, BKC, VEC, T, true, 1>(a, blockIdx.x, blockIdx.z, nt); return;
            default: break;
        }
    }
    gemm_body<AKC, BKC, VEC, T, BF, S16M>(a, blockIdx.x, blockIdx.z, nt);
}

// ... and products of ANY layout (dlwp_gemm_group_begin / _end: a queue of independent small products launched together): the four
// layout bodies of the generic 64 x 64 configuration in one kernel, picked per product
struct GemmGroupAny { GemmDev g[3]; int layout[3]; int s16m[3]; };      // layout: 2 (A k-contiguous) | 1 (B k-contiguous); s16m == 3: both operands
                                                                         // aligned bf16 arrays (16-byte loads, no widening)
template <bool BF>
__global__ __launch_bounds__(256) void gemm_group_any_kernel(GemmGroupAny gg) {
    const GemmDev a = gg.g[blockIdx.y];
    const int nt = a.ntn * a.ntm;
    if ((int)blockIdx.x >= nt || (long long)blockIdx.z >= a.nbatch * a.splits) return;
    const int layout = gg.layout[blockIdx.y];
    if constexpr (BF) {
        if (gg.s16m[blockIdx.y] == 3) {
            switch (layout) {
                case 3: gemm_body<true, true, 3, 1, true, 3>(a, blockIdx.x, blockIdx.z, nt); return;
                case 2: gemm_body<true, false, 3, 1, true, 3>(a, blockIdx.x, blockIdx.z, nt); return;
                case 0: gemm_body<false, false, 3, 1, true, 3>(a, blockIdx.x, blockIdx.z, nt); return;
                default: break;
            }
        }
    }
    switch (layout) {
        case 3: gemm_body<true, true, 3, 1, BF, 0>(a, blockIdx.x, blockIdx.z, nt); break;
        case 2: gemm_body<true, false, 3, 1, BF, 0>(a, blockIdx.x, blockIdx.z, nt); break;
        case 1: gemm_body<false, true, 3, 1, BF, 0>(a, blockIdx.x, blockIdx.z, nt); break;
        default: gemm_body<false, false, 3, 1, BF, 0>(a, blockIdx.x, blockIdx.z, nt); break;
    }
}

template <bool AKC, bool BKC, int VEC, int T, bool BF, int S16M = 0>
int gemm_launch_t(const GemmDev& a, dim3 grid, hipStream_t s) {
    constexpr int RW = Tile<T>::ROWS;
    constexpr int KS = gemm_bf_kstep(S16M, T, AKC, BKC);
    const size_t lds = BF ? sizeof(float) * 4 * ((RW * (KS + 8) > KS * (RW + 8) ? RW * (KS + 8) : KS * (RW + 8)) / 2)
                          : sizeof(float) * 4 * Tile<T>::FLOATS;
    int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(gemm_kernel<AKC, BKC, VEC, T, BF, S16M>), lds, "gemm");
    if (rc) return rc;
    dlwp_prof_scope prof(s, gemm_prof_flops(a), gemm_prof_bytes(a), "gemm_kernel<%s, %s, %d, %d, %s, %d>%s", AKC ? "true" : "false",
                         BKC ? "true" : "false", VEC, T, BF ? "true" : "false", S16M, gemm_prof_tag(a).s);
    hipLaunchKernelGGL((gemm_kernel<AKC, BKC, VEC, T, BF, S16M>), grid, dim3(256), lds, s, a);
    return DLWP_OK;
}

int g_gemm_bf16 = 0;    // dlwp_set_gemm_precision

// ---- (round 3) both operands bf16 arrays, both k-contiguous ("NT": y = x W^T): 128 x 128 x 64 tiles staged by LDS-DMA.
// The register-staged kernel above spends a K-step of 32 on 4 global loads + 4 ds_write_b128 + a barrier per 16 MFMAs and
// reaches ~480 TFLOP/s at the FourCastNet shapes (19 % of the bf16 peak).  Here a K-step is 64 deep (32 MFMAs per wave), the
// operand tiles go global -> LDS with global_load_lds_dwordx4 (no staging registers, no LDS store pass), two stages, the loads of
// step k + 1 in flight across the barriers of step k (raw s_barrier + counted vmcnt: cdna_hip_programming.md section 5,
// "Pipelining across barriers").  LDS image of an operand tile: [128 rows][64 bf16] in 16-byte chunks, chunk c of row r stored
// at chunk position c ^ ((r >> 1) & 7) -- the LDS side of an LDS-DMA is lane-linear, so the permutation is applied to the
// per-lane GLOBAL address; a 16-lane group of a ds_read_b128 fragment read then covers all 64 banks.
// A weight-gradient ("TN", both tiles [k][row], K = tokens) variant was built and measured as well: unsplit it has 96 - 144
// workgroups, one per CU, and a K-step costs ~1.4 us per workgroup whatever the pipeline depth (two or three stages: 364 us at
// 3072 x 768 x 16200 against 183 us for the register-staged split-K kernel) -- the activation products only win because two
// workgroups per CU cover each other; the weight gradients stay on gemm_kernel.
// A four-stage ring (128 KB, three K-steps of loads in flight, one workgroup per CU) was measured too: 25 - 60 % slower wherever
// two workgroups per CU fit, equal at the 256-tile shapes -- the second workgroup hides more than the deeper ring does.
// Epilogue = the register-staged kernel's (bias, residual before / after the activation, GELU, GELU' multiply, stored
// pre-activation, fp32 or bf16 output, accumulate), through an LDS tile in two halves of 64 rows.
// BKC = false ("NN": gx = g W with W [N][K] read as the [k][n] operand): the B tile is [64 k][128 n], chunk c of k-row kr stored at
// chunk position c ^ (2 (kr & 3)); its MFMA fragments come through the hardware transpose read (ds_read_b64_tr_b16, as in
// TileIO::frag_bf16): per 16-lane group four k-rows x 32 bytes, 32 different banks.
constexpr int GT = 128, GK = 64;

// GN = 96 (round 6): a 128 x 96 output tile.  Every feature width of the C4 models is a multiple of 96 (96 .. 1536) and several of their
// products have just under one 128 x 128 tile per CU or pad a quarter of the last column tile (8192 x 384: 192 tiles -> 256 of 128 x 96;
// N = 96 / 192 / 288: 25 % padding -> none).  The B tile is still staged 128 rows / columns wide (the loader, the swizzles and the LDS
// image are the 128-wide kernel's; the rows past 96 re-read the next tile's or the clamped last rows and are never used); the wave grid
// stays 2 x 2 with 64 x 48 per wave (4 x 3 MFMA tiles) and the epilogue runs on 192 of the 256 threads (24 float4 columns x 8 rows).
// A ring of three / four stages with one barrier per step (loads of two / three K-steps in flight) was measured in round 6 for the products
// with one workgroup per CU: bit-identical, 3 - 6 % faster back to back at K = 384, nothing in the Pangu / Swin C4 steps (a lone workgroup's
// K-step is bound by its own LDS reads and MFMAs not overlapping, not by the latency of the next loads): not kept (profiles/r06_experiments.md).
template <bool BKC, int KD, int GN = 128>                  // KD = depth of a K-step: 64 (two workgroups per CU) or 32 (three / four)
__global__ __launch_bounds__(256) void gemm_glds_kernel(GemmDev a) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    __bf16* lds = reinterpret_cast<__bf16*>(gsm);          // [2 stages][A | B][128][KD]
    constexpr int TILE = GT * KD;                          // bf16 elements of one operand tile
    constexpr int NI = KD / 16;                            // LDS-DMA instructions per wave, operand and stage
    constexpr int CPR = KD / 8, LCPR = KD == 64 ? 3 : 2;   // 16-byte chunks per k-contiguous row
    const int lane = lane_id(), w = wave_id(), r = lane & 15, g = lane >> 4, tid = threadIdx.x;
    int tile_id = blockIdx.x;
    {
        const int nt = gridDim.x, full = (nt / 8) * 8;     // XCD-aware order, as in gemm_kernel
        if (tile_id < full) tile_id = (tile_id % 8) * (nt / 8) + tile_id / 8;
    }
    constexpr int GM = 8;
    const int grp = tile_id / (GM * a.ntn), within = tile_id - grp * GM * a.ntn;
    const int rows_in = min(GM, a.ntm - grp * GM);
    const int nt_ = within / rows_in, mt = grp * GM + (within - nt_ * rows_in);
    const int m0 = mt * GT, n0 = nt_ * GN;
    const __bf16* A = reinterpret_cast<const __bf16*>(a.A);
    const __bf16* B = reinterpret_cast<const __bf16*>(a.B);
    // chunk swizzle of a k-contiguous row: a 16-lane group of a ds_read_b128 fragment read (16 rows, one chunk column) must
    // cover 16 different 16-byte slots of the 256-byte bank row: 128-byte rows pair up, 64-byte rows come four to a bank row
    auto sw = [](int row) { return KD == 64 ? (row >> 1) & 7 : (row >> 2) & 3; };
    // per-lane source rows / chunks of the LDS-DMA instructions per operand: LDS chunk p = (4 i + w) * 64 + lane holds logical
    // chunk (p % CPR) ^ sw(row) of row p / CPR (rows past the matrix edge re-read the last row: never stored)
    const __bf16* asrc[NI];
    const __bf16* bsrc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int p = (4 * i + w) * 64 + lane, row = p >> LCPR, c = (p & (CPR - 1)) ^ sw(row);
        asrc[i] = A + (long long)min(m0 + row, a.M - 1) * a.lda + 8 * c;
        if (BKC) {
            bsrc[i] = B + (long long)min(n0 + row, a.N - 1) * a.ldb + 8 * c;
        } else {
            const int kr = p >> 4, cn = (p & 15) ^ (2 * (kr & 3));        // k-row of the tile, logical 8-column chunk
            bsrc[i] = B + (long long)kr * a.ldb + min(n0 + 8 * cn, a.N - 8);
        }
    }
    auto issue = [&](int stage, int k0) {
        __bf16* As = lds + stage * 2 * TILE;
        __bf16* Bs = As + TILE;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            // wave-uniform LDS base of this instruction; the hardware adds lane * 16 bytes
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + k0),
                                             (__attribute__((address_space(3))) void*)(As + (4 * i + w) * 512), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[i] + (BKC ? (long long)k0 : (long long)k0 * a.ldb)),
                                             (__attribute__((address_space(3))) void*)(Bs + (4 * i + w) * 512), 16, 0, 0);
        }
    };
    constexpr int NJ = GN / 32;                            // 16-column MFMA tiles per wave
    const int wm = (w >> 1) * 64, wn = (w & 1) * (GN / 2);
    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = a.K / KD;
    DLWP_STAMP(10);
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt == 1) DLWP_STAMP(11);
        if (kt + 1 < nk) {
            issue((kt + 1) & 1, (kt + 1) * KD);
            // this step's DMAs have landed; the next step's stay in flight
            if (KD == 64) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                              // ... for every wave of the workgroup
        asm volatile("" ::: "memory");
        const __bf16* As = lds + (kt & 1) * 2 * TILE;
        const __bf16* Bs = As + TILE;
#pragma unroll
        for (int kk = 0; kk < KD / 32; ++kk) {
            bf16x8 af[4], bf[NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm + 16 * i + r, c = (4 * kk + g) ^ sw(row);
                af[i] = *reinterpret_cast<const bf16x8*>(As + row * KD + 8 * c);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (BKC) {
                    const int row = wn + 16 * j + r, c = (4 * kk + g) ^ sw(row);
                    bf[j] = *reinterpret_cast<const bf16x8*>(Bs + row * KD + 8 * c);
                } else {
                    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
                    const int kr = 32 * kk + 8 * g + (r >> 2), nn = wn + 16 * j + 4 * (r & 3);      // (kr & 3) == (kr + 4) & 3
                    const __bf16* p0 = Bs + kr * GT + 8 * ((nn >> 3) ^ (2 * (kr & 3))) + (nn & 4);
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * GT));
                    bf[j] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // every wave has read this stage before it is refilled
    }
    DLWP_STAMP(12);
    // ---- epilogue (two halves of 64 rows through the dead operand buffers; 16-byte global accesses)
    constexpr int LDE = GT + 4, C4 = GN / 4, RPP = 8, NPASS = 64 / RPP;
    float* tile = gsm;
    const bool epi_thread = tid < RPP * C4;                // GN = 96: 192 of the 256 threads
    const int c4 = tid % C4, n = n0 + 4 * c4;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias && n < a.N && epi_thread) bv = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) lds_barrier();
        if (wm / 64 == half) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) tile[(i * 16 + 4 * g + q) * LDE + wn + j * 16 + r] = acc[i][j][q];
        }
        lds_barrier();
        if (epi_thread) epilogue_rows<NPASS, RPP, C4, LDE, GLDS_EPI_G>(a, tile, tid, m0 + 64 * half, n, bv);
    }
    DLWP_STAMP(13);
}

// ---- weight gradients ("TN": gW = g^T x, both operands [k = tokens][row]): both tiles are [KD k][128] images with the NN
// kernel's B-tile swizzle and both fragments come through the transposing read.  K is split over blockIdx.z (fp32 atomic adds
// into the zeroed / accumulating output, as gemm_kernel's split-K); the token count need not be a multiple of the K-step: the
// DMAs of a partial last step re-read row K - 1 and the A fragments of k >= K are zeroed.  rowsum (bias gradient) = sum over k
// of the A fragments, flushed by the n0 == 0 tiles.
template <int KD>
__global__ __launch_bounds__(256) void gemm_glds_tn_kernel(GemmDev a) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    __bf16* lds = reinterpret_cast<__bf16*>(gsm);          // [2 stages][A | B][KD][128]
    constexpr int TILE = GT * KD, NI = KD / 16;
    const int lane = lane_id(), w = wave_id(), r = lane & 15, g = lane >> 4;
    int tile_id = blockIdx.x, slice = blockIdx.z;
    if (a.xcd_splitk) {                  // few tiles x (8 n) slices: the tiles of one slice on one XCD (see gemm_body)
        const int nt = gridDim.x, id = blockIdx.x + nt * blockIdx.z, xcd = id & 7, j = id >> 3;
        tile_id = j % nt;
        slice = (j / nt) * 8 + xcd;
    } else {
        const int nt = gridDim.x, full = (nt / 8) * 8;
        if (tile_id < full) tile_id = (tile_id % 8) * (nt / 8) + tile_id / 8;
    }
    const int mt = tile_id / a.ntn, nt_ = tile_id - mt * a.ntn;
    const int m0 = mt * GT, n0 = nt_ * GT;
    const int kbeg = slice * a.kchunk, kend = min(a.K, kbeg + a.kchunk);
    const __bf16* A = reinterpret_cast<const __bf16*>(a.A);
    const __bf16* B = reinterpret_cast<const __bf16*>(a.B);
    int krow[NI], acol[NI], bcol[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int p = (4 * i + w) * 64 + lane, kr = p >> 4, c = (p & 15) ^ (2 * (kr & 3));
        krow[i] = kr;
        acol[i] = min(m0 + 8 * c, a.M - 8);
        bcol[i] = min(n0 + 8 * c, a.N - 8);
    }
    auto issue = [&](int stage, int k0) {
        __bf16* As = lds + stage * 2 * TILE;
        __bf16* Bs = As + TILE;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const long long kk = min(k0 + krow[i], a.K - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + kk * a.lda + acol[i]),
                                             (__attribute__((address_space(3))) void*)(As + (4 * i + w) * 512), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(B + kk * a.ldb + bcol[i]),
                                             (__attribute__((address_space(3))) void*)(Bs + (4 * i + w) * 512), 16, 0, 0);
        }
    };
    const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias-gradient by-product: the row sums of the A tile are the same for every column tile -- each column tile's workgroups take
    // the K-steps kt = nt_ (mod ntn), so the extra VALU work (it does not overlap the MFMAs) is spread evenly instead of making
    // the n0 == 0 workgroups the slow ones of the round
    float rsum[4] = {0.f, 0.f, 0.f, 0.f};
    const bool has_rsum = a.rowsum && !(w & 1);
    const int nk = (kend - kbeg + KD - 1) / KD;
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    auto frag = [&](const __bf16* tile, int col0, int kk) {      // rows col0 + r of the operand, k = 32 kk + 8 g .. + 7
        const int kr = 32 * kk + 8 * g + (r >> 2), cc = col0 + 4 * (r & 3);
        const __bf16* p0 = tile + kr * GT + 8 * ((cc >> 3) ^ (2 * (kr & 3))) + (cc & 4);
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * GT));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    if (nk > 0) issue(0, kbeg);
    for (int kt = 0; kt < nk; ++kt) {
        const int k0 = kbeg + kt * KD;
        if (kt + 1 < nk) {
            issue((kt + 1) & 1, k0 + KD);
            if (KD == 64) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const __bf16* As = lds + (kt & 1) * 2 * TILE;
        const __bf16* Bs = As + TILE;
        const bool tail = k0 + KD > kend;
#pragma unroll
        for (int kk = 0; kk < KD / 32; ++kk) {
            bf16x8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = frag(As, wm + 16 * i, kk);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = frag(Bs, wn + 16 * j, kk);
            if (tail) {
                const int kl = kend - (k0 + 32 * kk + 8 * g);        // this lane's fragment elements e >= kl lie past the end
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (e >= kl) af[i][e] = (__bf16)0.f;
            }
            if (has_rsum && kt % a.ntn == nt_) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) rsum[i] += (float)af[i][e];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (has_rsum) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = rsum[i];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int m = m0 + wm + 16 * i + r;
            if (g == 0 && m < a.M) atomic_add_f32(&a.rowsum[m], v);
        }
    }
    const bool atomic = a.splits > 1;
    float* out = a.slab ? a.slab + (long long)slice * a.M * a.N : a.C;
    const int ldo = a.slab ? a.N : a.ldc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + wm + i * 16 + 4 * g + q, n = n0 + wn + j * 16 + r;
                if (m < a.M && n < a.N) {
                    float* dst = out + (long long)m * ldo + n;
                    if (a.slab) { *dst = acc[i][j][q]; continue; }
                    if (atomic) atomic_add_f32(dst, acc[i][j][q]);          // C zeroed by the caller (or accumulating)
                    else *dst = a.accumulate ? *dst + acc[i][j][q] : acc[i][j][q];
                }
            }
}

// C (+)= sum over the K slices of the slab, in slice order (bit-reproducible, unlike the float atomics it replaces: the atomic
// epilogue of the sliced kernel cost 25-30 % of its time, profiles/r03_gemm_glds_tn.txt)
__global__ __launch_bounds__(256) void gemm_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ C, int M, int N, int ldc,
                                                               int slices, int accumulate) {
    const int n4 = N / 4;
    const long long total = (long long)M * n4, plane = (long long)M * N;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int m = (int)(e / n4), n = (int)(e - (long long)m * n4) * 4;
        const float* sp = slab + (long long)m * N + n;
        f32x4 v = *reinterpret_cast<const f32x4*>(sp);
        for (int z = 1; z < slices; ++z) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(sp + z * plane);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += u[k];
        }
        float* cp = C + (long long)m * ldc + n;
        if (accumulate) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += cp[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) cp[k] = v[k];
    }
}

// Scratch for the slices of the sliced weight-gradient kernel.  One slab per DEVICE, grow-only and never freed: a hipGraph captured
// earlier keeps the pointer it was captured with, so a slab that was handed out once must stay valid for the life of the
// process (the round-3 version freed and re-allocated it when a later, larger product came along, behind a
// hipDeviceSynchronize -- a replay of an older graph then wrote into freed memory).  Growing allocates a NEW block (outside
// stream captures only; a capture that would need a larger one keeps the float-atomic epilogue) and leaves the old ones
// alone; at most log2(largest / smallest) blocks exist per device.  No synchronisation, nothing is ever released.
// As with dlwp_sumsq's partials, calls on different streams of one device at the same time would share the slab: the training
// paths issue their GEMMs on one stream.  dlwp_wgrad_segments (csrc/wgrad_multi.hip) takes its slab from the caller instead.
struct TnSlab { float* p = nullptr; size_t bytes = 0; };
static TnSlab g_tn_slabs[64];                              // indexed by device ordinal
static std::mutex g_tn_slab_mutex;
static float* tn_slab_for(hipStream_t s, size_t bytes) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(g_tn_slab_mutex);
    TnSlab& slab = g_tn_slabs[dev];
    if (slab.p && slab.bytes >= bytes) return slab.p;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
    const size_t want = std::max(bytes, 2 * slab.bytes);   // geometric growth bounds the number of retired blocks
    float* p = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&p), want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    slab.p = p;                                            // the previous block (if any) stays allocated: graphs may hold it
    slab.bytes = want;
    return p;
}

// weight-gradient products the TN kernel takes: both operands bf16 arrays [k][row], fp32 output, no epilogue, one batch, the
// caller already prepared for split-K (zeroed or accumulating output)
static bool gemm_glds_tn_applies(const GemmDev& a) {
    const bool off = dlwp_tune_on("GEMM_NOGLDS") || dlwp_tune_on("GEMM_NOGLDS_TN");
    if (off || !g_gemm_bf16 || (a.dt & (DT_A | DT_B | DT_C | DT_R)) != (DT_A | DT_B)) return false;
    if (a.M % 8 || a.N % 8 || a.lda % 8 || a.ldb % 8 || (uintptr_t)a.A % 16 || (uintptr_t)a.B % 16) return false;
    if (a.nbatch != 1 || a.atomic_out || a.bias || a.residual || a.preact || a.act || a.act_b || a.bias_row) return false;
    return a.M >= GT && a.N >= GT && a.K >= 1024 && a.splits > 1;
}
static int gemm_glds_tn_launch(const GemmDev& a_in, hipStream_t s) {
    GemmDev a = a_in;
    a.ntn = ceil_div(a.N, GT);
    a.ntm = ceil_div(a.M, GT);
    // K-step depth and slices from the sweep in profiles/r03_gemm_glds_tn.txt: the grid should fill the resident slots once (a second,
    // partial round costs as much as a full one): 64 deep = 64 KB of LDS, two workgroups per CU; 32 deep = 32 KB and 160 VGPRs, three
    // per CU, which pays once there are enough output tiles (FourCastNet's 3072 x 768: 120 us against 133)
    const int kd_env = dlwp_tune("GEMM_GLDS_TN_KD");
    const bool shallow = kd_env != DLWP_TUNE_UNSET ? kd_env == 32 : a.ntn * a.ntm >= 96;
    const int kd = shallow ? 32 : 64;
    const int wg_env = dlwp_tune("GEMM_GLDS_TN_WGS");
    const int slots = wg_env != DLWP_TUNE_UNSET ? wg_env : (shallow ? 768 : 448);
    // at least eight K-steps per slice: shorter slices only add partial tiles to combine (8192 tokens x 512 x 256, graph timing,
    // tools/bench_gemm_graph.py: 32 slices 22.6 us, 16 slices 17.4 us)
    int splits = std::max(1, std::min(slots / (a.ntn * a.ntm), a.K / (8 * kd)));
    a.kchunk = ceil_div(ceil_div(a.K, splits), kd) * kd;
    a.xcd_splitk = 0;
    if (splits >= 8 && a.ntn * a.ntm < 64 && (dlwp_tune_or("GEMM_XCD_SPLITK", 1) & 2) != 0) {
        // few tiles, many slices: a slice count that is a multiple of 8 keeps the tiles of a slice on one XCD (round 5)
        for (int s8 = round_up(splits, 8); s8 >= 8 && s8 > splits / 2; s8 -= 8) {
            const int kc = ceil_div(ceil_div(a.K, s8), kd) * kd;
            if (ceil_div(a.K, kc) % 8 == 0) { a.kchunk = kc; a.xcd_splitk = 1; break; }
        }
    }
    a.splits = std::max(2, ceil_div(a.K, a.kchunk));             // > 1: the atomic epilogue (the caller zeroed C for its own split)
    const size_t lds = (size_t)2 * 2 * GT * kd * 2;
    const dim3 grid(a.ntn * a.ntm, 1, ceil_div(a.K, a.kchunk));
    const bool no_slab = dlwp_tune_on("GEMM_TN_ATOMIC");
    // few output tiles cut into many slices (SFNO's 512 x 256: 8 tiles x 32) read more slab in the reduction than the atomics cost
    const int slab_min_env = dlwp_tune("GEMM_TN_SLAB_MIN");
    const int slab_min = slab_min_env != DLWP_TUNE_UNSET ? slab_min_env : 24;
    a.slab = (no_slab || a.ntn * a.ntm < slab_min || a.N % 4 || a.ldc % 4 || (uintptr_t)a.C % 16) ? nullptr
                                                                         : tn_slab_for(s, sizeof(float) * grid.z * (size_t)a.M * a.N);
    int rc;
    {
    dlwp_prof_scope prof(s, gemm_prof_flops(a), gemm_prof_bytes(a), "gemm_glds_tn_kernel<%d>%s", kd, gemm_prof_tag(a).s);
    if (shallow) {
        if ((rc = dlwp_ensure_lds(reinterpret_cast<const void*>(gemm_glds_tn_kernel<32>), lds, "gemm_glds_tn"))) return rc;
        hipLaunchKernelGGL(gemm_glds_tn_kernel<32>, grid, dim3(256), lds, s, a);
    } else {
        if ((rc = dlwp_ensure_lds(reinterpret_cast<const void*>(gemm_glds_tn_kernel<64>), lds, "gemm_glds_tn"))) return rc;
        hipLaunchKernelGGL(gemm_glds_tn_kernel<64>, grid, dim3(256), lds, s, a);
    }
    }
    if (a.slab) {
        // the caller zeroed C for its own split-K when it does not accumulate: adding to it is the same either way
        const long long units = (long long)a.M * (a.N / 4);
        hipLaunchKernelGGL(gemm_slab_reduce_kernel, dim3((unsigned)std::min<long long>((units + 255) / 256, 2048)), dim3(256), 0, s,
                           a.slab, a.C, a.M, a.N, a.ldc, (int)grid.z, 1);
    }
    return DLWP_OK;
}

// ---- (round 3) 256 x 256 x 64 tiles, eight waves in two groups half a phase apart (the CDNA guide's "8-phase" schedule, built
// here from its description): y = x W^T, both operands bf16 arrays with k contiguous.
// A wave (wr = w >> 2, wc = w & 3) owns rows wr * 128 .. + 128, columns wc * 64 .. + 64 of the tile = four quadrants
// (mq, nq) of 64 x 32, sixteen 16x16x32 MFMAs each per K-tile.  A K-tile's operands live in LDS as four HALF-TILES of 16 KB:
// A0 / A1 = the 128 rows every wave reads for mq = 0 / 1 (local row lr = wr * 64 + x <-> tile row wr * 128 + mq * 64 + x), B0 / B1
// likewise for nq (local column wc * 32 + x <-> tile column wc * 64 + nq * 32 + x); two stages (K-tile parity) = 128 KB.
// Per K-tile a wave runs eight half-phases, a barrier after each:
//     0 read B0 (4 x b128, kept to the end) + A0 (8)         1 MFMA (mq0, nq0)
//     2 read B1 (4); issue A0, B0 of tile t + 2               3 MFMA (mq0, nq1)
//     4 read A1 (8); issue B1 of tile t + 2                   5 MFMA (mq1, nq1)
//     6 vmcnt(6): tile t + 1 has landed; issue A1 of t + 2    7 MFMA (mq1, nq0)
// Waves with wr = 1 take one extra barrier first, so they run one half-phase behind: while one group holds the matrix pipes the
// other reads fragments and issues DMAs.  Hazards (g = group, global half-phase = local + g):
//   WAR  a half-tile of stage t & 1 is refilled for tile t + 2 two half-phases after its local read slot, i.e. after the later
//        group's read of it has completed (every read slot ends with lgkmcnt(0) before its barrier);
//   RAW  tile t + 1's DMAs were issued during tile t - 1 (local slots 2, 4, 6); at slot 6 of tile t each wave waits until all but
//        the six DMAs it issued during tile t have landed -- group 1 does so one global half-phase later, still before the
//        barrier that precedes group 0's first read of tile t + 1 (global 8 t + 8).
// A DMA instruction moves 64 consecutive 16-byte chunks = 8 rows of a half-tile image (lane-linear LDS side), chunk c of local row
// lr stored at position c ^ ((lr >> 1) & 7); each group loads half of every half-tile (two DMAs per thread).
constexpr int P8T = 256;
// the barrier as inline assembly with a memory clobber: the compiler must not move LDS reads across it (the s_barrier builtin alone
// does not order memory accesses for the optimiser; a read hoisted above the barrier that follows the other group's vmcnt wait
// would see a half-tile before its DMA has landed)
__device__ __forceinline__ void p8_barrier() { asm volatile("s_barrier" ::: "memory"); }
template <bool DIRECT, bool BKC = true, bool WIDE = false>      // WIDE: the register epilogue in 128-byte rows (p8_rows8) -- its own instantiation: with both
                                             // epilogue forms in one kernel the allocator spilled 120 VGPRs.  BKC = false: B is [k][n] (gx = g W): its half-tile images are [64 k][128 local columns], fragments through the
                                             // transposing read (as in gemm_p8_tn_kernel).  DIRECT: accumulate C^T fragments (operands swapped in the MFMA) so that a lane holds four consecutive COLUMNS of a
                            // row and the epilogue stores straight from the registers (no LDS staging, no barriers)
__global__ __launch_bounds__(512) void gemm_p8_kernel(GemmDev a) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    __bf16* lds = reinterpret_cast<__bf16*>(gsm);          // [2 stages][A0 | A1 | B0 | B1][128][64]
    constexpr int HT = 128 * 64;                            // bf16 elements of a half-tile image (16 KB)
    const int lane = lane_id(), w = threadIdx.x >> 6, r = lane & 15, g = lane >> 4, tid = threadIdx.x;
    const int wr = w >> 2, wc = w & 3;
    const __bf16* A = reinterpret_cast<const __bf16*>(a.A);
    const __bf16* B = reinterpret_cast<const __bf16*>(a.B);
    // persistent: workgroup b takes tiles b, b + gridDim.x, ... (the global stores of a tile's epilogue drain while the next tile's
    // DMAs and MFMAs run: with one workgroup per CU nothing else would cover them)
    const int ntiles = a.ntm * a.ntn;
    for (int raw = blockIdx.x; raw < ntiles; raw += gridDim.x) {
    int tile_id = raw;
    {
        const int full = (ntiles / 8) * 8;                  // XCD-aware order, as in gemm_kernel
        if (tile_id < full) tile_id = (tile_id % 8) * (ntiles / 8) + tile_id / 8;
    }
    constexpr int GM = 4;
    const int grp = tile_id / (GM * a.ntn), within = tile_id - grp * GM * a.ntn;
    const int rows_in = min(GM, a.ntm - grp * GM);
    const int nt_ = within / rows_in, mt = grp * GM + (within - nt_ * rows_in);
    const int m0 = mt * P8T, n0 = nt_ * P8T;
    // DMA sources: half-tile chunk p = i * 512 + tid (i = 0, 1): local row p >> 3, logical chunk (p & 7) ^ swizzle
    const __bf16* asrc[2][2];        // [half][i]
    const __bf16* bsrc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = i * 512 + tid, lr = p >> 3, c = (p & 7) ^ ((lr >> 1) & 7);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int trow = (lr >> 6) * 128 + h * 64 + (lr & 63);            // A: tile row of local row lr in half h
            const int tcol = (lr >> 5) * 64 + h * 32 + (lr & 31);            // B: tile column of local column lr in half h
            asrc[h][i] = A + (long long)min(m0 + trow, a.M - 1) * a.lda + 8 * c;
            if (BKC) {
                bsrc[h][i] = B + (long long)min(n0 + tcol, a.N - 1) * a.ldb + 8 * c;
            } else {
                // chunk p of a [64 k][128] image: k-row p >> 4, position p & 15 holds logical chunk (p & 15) ^ 2 (kr & 3)
                const int kr = p >> 4, lc = 8 * ((p & 15) ^ (2 * (kr & 3)));
                bsrc[h][i] = B + (long long)kr * a.ldb + min(n0 + (lc >> 5) * 64 + h * 32 + (lc & 31), a.N - 8);
            }
        }
    }
    // which: 0 A0, 1 A1, 2 B0, 3 B1
    auto issue = [&](int stage, int which, int k0) {
        __bf16* dst = lds + (stage * 4 + which) * HT;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const __bf16* src = which < 2 ? asrc[which & 1][i] + k0 : bsrc[which & 1][i] + (BKC ? (long long)k0 : (long long)k0 * a.ldb);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + (i * 512 + w * 64) * 8), 16, 0, 0);
        }
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = a.K / 64;
    if (raw == blockIdx.x) DLWP_STAMP(14);
    // prologue: tiles 0 and 1 in full
#pragma unroll
    for (int t = 0; t < 2; ++t)
        if (t < nk) {
#pragma unroll
            for (int which = 0; which < 4; ++which) issue(t, which, t * 64);
        }
    if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    p8_barrier();
    if (wr == 1) p8_barrier();             // group 1 runs one half-phase behind
    if (raw == blockIdx.x) DLWP_STAMP(15);
    bf16x8 af[8], b0[4], b1[4];
    auto read_a = [&](int stage, int mq) {
        const __bf16* img = lds + (stage * 4 + mq) * HT;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int lr = wr * 64 + 16 * i + r, c = (4 * kk + g) ^ ((lr >> 1) & 7);
                af[2 * i + kk] = *reinterpret_cast<const bf16x8*>(img + lr * 64 + 8 * c);
            }
    };
    auto read_b = [&](int stage, int nq, bf16x8 (&bf)[4]) {
        const __bf16* img = lds + (stage * 4 + 2 + nq) * HT;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (BKC) {
                    const int lr = wc * 32 + 16 * j + r, c = (4 * kk + g) ^ ((lr >> 1) & 7);
                    bf[2 * j + kk] = *reinterpret_cast<const bf16x8*>(img + lr * 64 + 8 * c);
                } else {
                    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
                    const int kr = 32 * kk + 8 * g + (r >> 2), c2 = wc * 32 + 16 * j + 4 * (r & 3);
                    const __bf16* p0 = img + kr * 128 + 8 * ((c2 >> 3) ^ (2 * (kr & 3))) + (c2 & 4);
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * 128));
                    bf[2 * j + kk] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
            }
    };
    auto mma = [&](int mq, int nq, const bf16x8 (&bf)[4]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[4 * mq + i][2 * nq + j] = DIRECT ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[2 * j + kk], af[2 * i + kk], acc[4 * mq + i][2 * nq + j], 0, 0, 0)
                                                         : __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2 * i + kk], bf[2 * j + kk], acc[4 * mq + i][2 * nq + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto end_read = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        p8_barrier();
    };
    for (int t = 0; t < nk; ++t) {
        const int st = t & 1, k2 = (t + 2) * 64;
        const bool more = t + 2 < nk;
        // 0
        read_b(st, 0, b0);
        read_a(st, 0);
        end_read();
        // 1
        mma(0, 0, b0);
        p8_barrier();
        // 2
        read_b(st, 1, b1);
        if (more) { issue(st, 0, k2); issue(st, 2, k2); }
        end_read();
        // 3
        mma(0, 1, b1);
        p8_barrier();
        // 4
        read_a(st, 1);
        if (more) issue(st, 3, k2);
        end_read();
        // 5
        mma(1, 1, b1);
        p8_barrier();
        // 6: tile t + 1 must have landed (its DMAs are older than the six of this tile)
        if (t + 1 < nk) {
            if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (more) issue(st, 1, k2);
        p8_barrier();
        // 7
        mma(1, 0, b0);
        p8_barrier();
    }
    if (wr == 0) p8_barrier();             // group 0 catches up with group 1's last half-phase
    if (raw == blockIdx.x) DLWP_STAMP(16);
    if constexpr (DIRECT && WIDE) {
        // ---- epilogue straight from the registers in 128-byte rows (p8_rows8): eight columns per lane, two pieces per 16-row block
        const int nw = n0 + wc * 64 + 32 * (r >> 3) + 16 * (g & 1) + 4 * (g & ~1);
        f32x4 bv8[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if (a.bias && !a.bias_row && nw < a.N) {
            bv8[0] = *reinterpret_cast<const f32x4*>(a.bias + nw);
            bv8[1] = *reinterpret_cast<const f32x4*>(a.bias + nw + 4);
        }
        constexpr int NI = P8_WIDE_NI;                 // 16-row blocks per round: 2 NI pieces of eight columns in flight per lane
#pragma unroll
        for (int I0 = 0; I0 < 8; I0 += NI) {
            f32x4 v[2 * NI][2];
            int m[2 * NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int I = I0 + i;
                m[2 * i] = m0 + wr * 128 + (I >> 2) * 64 + 16 * (I & 3) + (r & 7);
                m[2 * i + 1] = m[2 * i] + 8;
            }
            epilogue_group8<2 * NI>(a, v, m, nw, bv8, [&](f32x4 (&vv)[2 * NI][2]) {
#pragma unroll
                for (int i = 0; i < NI; ++i) p8_rows8(acc[I0 + i], vv[2 * i], vv[2 * i + 1]);
            });
        }
    } else if constexpr (DIRECT) {
        // ---- epilogue straight from the registers: acc[I][J][q] = C[row wr * 128 + (I >> 2) * 64 + 16 (I & 3) + r][column wc * 64 + (J >> 1) * 32 + (J & 1) * 16 + 4 g + q]
#pragma unroll
        for (int J = 0; J < 4; ++J) {
            const int n = n0 + wc * 64 + (J >> 1) * 32 + (J & 1) * 16 + 4 * g;
            f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.bias && !a.bias_row && n < a.N) bv = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
            for (int I0 = 0; I0 < 8; I0 += 4) {
                f32x4 v[4];
                int m[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int I = I0 + i;
                    v[i] = acc[I][J];
                    m[i] = m0 + wr * 128 + (I >> 2) * 64 + 16 * (I & 3) + r;
                }
                epilogue_group<4>(a, v, m, n, bv);
            }
        }
    } else {
    // ---- epilogue: four passes of 64 tile rows through an fp32 LDS tile [64][260]; pass = 2 wr + mq
    constexpr int LDE = P8T + 4, C4 = P8T / 4, RPP = 512 / C4, NPASS = 64 / RPP;
    float* tile = gsm;
    const int c4 = tid % C4, n = n0 + 4 * c4;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias && n < a.N) bv = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        lds_barrier();
        if (wr == (pass >> 1)) {
            const int mq = pass & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        // acc[4 mq + i][j']: rows mq * 64 + 16 i + 4 g + q of the wave's 128, columns wc * 64 + (j' >> 1) * 32 + (j' & 1) * 16 + r
                        tile[(16 * i + 4 * g + q) * LDE + wc * 64 + (j >> 1) * 32 + (j & 1) * 16 + r] = acc[4 * mq + i][j][q];
                    }
        }
        lds_barrier();
        epilogue_rows<NPASS, RPP, C4, LDE, 4>(a, tile, tid, m0 + 64 * pass, n, bv);
    }
    }
        lds_barrier();          // the staging tile is read out before the next tile's DMAs land on it
        if (raw == blockIdx.x) DLWP_STAMP(17);
    }
}

// shapes the LDS-DMA kernel takes: both operands bf16 arrays with k contiguous and 16-byte aligned rows, K a multiple of 64, one
// plain product (no split-K / batches / row sums / row bias), the aligned epilogue, enough tiles to be worth 128 x 128
// column-tile width of gemm_glds_kernel: 96 when N is a multiple of 96 and rounds x width (whole rounds of one workgroup per CU, work per
// tile ~ its width) comes out lower than with 128-wide tiles; ties go to 128 when N is a multiple of 128.  GEMM_GLDS_N96: 0 never, 1 by
// this rule (default), 2 wherever N % 96 == 0.  GEMM_GLDS_N96_TIE_K: ties go to 96 up to this K (default 192).
static int gemm_glds_tile_n(const GemmDev& a) {
    const int mode = dlwp_tune_or("GEMM_GLDS_N96", 1);
    if (mode == 0 || a.N % 96 != 0) return GT;
    if (mode == 2) return 96;
    const long long ncu = 256, mt = ceil_div(a.M, GT);
    const long long c128 = ceil_div(mt * ceil_div(a.N, GT), ncu) * 128, c96 = ceil_div(mt * (a.N / 96), ncu) * 96;
    // ties: 96 where 128 would pad, and for short products (K <= 192: launch / epilogue bound, the finer tiles fill the last round better --
    // Swin C4 443 -> 449 samples/s with every tie at 96, AFNO 721 53.6 -> 52.3 with its K = 768 products there too)
    return c96 < c128 || (c96 == c128 && (a.N % GT != 0 || a.K <= dlwp_tune_or("GEMM_GLDS_N96_TIE_K", 192))) ? 96 : GT;
}
static bool gemm_glds_applies(const GemmDev& a, bool akc, bool bkc) {
    const bool off = dlwp_tune_on("GEMM_NOGLDS");
    if (off || !g_gemm_bf16 || !akc || (a.dt & (DT_A | DT_B)) != (DT_A | DT_B)) return false;
    if (a.K % 32 || a.lda % 8 || a.ldb % 8 || (uintptr_t)a.A % 16 || (uintptr_t)a.B % 16) return false;
    if (!bkc && a.N % 8) return false;
    if (a.splits != 1 || a.nbatch != 1 || a.atomic_out || a.rowsum || a.bias_row || a.act_b || !a.vec_epi) return false;
    const bool force = dlwp_tune_on("GEMM_GLDS_FORCE");          // measurement: skip the shape heuristic below
    if (force) return a.M >= GT && a.N >= 64;
    // at least one workgroup per CU: below that the 64 x 64 kernel's shorter prologue wins (measured, profiles/r03_gemm_bench.txt: Pangu
    // 8192 x 192 x 768 104 vs 120 TFLOP/s, 2048 x 1536 x 384 77 vs 93; round 6, in the step: 128 tiles instead of 256 costs Pangu C4 1.3 %).
    // K: rounds 3 - 5 asked for four 64-deep K-steps; with >= 256 tiles the kernel wins from K = 96 on (three 32-deep steps) -- back to back
    // 32768 x 768 x 192 36.2 -> 26.1 us, 65536 x 384 x 96 28.8 -> 21.7 us (profiles/r06_gemm_vs_vendor.txt), in the step Swin C4 390.5 -> 396.5,
    // Pangu C4 107.0 -> 107.9 samples/s (profiles/r06_gemm_glds_threshold_sweep.txt)
    const int mink = dlwp_tune_or("GEMM_GLDS_MINK", 96), mintiles = dlwp_tune_or("GEMM_GLDS_MINTILES", 256);
    return a.K >= mink && (long long)ceil_div(a.M, GT) * ceil_div(a.N, gemm_glds_tile_n(a)) >= mintiles;
}
static int gemm_glds_launch(const GemmDev& a_in, bool bkc, hipStream_t s) {
    GemmDev a = a_in;
    const int gn = gemm_glds_tile_n(a);
    a.ntn = ceil_div(a.N, gn);
    a.ntm = ceil_div(a.M, GT);
    const int kd_env = dlwp_tune("GEMM_GLDS_KD");
    // depth of a K-step (profiles/r03_gemm_glds_kd.txt): 32 deep leaves room for three or four workgroups per CU, which wins for
    // y = x W^T up to K ~ 1000 (16200 x 3072 x 768: 122 -> 110 us) and for gx = g W with at least four column tiles (16200 x 768 x 3072:
    // 122 -> 96 us, 8192^3: 841 -> 945 TFLOP/s); the deep, long products stay at 64 (8192^3 y: 1001 TFLOP/s against 828)
    const bool shallow = a.K % GK != 0 || (kd_env != DLWP_TUNE_UNSET ? kd_env == 32 : (bkc ? a.K <= 1024 : a.N >= 512));
    // 64 KB at KD = 64 (the epilogue's 64 x 132 fp32 half tile fits inside); 33 KB (that half tile) at KD = 32
    const size_t lds = shallow ? sizeof(float) * 64 * (GT + 4) : (size_t)2 * 2 * GT * GK * 2;
    int rc;
#define GLDS_GO2(BKC_, KD_, GN_)                                                                                        \
    do {                                                                                                                \
        if ((rc = dlwp_ensure_lds(reinterpret_cast<const void*>(gemm_glds_kernel<BKC_, KD_, GN_>), lds, "gemm_glds"))) return rc; \
        dlwp_prof_scope prof(s, gemm_prof_flops(a), gemm_prof_bytes(a), "gemm_glds_kernel<%s, %d, %d>%s", BKC_ ? "true" : "false", KD_, GN_,    \
                             gemm_prof_tag(a).s);                                                                        \
        hipLaunchKernelGGL((gemm_glds_kernel<BKC_, KD_, GN_>), dim3(a.ntn * a.ntm), dim3(256), lds, s, a);               \
    } while (0)
#define GLDS_GO(BKC_, KD_) do { if (gn == 96) GLDS_GO2(BKC_, KD_, 96); else GLDS_GO2(BKC_, KD_, 128); } while (0)
    if (bkc) { if (shallow) GLDS_GO(true, 32); else GLDS_GO(true, 64); }
    else     { if (shallow) GLDS_GO(false, 32); else GLDS_GO(false, 64); }
#undef GLDS_GO2
#undef GLDS_GO
    return DLWP_OK;
}


// ---- the same schedule for weight gradients ("TN": gW = g^T x, both operands [k = tokens][row] bf16 arrays): the half-tile images
// are [64 k][128 local columns] (chunk cm of k-row kr at position cm ^ 2 (kr & 3), as the 128^2 kernels' [k][n] tile), every fragment
// comes through two transposing reads.  Work items = (tile, K slice), dealt to the persistent workgroups; a slice is a whole number
// of K-tiles, the last one may end inside a K-tile (the DMAs re-read row K - 1, the A fragments past the end are zeroed); every item
// writes its 256 x 256 partial to the slab with plain stores through the staged epilogue and gemm_slab_reduce_kernel adds the slices
// in order.  Bias gradient = row sums of the A fragments (each column tile takes every ntn-th K-tile).
__global__ __launch_bounds__(512) void gemm_p8_tn_kernel(GemmDev a) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    __bf16* lds = reinterpret_cast<__bf16*>(gsm);          // [2 stages][A0 | A1 | B0 | B1][64 k][128]
    constexpr int HT = 64 * 128;
    const int lane = lane_id(), w = threadIdx.x >> 6, r = lane & 15, g = lane >> 4, tid = threadIdx.x;
    const int wr = w >> 2, wc = w & 3;
    const __bf16* A = reinterpret_cast<const __bf16*>(a.A);
    const __bf16* B = reinterpret_cast<const __bf16*>(a.B);
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int ntiles = a.ntm * a.ntn, nitems = ntiles * a.splits;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int zs = item / ntiles, tile_id = item - zs * ntiles;
    const int mt = tile_id / a.ntn, nt_ = tile_id - mt * a.ntn;
    const int m0 = mt * P8T, n0 = nt_ * P8T;
    const int kbeg = zs * a.kchunk, kend = min(a.K, kbeg + a.kchunk);
    const int nk = (kend - kbeg + 63) / 64;
    // DMA sources: chunk p = i * 512 + tid of a half-tile image: k-row p >> 4, position p & 15 holds logical chunk (p & 15) ^ 2 (kr & 3)
    int krow[2], acol[2][2], bcol[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = i * 512 + tid, kr = p >> 4, lc = 8 * ((p & 15) ^ (2 * (kr & 3)));
        krow[i] = kr;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            acol[h][i] = min(m0 + (lc >> 6) * 128 + h * 64 + (lc & 63), a.M - 8);
            bcol[h][i] = min(n0 + (lc >> 5) * 64 + h * 32 + (lc & 31), a.N - 8);
        }
    }
    auto issue = [&](int stage, int which, int k0) {        // which: 0 A0, 1 A1, 2 B0, 3 B1
        __bf16* dst = lds + (stage * 4 + which) * HT;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long kk = min(k0 + krow[i], a.K - 1);
            const __bf16* src = which < 2 ? A + kk * a.lda + acol[which & 1][i] : B + kk * a.ldb + bcol[which & 1][i];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + (i * 512 + w * 64) * 8), 16, 0, 0);
        }
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float rsum[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rsum[i] = 0.f;
    const bool has_rsum = a.rowsum && wc == 0;
#pragma unroll
    for (int t = 0; t < 2; ++t)
        if (t < nk) {
#pragma unroll
            for (int which = 0; which < 4; ++which) issue(t, which, kbeg + t * 64);
        }
    if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    p8_barrier();
    if (wr == 1) p8_barrier();
    bf16x8 af[8], b0[4], b1[4];
    auto frag = [&](const __bf16* img, int cc, int kk) {      // local columns cc .. : lane (r, g) gets column cc' = cc + r... rows k = 32 kk + 8 g ..+7
        const int kr = 32 * kk + 8 * g + (r >> 2), c2 = cc + 4 * (r & 3);
        const __bf16* p0 = img + kr * 128 + 8 * ((c2 >> 3) ^ (2 * (kr & 3))) + (c2 & 4);
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * 128));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto read_a = [&](int stage, int mq, int k0, int kt) {
        const __bf16* img = lds + (stage * 4 + mq) * HT;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) af[2 * i + kk] = frag(img, wr * 64 + 16 * i, kk);
        if (k0 + 64 > kend) {                               // the slice ends inside this K-tile: zero the A elements past the end
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int kl = kend - (k0 + 32 * kk + 8 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (e >= kl) af[2 * i + kk][e] = (__bf16)0.f;
            }
        }
        if (has_rsum && kt % a.ntn == nt_) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int e = 0; e < 8; ++e) rsum[4 * mq + i] += (float)af[2 * i + kk][e];
        }
    };
    auto read_b = [&](int stage, int nq, bf16x8 (&bf)[4]) {
        const __bf16* img = lds + (stage * 4 + 2 + nq) * HT;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) bf[2 * j + kk] = frag(img, wc * 32 + 16 * j, kk);
    };
    auto mma = [&](int mq, int nq, const bf16x8 (&bf)[4]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[4 * mq + i][2 * nq + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2 * i + kk], bf[2 * j + kk], acc[4 * mq + i][2 * nq + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto end_read = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        p8_barrier();
    };
    for (int t = 0; t < nk; ++t) {
        const int st = t & 1, k0 = kbeg + t * 64, k2 = k0 + 128;
        const bool more = t + 2 < nk;
        read_b(st, 0, b0);
        read_a(st, 0, k0, t);
        end_read();
        mma(0, 0, b0);
        p8_barrier();
        read_b(st, 1, b1);
        if (more) { issue(st, 0, k2); issue(st, 2, k2); }
        end_read();
        mma(0, 1, b1);
        p8_barrier();
        read_a(st, 1, k0, t);
        if (more) issue(st, 3, k2);
        end_read();
        mma(1, 1, b1);
        p8_barrier();
        if (t + 1 < nk) {
            if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (more) issue(st, 1, k2);
        p8_barrier();
        mma(1, 0, b0);
        p8_barrier();
    }
    if (wr == 0) p8_barrier();
    if (has_rsum) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = rsum[i];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int m = m0 + wr * 128 + (i >> 2) * 64 + 16 * (i & 3) + r;
            if (g == 0 && m < a.M) atomic_add_f32(&a.rowsum[m], v);
        }
    }
    // ---- epilogue: the partial tile to this slice's slab plane (plain stores, four passes of 64 rows through LDS)
    constexpr int LDE = P8T + 4, C4 = P8T / 4, RPP = 512 / C4, NPASS = 64 / RPP;
    float* tile = gsm;
    GemmDev e = a;
    e.C = a.slab + (long long)zs * a.M * a.N;
    e.ldc = a.N;
    e.bias = nullptr; e.residual = nullptr; e.preact = nullptr; e.act = 0; e.accumulate = 0; e.dt = 0; e.bias_row = 0;
    const int n = n0 + 4 * (tid % C4);
    const f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        lds_barrier();
        if (wr == (pass >> 1)) {
            const int mq = pass & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        tile[(16 * i + 4 * g + q) * LDE + wc * 64 + (j >> 1) * 32 + (j & 1) * 16 + r] = acc[4 * mq + i][j][q];
        }
        lds_barrier();
        epilogue_rows<NPASS, RPP, C4, LDE, 4>(e, tile, tid, m0 + 64 * pass, n, bv);
    }
    lds_barrier();
    }
}


static int gemm_p8_launch(const GemmDev& a_in, bool bkc, hipStream_t s) {
    GemmDev a = a_in;
    a.ntn = ceil_div(a.N, P8T);
    a.ntm = ceil_div(a.M, P8T);
    const size_t lds = (size_t)2 * 4 * 128 * 64 * 2;          // 128 KB: two stages of four half-tile images (the epilogue's 64 x 260 fp32 tile fits inside)
    int rc;
    const bool direct = !dlwp_tune_on("GEMM_P8_STAGED");
    {   // 128-byte-row register epilogue (round 6): every tensor the epilogue touches in whole, aligned 8-column pieces
        auto al = [&](const void* p, bool bf) { return !p || (uintptr_t)p % (bf ? 16 : 32) == 0; };
        a.wide_epi = dlwp_tune_or("GEMM_P8_WIDE", 1) != 0 && a.N % 8 == 0 && a.ldc % 8 == 0 && al(a.C, a.dt & DT_C) && al(a.preact, a.dt & DT_C) &&
                     al(a.residual, a.dt & DT_R) && (!a.bias || a.bias_row || (uintptr_t)a.bias % 16 == 0);
    }
    static const int ncu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
#define P8_GO(D_, B_, W_)                                                                                               \
    do {                                                                                                                \
        if ((rc = dlwp_ensure_lds(reinterpret_cast<const void*>(gemm_p8_kernel<D_, B_, W_>), lds, "gemm_p8"))) return rc; \
        dlwp_prof_scope prof(s, gemm_prof_flops(a), gemm_prof_bytes(a), "gemm_p8_kernel<%s, %s, %s>%s", D_ ? "true" : "false", B_ ? "true" : "false", \
                             W_ ? "true" : "false", gemm_prof_tag(a).s);                                                 \
        hipLaunchKernelGGL((gemm_p8_kernel<D_, B_, W_>), dim3(std::min(a.ntn * a.ntm, ncu)), dim3(512), lds, s, a);     \
    } while (0)
    if (direct && a.wide_epi) { if (bkc) P8_GO(true, true, true); else P8_GO(true, false, true); }
    else if (bkc) { if (direct) P8_GO(true, true, false); else P8_GO(false, true, false); }
    else          { if (direct) P8_GO(true, false, false); else P8_GO(false, false, false); }
#undef P8_GO
    return DLWP_OK;
}

int g_gemm_tile256 = 0;    // dlwp_set_gemm_tile256: 0 by shape (below), 1 wherever the kernel applies, -1 never

// weight gradients on the 256 x 256 two-group kernel.  Measured (profiles/r03_gemm_p8.txt): 4096^3 707 against 656 TFLOP/s for the
// 128 x 128 sliced kernel, 8192^3 764 / 762, FourCastNet 3072 x 768 x 16200 639 / 689, 768 x 3072 x 16200 652 / 640 -- with every fragment
// through two transposing reads the K loop does not reach the y = x W^T kernel's rate and the gain is within the noise: NOT taken by
// shape, only when forced (dlwp_set_gemm_tile256(1) / DLWP_GEMM_P8: tests, measurements).  Needs the slab.
static int gemm_p8_tn_launch(const GemmDev& a_in, hipStream_t s, bool* taken) {
    *taken = false;
    const bool off = dlwp_tune_on("GEMM_NOP8_TN");
    if (off || g_gemm_tile256 < 0 || a_in.M < P8T || a_in.N < P8T || a_in.N % 4 || a_in.ldc % 4 || (uintptr_t)a_in.C % 16) return DLWP_OK;
    GemmDev a = a_in;
    a.ntn = ceil_div(a.N, P8T);
    a.ntm = ceil_div(a.M, P8T);
    static const int ncu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    const int tiles = a.ntn * a.ntm;
    int splits = std::max(1, std::min(ncu / tiles, a.K / (16 * 64)));
    a.kchunk = ceil_div(ceil_div(a.K, splits), 64) * 64;
    splits = ceil_div(a.K, a.kchunk);
    const bool force = dlwp_tune_on("GEMM_P8");
    if (!force && g_gemm_tile256 <= 0) return DLWP_OK;
    a.splits = splits;
    a.slab = tn_slab_for(s, sizeof(float) * splits * (size_t)a.M * a.N);
    if (!a.slab) return DLWP_OK;
    const size_t lds = (size_t)2 * 4 * 128 * 64 * 2;
    int rc;
    if ((rc = dlwp_ensure_lds(reinterpret_cast<const void*>(gemm_p8_tn_kernel), lds, "gemm_p8_tn"))) return rc;
    hipLaunchKernelGGL(gemm_p8_tn_kernel, dim3(std::min(tiles * splits, ncu)), dim3(512), lds, s, a);
    const long long units = (long long)a.M * (a.N / 4);
    hipLaunchKernelGGL(gemm_slab_reduce_kernel, dim3((unsigned)std::min<long long>((units + 255) / 256, 2048)), dim3(256), 0, s,
                       a.slab, a.C, a.M, a.N, a.ldc, splits, 1);
    *taken = true;
    return DLWP_OK;
}


// the 256 x 256 kernel runs one workgroup per CU: nothing covers its prologue (two K-tiles of DMAs) and epilogue, so it needs a long K
// to pay -- 8192^3: 1351 against 1017 TFLOP/s, 4096^3 1186 / 983, 16200 x 768 x 3072: 898 / 788, but 16200 x 3072 x 768: 686 / 693 and every
// K <= 512 shape loses (profiles/r03_gemm_p8.txt): taken from K = 2048 with at least 128 tiles, or when forced
static bool gemm_p8_applies(const GemmDev& a, bool akc, bool bkc) {
    const bool env_on = dlwp_tune_on("GEMM_P8");
    if (g_gemm_tile256 < 0 || !akc || a.K % 64 || a.M < P8T || a.N < P8T) return false;
    const long long tiles = (long long)ceil_div(a.M, P8T) * ceil_div(a.N, P8T);
    const int mink_env = dlwp_tune("GEMM_P8_MINK");
    const int mink = mink_env != DLWP_TUNE_UNSET ? mink_env : 2048;
    // one workgroup per CU: the last round should be at least 70 % full.  Round 5 asked for 80 % (16200 x 768 x 3072: 192 tiles = 75 % of one
    // round, where the 128 x 128 LDS-DMA kernel was then 0.4 % of the C5 step faster); with the 128-byte-row epilogue (round 6) this kernel
    // wins there: C5 720 x 1440 step 16.25 -> 16.08 ms, 721 x 1440 / Pangu / Swin unchanged (profiles/r06_gemm_p8_dispatch_sweep.txt)
    const long long rounds = (tiles + 255) / 256;
    const int fill_pct = dlwp_tune_or("GEMM_P8_FILL", 70);
    const bool fills = tiles >= 128 && 100 * tiles >= (long long)fill_pct * rounds * 256;
    return env_on || g_gemm_tile256 > 0 || (a.K >= mink && fills);
}

// ---- a queue of independent small products (dlwp_gemm_group_begin ... dlwp_gemm_group_end): while it is open, every product
// that fits the generic 64 x 64 configuration and is small (latency-bound by itself) is parked instead of launched; _end launches
// up to three parked products as ONE grid (gemm_group_any_kernel).  Everything else launches at once, as usual -- the products
// inside a begin / end pair must not depend on each other.
struct GemmQueue { bool open = false; int n = 0; GemmGroupAny gg; };
static thread_local GemmQueue g_queue;      // per host thread: a backward pass on another autograd thread has its own (empty) queue
static int gemm_queue_flush(hipStream_t s) {
    GemmQueue& q = g_queue;
    if (q.n == 0) return DLWP_OK;
    unsigned gx = 0, gz = 0;
    for (int i = 0; i < q.n; ++i) {
        gx = std::max(gx, (unsigned)(q.gg.g[i].ntn * q.gg.g[i].ntm));
        gz = std::max(gz, (unsigned)(q.gg.g[i].nbatch * q.gg.g[i].splits));
    }
    const dim3 grid(gx, q.n, gz);
    int rc;
    if (g_gemm_bf16) {
        constexpr int KS = gemm_bf_kstep(0, 1, false, false);
        const size_t lds = sizeof(float) * 4 * ((64 * (KS + 8) > KS * (64 + 8) ? 64 * (KS + 8) : KS * (64 + 8)) / 2);
        if ((rc = dlwp_ensure_lds(reinterpret_cast<const void*>(gemm_group_any_kernel<true>), lds, "gemm_group_any"))) return rc;
        hipLaunchKernelGGL(gemm_group_any_kernel<true>, grid, dim3(256), lds, s, q.gg);
    } else {
        const size_t lds = sizeof(float) * 4 * Tile<1>::FLOATS;
        if ((rc = dlwp_ensure_lds(reinterpret_cast<const void*>(gemm_group_any_kernel<false>), lds, "gemm_group_any"))) return rc;
        hipLaunchKernelGGL(gemm_group_any_kernel<false>, grid, dim3(256), lds, s, q.gg);
    }
    q.n = 0;
    return DLWP_OK;
}
// park a product if the queue is open and it qualifies; a full queue is flushed first
static bool gemm_queue_take(const GemmDev& a_in, bool akc, bool bkc, int vec, int T, hipStream_t s, int* rc) {
    *rc = DLWP_OK;
    GemmQueue& q = g_queue;
    if (!q.open || vec != 3 || T != 1 || a_in.act_b) return false;
    // small, latency-bound products only: at most 2.2 GFLOP (the SFNO block-tail products are 2.1; Pangu's 8192 x 384 x 1152 input
    // gradient at 7.2 GFLOP belongs on its own kernel: parked, the C4 step went from 14.3 to 15.1 ms)
    const double flop = 2.0 * a_in.M * a_in.N * (double)a_in.K * a_in.nbatch;
    if (flop > 2.2e9 || a_in.K > 16384) return false;
    if (q.n == 3 && (*rc = gemm_queue_flush(s))) return true;
    GemmDev a = a_in;
    a.ntn = ceil_div(a.N, 64);
    a.ntm = ceil_div(a.M, 64);
    q.gg.g[q.n] = a;
    q.gg.layout[q.n] = (akc ? 2 : 0) | (bkc ? 1 : 0);
    {   // both operands bf16 arrays on the 16-byte path (the conditions gemm_launch uses for a single product)
        auto ok16 = [&](const float* p, int ld, bool kc, int rows, long long s1, long long s2) {
            return (uintptr_t)p % 16 == 0 && ld % 8 == 0 && (kc ? a.K % 8 == 0 && a.kchunk % 8 == 0 : rows % 8 == 0) &&
                   (a.nbatch == 1 || (s1 % 8 == 0 && s2 % 8 == 0));
        };
        q.gg.s16m[q.n] = g_gemm_bf16 ? ((((a.dt & DT_A) && ok16(a.A, a.lda, akc, a.M, a.sA1, a.sA2)) ? 1 : 0) |
                                        (((a.dt & DT_B) && ok16(a.B, a.ldb, bkc, a.N, a.sB1, a.sB2)) ? 2 : 0))
                                     : 0;
    }
    ++q.n;
    return true;
}

template <bool AKC, bool BKC>
int gemm_launch(const GemmDev& a_in, int vec, int T, hipStream_t s) {
    {
        int qrc;
        if (gemm_queue_take(a_in, AKC, BKC, vec, T, s, &qrc)) return qrc;
    }
    if (gemm_glds_applies(a_in, AKC, BKC) && gemm_p8_applies(a_in, AKC, BKC)) return gemm_p8_launch(a_in, BKC, s);
    if (gemm_glds_applies(a_in, AKC, BKC)) return gemm_glds_launch(a_in, BKC, s);
    if (!AKC && !BKC && gemm_glds_tn_applies(a_in)) {
        bool taken = false;
        if (int rc = gemm_p8_tn_launch(a_in, s, &taken)) return rc;
        if (taken) return DLWP_OK;
        return gemm_glds_tn_launch(a_in, s);
    }
    const int edge = 64 * T;
    GemmDev a = a_in;
    a.ntn = ceil_div(a.N, edge);
    a.ntm = ceil_div(a.M, edge);
    const dim3 grid(a.ntn * a.ntm, 1, a.nbatch * a.splits);
    if (T == 2 && vec != 3) vec = 0;      // the 128-wide tile exists for fully aligned operands only
    // bf16 arrays in memory (dlwp_gemm_mixed): whole groups of 8 along the contiguous dimension and 16-byte aligned -> the
    // 16-byte path; otherwise TileIO::load widens them element-wise (correct, slower)
    if (g_gemm_bf16 && vec == 3 && (a.dt & (DT_A | DT_B)) && !a.act_b) {
        auto ok16 = [&](const float* p, int ld, bool kc, int rows, long long s1, long long s2) {
            return (uintptr_t)p % 16 == 0 && ld % 8 == 0 && (kc ? a.K % 8 == 0 && a.kchunk % 8 == 0 : rows % 8 == 0) &&
                   (a.nbatch == 1 || (s1 % 8 == 0 && s2 % 8 == 0));
        };
        const int m16 = (((a.dt & DT_A) && ok16(a.A, a.lda, AKC, a.M, a.sA1, a.sA2)) ? 1 : 0) |
                        (((a.dt & DT_B) && ok16(a.B, a.ldb, BKC, a.N, a.sB1, a.sB2)) ? 2 : 0);
#define GEMM_S16(TT)                                                                       \
        switch (m16) {                                                                     \
            case 1: return gemm_launch_t<AKC, BKC, 3, TT, true, 1>(a, grid, s);           \
            case 2: return gemm_launch_t<AKC, BKC, 3, TT, true, 2>(a, grid, s);           \
            case 3: return gemm_launch_t<AKC, BKC, 3, TT, true, 3>(a, grid, s);           \
            default: break;                                                                \
        }
        if (T == 2) { GEMM_S16(2) } else { GEMM_S16(1) }
#undef GEMM_S16
    }
#define GEMM_VEC_SWITCH(TT, BFV)                                                          \
    switch (vec) {                                                                         \
        case 3: return gemm_launch_t<AKC, BKC, 3, TT, BFV>(a, grid, s);                   \
        case 2: if (TT == 1) return gemm_launch_t<AKC, BKC, 2, 1, BFV>(a, grid, s);       \
                return gemm_launch_t<AKC, BKC, 0, TT, BFV>(a, grid, s);                   \
        case 1: if (TT == 1) return gemm_launch_t<AKC, BKC, 1, 1, BFV>(a, grid, s);       \
                return gemm_launch_t<AKC, BKC, 0, TT, BFV>(a, grid, s);                   \
        default: return gemm_launch_t<AKC, BKC, 0, TT, BFV>(a, grid, s);                  \
    }
    if (g_gemm_bf16) {
        if (T == 2) { GEMM_VEC_SWITCH(2, true) }
        GEMM_VEC_SWITCH(1, true)
    }
    if (T == 2) { GEMM_VEC_SWITCH(2, false) }
    GEMM_VEC_SWITCH(1, false)
#undef GEMM_VEC_SWITCH
}

}  // namespace

// Tile edge selection.  The 128 x 128 instantiation (T = 2) needs ~200 VGPRs and 74 KB of LDS, i.e. two waves per SIMD,
// and measured 5-7 % SLOWER than 64 x 64 (T = 1: ~104 VGPRs, four waves per SIMD hide the LDS fragment reads behind the
// other waves' MFMAs) on the SFNO (M = 512-32768, N = K = 256-512) and FourCastNet-scale (16200 x 3072 x 768) products,
// so it is only used when DLWP_GEMM_TILE=128 asks for it (kept for re-measurement with bf16 operands).
// With both operands stored as bf16 (16-byte loads, 4 staging registers per operand) the picture changes for deep products:
// profiles/r02_gemm_bench.txt, 16200 x 3072 x 768: 395 -> 479 TFLOP/s, 16200 x 768 x 3072 (gx): 438 -> 656; at K = 256-512
// (SFNO) the 128 tile still loses (264 -> 176), so it is taken from K = 768 on.
// can_split: no epilogue, so a long K may be cut into slices (weight gradients: few output tiles, K = tokens) -- the slices
// fill the chip where the tiles alone would not.
static int gemm_tile_for(int M, int N, long long nbatch, int K = 0, int dt = 0, bool can_split = false) {
    const int tile_env = dlwp_tune("GEMM_TILE"), forced = tile_env == DLWP_TUNE_UNSET ? 0 : (tile_env == 128 ? 2 : 1);
    if (M < 128 || N < dlwp_tune_or("GEMM_TILE128_MINN", 128)) return 1;      // (knob: 96-wide outputs on the 128 tile, measurement)
    const long long tiles = (long long)ceil_div(M, 128) * ceil_div(N, 128) * nbatch;
    const bool fills = tiles * (can_split ? std::max(1, K / (4 * BK)) : 1) >= 224;
    if (forced) return forced == 2 && fills ? 2 : 1;
    return g_gemm_bf16 && (dt & 3) == 3 && K >= 768 && fills ? 2 : 1;
}

static int gemm_dispatch(GemmDev& a, int transA, int transB, int T, void* stream) {
    // 16-byte loads need every row start and every group of four along the contiguous dimension to be aligned and whole;
    // decided per operand (a weight matrix with an odd row length must not force scalar loads on the activations)
    bool vecA = ((uintptr_t)a.A % ((a.dt & DT_A) ? 8 : 16) == 0) && a.lda % 4 == 0 && (transA ? a.M % 4 == 0 : a.K % 4 == 0);
    bool vecB = ((uintptr_t)a.B % ((a.dt & DT_B) ? 8 : 16) == 0) && a.ldb % 4 == 0 && (transB ? a.K % 4 == 0 : a.N % 4 == 0);
    if (a.nbatch > 1) {
        vecA = vecA && a.sA1 % 4 == 0 && a.sA2 % 4 == 0;
        vecB = vecB && a.sB1 % 4 == 0 && a.sB2 % 4 == 0;
    }
    const int vec = (vecA ? 1 : 0) | (vecB ? 2 : 0);
    a.vec_epi = a.splits == 1 && !a.atomic_out && a.N % 4 == 0 && a.ldc % 4 == 0 && (uintptr_t)a.C % 16 == 0 &&
                (!a.bias || a.bias_row || (uintptr_t)a.bias % 16 == 0) && (!a.residual || (uintptr_t)a.residual % 16 == 0) &&
                (!a.preact || (uintptr_t)a.preact % 16 == 0);
    if (a.nbatch > 1)
        a.vec_epi = a.vec_epi && a.sC1 % 4 == 0 && a.sC2 % 4 == 0 && a.sR1 % 4 == 0 && a.sR2 % 4 == 0 &&
                    (a.bias_row || (a.sBi1 % 4 == 0 && a.sBi2 % 4 == 0));
    const hipStream_t s = (hipStream_t)stream;
    int rc;
    const bool trace = dlwp_tune_on("GEMM_TRACE");      // shape census of an eager step: sort | uniq -c
    if (trace)
        fprintf(stderr, "gemm %c%c M=%d N=%d K=%d nb=%lld splits=%d dt=%d vec=%d epi=%d bias=%d res=%d act=%d rowsum=%d\n", transA ? 'T' : 'N',
                transB ? 'T' : 'N', a.M, a.N, a.K, (long long)a.nbatch, a.splits, a.dt, vec, (int)a.vec_epi, a.bias != nullptr,
                a.residual != nullptr, a.act, a.rowsum != nullptr);
    // A is k-contiguous when not transposed ([M][K]); B is k-contiguous when transposed ([N][K])
    if (!transA && transB) rc = gemm_launch<true, true>(a, vec, T, s);
    else if (!transA && !transB) rc = gemm_launch<true, false>(a, vec, T, s);
    else if (transA && transB) rc = gemm_launch<false, true>(a, vec, T, s);
    else rc = gemm_launch<false, false>(a, vec, T, s);
    if (rc) return rc;
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

// internal entry for the channels-first (wide FNO) path: everything dlwp_gemm_batched does plus row bias, GELU on the B
// operand, a bias-gradient row sum and batches that accumulate into one shared output (dlwpmi_internal.h)
int dlwp_gemm_run(const dlwp_gemm_args& g, hipStream_t stream) {
    DLWP_REQUIRE(g.A && g.B && g.C && g.M > 0 && g.N > 0 && g.K > 0 && g.nb >= 1, DLWP_E_INVALID, "gemm_run: bad argument");
    const bool epilogue = g.bias || g.act || g.preact || g.residual;
    const bool shared_out = g.nb > 1 && g.sC == 0;
    DLWP_REQUIRE(!(shared_out && epilogue), DLWP_E_INVALID, "gemm_run: batches sharing one output take no epilogue");
    const long long tiles = (long long)ceil_div(g.N, 64) * ceil_div(g.M, 64) * g.nb;
    int splits = 1;
    if (!epilogue && tiles < 256 && g.K >= 8 * BK) splits = (int)std::min<long long>(ceil_div(512, (int)tiles), g.K / (4 * BK));
    int kchunk = ceil_div(ceil_div(g.K, splits), BK) * BK;
    splits = ceil_div(g.K, kchunk);
    DLWP_REQUIRE((long long)g.nb * splits <= 65535, DLWP_E_UNSUPPORTED, "gemm_run: more than 65535 batches x splits");
    const bool atomic = shared_out || splits > 1;
    if (atomic && !g.accumulate) {
        for (int z = 0; z < (shared_out ? 1 : g.nb); ++z)
            if (int zrc = dlwp_zero_2d_f32(g.C + z * g.sC, g.ldc, g.M, g.N, stream)) return zrc;
    }
    GemmDev a{g.A, g.B, g.bias, g.residual, g.C, g.preact, g.rowsum, g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.act,
              g.accumulate, kchunk, splits, g.nb, 1, g.res_before_act, 0, 0, 0, g.sA, 0, g.sB, 0, g.sC, 0, g.sR, 0, 0, 0, 0.f,
              g.bias_row, g.act_b, shared_out ? 1 : 0};
    return gemm_dispatch(a, g.transA, g.transB, 1, stream);
}

extern "C" int dlwp_set_gemm_precision(int mode) {
    DLWP_REQUIRE(mode == 0 || mode == 1, DLWP_E_INVALID, "set_gemm_precision: 0 = fp32 operands, 1 = bf16 operands (fp32 accumulate)");
    g_gemm_bf16 = mode;
    return DLWP_OK;
}

extern "C" int dlwp_get_gemm_precision(void) { return g_gemm_bf16; }
extern "C" int dlwp_gemm_group_begin(void) {
    DLWP_REQUIRE(!g_queue.open, DLWP_E_INVALID, "gemm_group_begin: a group is already open");
    const bool off = dlwp_tune_on("GEMM_NOGROUP");
    g_queue.open = !off;
    g_queue.n = 0;
    return DLWP_OK;
}
extern "C" int dlwp_gemm_group_end(void* stream) {
    g_queue.open = false;
    const int rc = gemm_queue_flush((hipStream_t)stream);
    if (rc) return rc;
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}
extern "C" int dlwp_set_gemm_tile256(int mode) {
    DLWP_REQUIRE(mode >= -1 && mode <= 1, DLWP_E_INVALID, "set_gemm_tile256: mode must be -1 (never), 0 (by shape) or 1 (wherever it applies)");
    g_gemm_tile256 = mode;
    return DLWP_OK;
}

static int dtypes_ok(int dt, int accumulate, const char* who) {
    DLWP_REQUIRE(dt >= 0 && dt < 16, DLWP_E_INVALID, "%s: dtypes is a mask of 1 (A) | 2 (B) | 4 (C, preact) | 8 (residual)", who);
    DLWP_REQUIRE(!((dt & DT_C) && accumulate), DLWP_E_INVALID, "%s: accumulation needs an fp32 output", who);
    return DLWP_OK;
}

static int gemm_impl(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                     int transA, int transB, const float* bias, int act, float* preact, const float* residual,
                     int accumulate, float* rowsum, int dt, void* stream, const float* row_scale = nullptr, int scale_rows = 1) {
    DLWP_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, DLWP_E_INVALID, "gemm: NULL argument or empty shape");
    DLWP_REQUIRE(act == 0 || act == 1 || (act == ACT_GELU_STORE_D && preact), DLWP_E_INVALID,
                 "gemm: act must be 0 (none), 1 (gelu) or 7 (gelu with GELU' stored to a non-NULL preact)");
    if (int drc = dtypes_ok(dt, accumulate, "gemm")) return drc;
    const bool epilogue = bias || act || preact || residual || row_scale || (dt & DT_C);      // a bf16 output takes no split-K atomics
    const int T = gemm_tile_for(M, N, 1, K, dt, !epilogue);
    const int tiles = ceil_div(N, 64 * T) * ceil_div(M, 64 * T);
    int splits = 1;
    if (!epilogue && tiles < 256 && K >= 8 * BK) splits = std::min(ceil_div(512, tiles), K / (4 * BK));
    int kchunk = ceil_div(ceil_div(K, splits), BK) * BK;
    splits = ceil_div(K, kchunk);
    // a slice count that is a multiple of 8 lets the kernel keep the tiles of a K slice on one XCD (round 5): the nearest such
    // count at or above the heuristic's that survives the K-step rounding
    bool xcd_splitk = false;
    if (splits >= 8 && (dlwp_tune_or("GEMM_XCD_SPLITK", 1) & 1) != 0) {
        for (int s8 = round_up(splits, 8); s8 >= 8 && s8 > splits / 2; s8 -= 8) {
            const int kc = ceil_div(ceil_div(K, s8), BK) * BK, sp = ceil_div(K, kc);
            if (sp % 8 == 0) { kchunk = kc; splits = sp; xcd_splitk = true; break; }
        }
    }
    if (splits > 1 && !accumulate) {
        const int zrc = dlwp_zero_2d_f32(C, ldc, M, N, stream);
        if (zrc) return zrc;
    }
    GemmDev a{A, B, bias, residual, C, preact, rowsum, M, N, K, lda, ldb, ldc, act, accumulate, kchunk, splits,
              1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.f, 0, 0, 0};
    a.dt = dt;
    a.xcd_splitk = xcd_splitk ? 1 : 0;
    a.row_scale = row_scale;
    a.scale_rows = scale_rows;
    return gemm_dispatch(a, transA, transB, T, stream);
}

// Several weight-gradient products gW_i (+)= g_i^T x_i (K = tokens) with their bias gradients in ONE launch (gemm_group_kernel):
// each is prepared exactly as dlwp_gemm_mixed(transA = 1) would (split-K over the tokens, zero fill when not accumulating,
// row sums as the bias gradient) on the 64 x 64 register-staged kernel, which reads fp32 and bf16 operands alike.  Falls back to
// one launch per product when a product does not fit the grouped form (unaligned operands, more than three, an empty one).
extern "C" int dlwp_weight_grad_group(const dlwp_wgrad_desc* d, int n, void* stream) {
    DLWP_REQUIRE(d && n >= 1, DLWP_E_INVALID, "weight_grad_group: no products");
    for (int i = 0; i < n; ++i)
        DLWP_REQUIRE(d[i].g && d[i].x && d[i].gw && d[i].T > 0 && d[i].N > 0 && d[i].K > 0, DLWP_E_INVALID,
                     "weight_grad_group: product %d has a NULL operand or an empty shape", i);
    auto single = [&](const dlwp_wgrad_desc& q) {
        const int dt = (q.g_bf16 ? DT_A : 0) | (q.x_bf16 ? DT_B : 0);
        return gemm_impl((const float*)q.g, (const float*)q.x, q.gw, q.N, q.K, q.T, q.N, q.K, q.K, 1, 0, nullptr, 0, nullptr, nullptr,
                         q.accumulate, q.gb, dt, stream);
    };
    const bool off = dlwp_tune_on("GEMM_NOGROUP");
    // worth it while the products are latency-bound (SFNO C3, 8192 tokens: 5.68 -> 5.49 ms per step); at 32768 tokens each product fills
    // the chip by itself and the bf16 x bf16 one has a faster kernel of its own (11.5 -> 11.9 ms grouped)
    bool grouped = !off && n >= 2 && n <= 3;
    // ... and only small outputs: a FourCastNet-scale product (3072 x 768 over 16200 tokens) belongs on the LDS-DMA kernels (grouped on the
    // 64 x 64 kernel the C5 step went from 17.3 to 26.9 ms)
    const int maxt_env = dlwp_tune("WGRAD_GROUP_MAXT");
    const int maxt = maxt_env != DLWP_TUNE_UNSET ? maxt_env : 65536;
    for (int i = 0; i < n; ++i) grouped = grouped && d[i].T <= maxt && (long long)d[i].N * d[i].K <= 512 * 512;
    GemmGroup gg{};
    unsigned gx = 0, gz = 0;
    for (int i = 0; grouped && i < n; ++i) {
        const dlwp_wgrad_desc& q = d[i];
        const int M = q.N, N = q.K, K = q.T;                  // gW [N][K] = g^T [N][T] . x [T][K]
        const int dt = (q.g_bf16 ? DT_A : 0) | (q.x_bf16 ? DT_B : 0);
        const int tiles = ceil_div(N, 64) * ceil_div(M, 64);
        // slices: the products of the group share the chip, so the whole group gets the workgroups of one resident round
        // (896 by default, DLWP_WGRAD_GROUP_WGS): with 512 per product as a lone launch would choose, three products ran a
        // second partial round and twice the atomics (SFNO C3 step 5.33 -> 5.21 ms; neutral on Pangu / Swin / AFNO)
        const int wgs_env = dlwp_tune("WGRAD_GROUP_WGS");
        const int per_product = std::max(1, (wgs_env != DLWP_TUNE_UNSET ? wgs_env : 896) / n);
        int splits = 1;
        if (tiles < 256 && K >= 8 * BK) splits = std::min(ceil_div(per_product, tiles), K / (4 * BK));
        const int kchunk = ceil_div(ceil_div(K, splits), BK) * BK;
        splits = ceil_div(K, kchunk);
        GemmDev a{(const float*)q.g, (const float*)q.x, nullptr, nullptr, q.gw, nullptr, q.gb, M, N, K, M, N, N, 0, q.accumulate, kchunk, splits,
                  1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.f, 0, 0, 0};
        a.dt = dt;
        a.ntn = ceil_div(N, 64);
        a.ntm = ceil_div(M, 64);
        const bool vecA = ((uintptr_t)a.A % ((dt & DT_A) ? 8 : 16) == 0) && a.lda % 4 == 0 && a.M % 4 == 0;
        const bool vecB = ((uintptr_t)a.B % ((dt & DT_B) ? 8 : 16) == 0) && a.ldb % 4 == 0 && a.N % 4 == 0;
        if (!vecA || !vecB || splits * 1LL > 65535) { grouped = false; break; }
        a.vec_epi = splits == 1 && a.N % 4 == 0 && a.ldc % 4 == 0 && (uintptr_t)a.C % 16 == 0;
        gg.g[i] = a;
        // bf16 arrays that allow the 16-byte path (as gemm_launch decides it for a single product); K-step 32 for every member
        auto ok16 = [&](const float* p, int ld, int rows) { return (uintptr_t)p % 16 == 0 && ld % 8 == 0 && rows % 8 == 0; };
        gg.s16m[i] = (((dt & DT_A) && ok16(a.A, a.lda, a.M)) ? 1 : 0) | (((dt & DT_B) && ok16(a.B, a.ldb, a.N)) ? 2 : 0);
        gx = std::max(gx, (unsigned)(a.ntn * a.ntm));
        gz = std::max(gz, (unsigned)splits);
    }
    if (!grouped) {
        for (int i = 0; i < n; ++i)
            if (int rc = single(d[i])) return rc;
        return DLWP_OK;
    }
    for (int i = 0; i < n; ++i)
        if (gg.g[i].splits > 1 && !gg.g[i].accumulate)
            if (int zrc = dlwp_zero_2d_f32(gg.g[i].C, gg.g[i].ldc, gg.g[i].M, gg.g[i].N, stream)) return zrc;
    const dim3 grid(gx, n, gz);
    const hipStream_t s = (hipStream_t)stream;
    int rc;
    if (g_gemm_bf16) {
        constexpr int KS = gemm_bf_kstep(0, 1, false, false);
        const size_t lds = sizeof(float) * 4 * ((64 * (KS + 8) > KS * (64 + 8) ? 64 * (KS + 8) : KS * (64 + 8)) / 2);
        if ((rc = dlwp_ensure_lds(reinterpret_cast<const void*>(gemm_group_kernel<false, false, 3, 1, true, 0>), lds, "gemm_group"))) return rc;
        hipLaunchKernelGGL((gemm_group_kernel<false, false, 3, 1, true, 0>), grid, dim3(256), lds, s, gg);
    } else {
        const size_t lds = sizeof(float) * 4 * Tile<1>::FLOATS;
        if ((rc = dlwp_ensure_lds(reinterpret_cast<const void*>(gemm_group_kernel<false, false, 3, 1, false, 0>), lds, "gemm_group"))) return rc;
        hipLaunchKernelGGL((gemm_group_kernel<false, false, 3, 1, false, 0>), grid, dim3(256), lds, s, gg);
    }
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_gemm(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                         int transA, int transB, const float* bias, int act, float* preact, const float* residual,
                         int accumulate, float* rowsum, void* stream) {
    return gemm_impl(A, B, C, M, N, K, lda, ldb, ldc, transA, transB, bias, act, preact, residual, accumulate, rowsum, 0, stream);
}

extern "C" int dlwp_gemm_mixed(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                               int transA, int transB, const float* bias, int act, void* preact, const void* residual,
                               int accumulate, float* rowsum, int dtypes, void* stream) {
    return gemm_impl((const float*)A, (const float*)B, (float*)C, M, N, K, lda, ldb, ldc, transA, transB, bias, act,
                     (float*)preact, (const float*)residual, accumulate, rowsum, dtypes, stream);
}

extern "C" int dlwp_gemm_rowscale(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                  int transA, int transB, const float* bias, const void* residual, const float* row_scale,
                                  int rows_per_scale, int dtypes, void* stream) {
    DLWP_REQUIRE(row_scale && rows_per_scale > 0, DLWP_E_INVALID, "gemm_rowscale: row_scale is NULL or rows_per_scale < 1");
    return gemm_impl((const float*)A, (const float*)B, (float*)C, M, N, K, lda, ldb, ldc, transA, transB, bias, 0, nullptr,
                     (const float*)residual, 0, nullptr, dtypes, stream, row_scale, rows_per_scale);
}

static int gemm_batched_impl(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                             int transA, int transB, int nb1, int nb2, long long sA1, long long sA2, long long sB1,
                             long long sB2, long long sC1, long long sC2, const float* bias, long long sBi1,
                             long long sBi2, int act, float act_param, float* preact, const float* residual,
                             long long sR1, long long sR2, int res_before_act, int accumulate, int dt, void* stream) {
    DLWP_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && nb1 > 0 && nb2 > 0, DLWP_E_INVALID,
                 "gemm_batched: NULL argument or empty shape");
    if (int drc = dtypes_ok(dt, accumulate, "gemm_batched")) return drc;
    DLWP_REQUIRE(act >= 0 && act <= ACT_MUL, DLWP_E_INVALID,
                 "gemm_batched: act must be 0 (none), 1 (gelu), 2 (relu), 3 (softshrink), 4 / 5 / 6 (multiply by GELU' / ReLU' / "
                 "softshrink'(residual)), 7 (gelu, preact receives GELU') or 8 (multiply by residual)");
    DLWP_REQUIRE(!act_is_grad_mul(act) || (residual && !preact), DLWP_E_INVALID,
                 "gemm_batched: act 4-6 and 8 read the saved pre-activation / factor through `residual` and write no `preact`");
    DLWP_REQUIRE(act != ACT_GELU_STORE_D || preact, DLWP_E_INVALID, "gemm_batched: act 7 stores GELU' to `preact`: it must not be NULL");
    // reductions over a long K with few output tiles (weight gradients of block-diagonal layers): split K inside every
    // batch and combine with float atomics, like the plain entry
    const bool epilogue = bias || act || preact || residual || (dt & DT_C);
    const int T = gemm_tile_for(M, N, (long long)nb1 * nb2, K, dt, !epilogue);
    const long long tiles = (long long)ceil_div(N, 64 * T) * ceil_div(M, 64 * T) * nb1 * nb2;
    int splits = 1;
    if (!epilogue && tiles < 256 && K >= 8 * BK) splits = (int)std::min<long long>(ceil_div(512, (int)tiles), K / (4 * BK));
    int kchunk = ceil_div(ceil_div(K, splits), BK) * BK;
    splits = ceil_div(K, kchunk);
    DLWP_REQUIRE((long long)nb1 * nb2 * splits <= 65535, DLWP_E_UNSUPPORTED, "gemm_batched: more than 65535 batches (%d x %d)", nb1, nb2);
    if (splits > 1 && !accumulate) {
        // C must start from zero: every batch's [M x N] block (row pitch ldc)
        for (int z1 = 0; z1 < nb1; ++z1)
            for (int z2 = 0; z2 < nb2; ++z2)
                if (int zrc = dlwp_zero_2d_f32(C + z1 * sC1 + z2 * sC2, ldc, M, N, stream)) return zrc;
    }
    GemmDev a{A, B, bias, residual, C, preact, nullptr, M, N, K, lda, ldb, ldc, act, accumulate, kchunk, splits,
              nb1 * nb2, nb2, res_before_act, 0, 0, 0, sA1, sA2, sB1, sB2, sC1, sC2, sR1, sR2, sBi1, sBi2, act_param, 0, 0, 0};
    a.dt = dt;
    return gemm_dispatch(a, transA, transB, T, stream);
}

extern "C" int dlwp_gemm_batched(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                                 int transA, int transB, int nb1, int nb2, long long sA1, long long sA2, long long sB1,
                                 long long sB2, long long sC1, long long sC2, const float* bias, long long sBi1,
                                 long long sBi2, int act, float act_param, float* preact, const float* residual,
                                 long long sR1, long long sR2, int res_before_act, int accumulate, void* stream) {
    return gemm_batched_impl(A, B, C, M, N, K, lda, ldb, ldc, transA, transB, nb1, nb2, sA1, sA2, sB1, sB2, sC1, sC2, bias, sBi1,
                             sBi2, act, act_param, preact, residual, sR1, sR2, res_before_act, accumulate, 0, stream);
}

extern "C" int dlwp_gemm_batched_mixed(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                       int transA, int transB, int nb1, int nb2, long long sA1, long long sA2, long long sB1,
                                       long long sB2, long long sC1, long long sC2, const float* bias, long long sBi1,
                                       long long sBi2, int act, float act_param, void* preact, const void* residual,
                                       long long sR1, long long sR2, int res_before_act, int accumulate, int dtypes,
                                       void* stream) {
    return gemm_batched_impl((const float*)A, (const float*)B, (float*)C, M, N, K, lda, ldb, ldc, transA, transB, nb1, nb2, sA1,
                             sA2, sB1, sB2, sC1, sC2, bias, sBi1, sBi2, act, act_param, (float*)preact, (const float*)residual,
                             sR1, sR2, res_before_act, accumulate, dtypes, stream);
}

#ifdef DLWP_STAMPS
extern "C" int dlwp_debug_stamps_gemm(unsigned long long* host_out) {
    DLWP_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dlwp_stamps), sizeof(unsigned long long) * 32));
    return DLWP_OK;
}
#endif

// Three dependent token products in ONE launch: the tail of an SFNO block and its input-gradient chain (round 4).
//
// Reference: the block of torch_harmonics' SphericalFourierNeuralOperatorNet that dlwpbench constructs at
// /root/reference/src/dlwpbench/models/fno/fno.py:183-200 (and models/fourcastnet/fourcastnet.py:411-428): after the spectral
// filter,  t = GELU(y + inner_skip(x)),  out = fc2(GELU(fc1 t)) + x   (1x1 convolutions = token GEMMs).
//   forward   stage 1  z0 = y + x Ws^T + bs      t  = GELU(z0)            [T, C]
//             stage 2  z1 = t W1^T + b1          h  = GELU(z1)            [T, Hd]
//             stage 3  out = h W2^T + b2 (+ x)                            [T, C]
//   backward  stage 1  gh = (g W2) * GELU'(z1)                            [T, Hd]
//             stage 2  gt = (gh W1) * GELU'(z0)                           [T, C]   (= gradient of y)
//             stage 3  gx = gt Ws (+ g)                                   [T, C]
// As three launches of the token GEMM these cost 42 us forward and 36 us backward at the C3 shape (8192 tokens, C = 256, Hd = 512;
// profiles/r03_bf16_storage_sfno_step_kernel_stats.csv): each is a single round of latency-bound workgroups that writes a
// [T, 256..512] tensor for the next one to read back.  Here a workgroup owns 32 (or 64) tokens for the whole chain:
//   * the 655 KB of bf16 weights are NOT staged: every weight element is used by exactly one wave (the eight waves split the
//     output features of a stage), so the fragments go L2 -> registers, as whole 1 KB wave-loads from an image that holds them in
//     MFMA-fragment order (dlwp_mlp_chain_pack, once per optimizer step); a stage's fragments are requested one stage ahead;
//   * products are computed TRANSPOSED (weights = A operand, tokens = B operand): a lane then owns four consecutive features of
//     one token, i.e. 8-byte bf16 / 16-byte fp32 pieces for the epilogue loads, the stores and the LDS hand-off;
//   * the stage outputs that the next stage consumes live in LDS as [token][feature] bf16 images (16-byte chunks XOR-swizzled
//     by the token row so that the ds_read_b128 fragment reads are conflict-free); the copies that the weight-gradient
//     products need (x, t, h / g, gh, gt as bf16 arrays) and the pre-activations leave from registers.
// Arithmetic = the bf16-operand token GEMM's: operands rounded to bf16, fp32 accumulation, fp32 epilogue, one rounding per store.
#include <algorithm>
#include <cstdlib>
#include "chain_frag.hip.h"
#include "dlwpmi_internal.h"

namespace {

using namespace chainfrag;

// image[(tile * KS + kk) * 64 + lane][e] = W'[16 tile + (lane & 15)][32 kk + 8 (lane >> 4) + e],  W' = W or W^T
__global__ __launch_bounds__(256) void chain_pack_kernel(const float* __restrict__ W, int ld, int rows, int cols, int transpose,
                                                         __bf16* __restrict__ img) {
    const int KS = cols / 32;
    const long long total = (long long)(rows / 16) * KS * 64;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const long long f = idx >> 6;
    const int kk = (int)(f % KS), tile = (int)(f / KS);
    const int row = 16 * tile + (lane & 15), k0 = 32 * kk + 8 * (lane >> 4);
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)(transpose ? W[(long long)(k0 + e) * ld + row] : W[(long long)row * ld + k0 + e]);
    *reinterpret_cast<bf16x8*>(img + idx * 8) = v;
}

// the six images of a block tail in one launch: blockIdx.y = image (forward Ws, W1, W2; backward W2^T, W1^T, Ws^T)
struct PackSix { const float* W[6]; int ld[6], rows[6], cols[6], tr[6]; __bf16* img[6]; };
constexpr int PACK_MANY = 8;          // block tails per dlwp_sfno_tail_pack_many launch
struct PackSixMany { PackSix b[PACK_MANY]; };
__global__ __launch_bounds__(256) void chain_pack6_kernel(PackSixMany many) {
    const PackSix& a = many.b[blockIdx.z];
    const int i = blockIdx.y;
    const int KS = a.cols[i] / 32;
    const long long total = (long long)(a.rows[i] / 16) * KS * 64;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const long long f = idx >> 6;
    const int kk = (int)(f % KS), tile = (int)(f / KS);
    const int row = 16 * tile + (lane & 15), k0 = 32 * kk + 8 * (lane >> 4);
    const float* W = a.W[i];
    const int ld = a.ld[i];
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)(a.tr[i] ? W[(long long)(k0 + e) * ld + row] : W[(long long)row * ld + k0 + e]);
    *reinterpret_cast<bf16x8*>(a.img[i] + idx * 8) = v;
}

struct ChainDev {
    const float* in;              // [T][K1] fp32: forward x (block input = outer-skip residual); backward g (gradient of the block output)
    __bf16* in_lp;                // [T][K1] bf16 copy of it (operand of a weight-gradient product); nullable
    const __bf16 *w1, *w2, *w3;   // fragment-order images of the three stage matrices [N_s][K_s]
    const float *b1, *b2, *b3;    // forward biases (nullable)
    const float* res1;            // forward: y [T][N1], added before the first activation
    const __bf16 *zin1, *zin2;    // backward: stored pre-activations z1 [T][N1], z0 [T][N2]
    __bf16 *z1, *a1, *z2, *a2;    // forward: z0, t, z1, h; backward: a1 = gh, a2 = bf16 copy of gt (z1 / z2 unused)
    float* a2f;                   // backward: gt in fp32 (the gradient that leaves through y)
    float* out;                   // [T][N3] fp32
    int T, outer;
    int res1_bf16;                // forward: res1 is a bf16 array
    int rot;                      // rotate the wave -> n-tile assignment by the workgroup index (spreads the L2 channels the CUs of an XCD hit at one time)
};

template <int K1, int N1, int N2, int N3, int MT, bool BWD>
__global__ __launch_bounds__(512) void mlp_chain_kernel(ChainDev a) {
    constexpr int ROWS = 16 * MT;
    constexpr int KS1 = K1 / 32, KS2 = N1 / 32, KS3 = N2 / 32;              // k-steps of the three stages
    constexpr int NTL1 = N1 / 16, NTL2 = N2 / 16, NTL3 = N3 / 16;           // n-tiles
    constexpr int NT1 = (NTL1 + 7) / 8, NT2 = (NTL2 + 7) / 8, NT3 = (NTL3 + 7) / 8;      // ... per wave
    constexpr int KH1 = KS1 / 2, KH2 = KS2 / 2, KH3 = KS3 / 2;              // weight fragments travel in two halves per stage
    static_assert(K1 % 64 == 0 && N1 % 64 == 0 && N2 % 64 == 0 && N3 % 16 == 0, "stage widths");
    extern __shared__ __attribute__((aligned(16))) float chain_smem[];
    __bf16* img0 = reinterpret_cast<__bf16*>(chain_smem);                   // [ROWS][K1]  stage-1 input
    __bf16* img1 = img0 + ROWS * K1;                                        // [ROWS][N1]  stage-1 output
    __bf16* img2 = img1 + ROWS * N1;                                        // [ROWS][N2]  stage-2 output
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
    const int w = ((tid >> 6) + (a.rot ? (int)(blockIdx.x >> 3) : 0)) & 7;     // which n-tiles this wave owns (workgroups b, b + 8, .. share an XCD)
    const int m0 = blockIdx.x * ROWS;
    DLWP_STAMP(0);

    // ---- requests of the prologue, oldest first: input tile, stage-1 weights, stage-1 epilogue operands, first half of stage 2
    InTile<ROWS, K1> xin;
    xin.issue(a.in, m0, a.T, tid);
    WFrag<NT1, KH1> w1a, w1b;
    wload<NT1, KS1, KH1, NTL1>(w1a, a.w1, w, lane, 0);
    wload<NT1, KS1, KH1, NTL1>(w1b, a.w1, w, lane, KH1);
    f32x4 e1f[MT][NT1];           // forward: y + bias
    bf16x4 e1z[MT][NT1];          // backward: z1
#pragma unroll
    for (int ni = 0; ni < NT1; ++ni) {
        const int n = 16 * min(w + 8 * ni, NTL1 - 1) + 4 * g;
        f32x4 bb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!BWD) bb = *reinterpret_cast<const f32x4*>(a.b1 + n);          // (the host passes zeros for an absent bias)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const long long m = min(m0 + 16 * mi + r, a.T - 1);
            e1z[mi][ni] = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            e1f[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (BWD) e1z[mi][ni] = *reinterpret_cast<const bf16x4*>(a.zin1 + m * N1 + n);
            else if (a.res1_bf16) {
                const bf16x4 yv = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(a.res1) + m * N1 + n);
                e1f[mi][ni] = f32x4{(float)yv[0], (float)yv[1], (float)yv[2], (float)yv[3]} + bb;
            } else e1f[mi][ni] = *reinterpret_cast<const f32x4*>(a.res1 + m * N1 + n) + bb;
        }
    }
    WFrag<NT2, KH2> w2a, w2b;
    constexpr bool EARLY2 = NT1 * KS1 <= 16;      // registers: a wide first stage (backward: 32 fragments) leaves no room for stage 2's yet
    if (EARLY2) wload<NT2, KS2, KH2, NTL2>(w2a, a.w2, w, lane, 0);

    // ---- input tile -> bf16 image (and the bf16 copy for the weight-gradient product)
    xin.commit(img0, a.in_lp, m0, a.T, tid);
    DLWP_STAMP(1);
    lds_barrier();
    DLWP_STAMP(2);

    // ---- stage 1
    f32x4 acc1[MT][NT1];
    zero_acc<MT, NT1>(acc1);
    mma<MT, NT1, KH1, K1>(acc1, w1a, img0, 0, r, g);
    if (!EARLY2) wload<NT2, KS2, KH2, NTL2>(w2a, a.w2, w, lane, 0);
    mma<MT, NT1, KH1, K1>(acc1, w1b, img0, KH1, r, g);
    DLWP_STAMP(3);
    // requests for stage 2's epilogue and the rest of its weights go out before this stage's stores
    f32x4 e2b[NT2];
    bf16x4 e2z[MT][NT2];
#pragma unroll
    for (int ni = 0; ni < NT2; ++ni) {
        const int n = 16 * min(w + 8 * ni, NTL2 - 1) + 4 * g;
        e2b[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!BWD) e2b[ni] = *reinterpret_cast<const f32x4*>(a.b2 + n);
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            e2z[mi][ni] = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            if (BWD) e2z[mi][ni] = *reinterpret_cast<const bf16x4*>(a.zin2 + (long long)min(m0 + 16 * mi + r, a.T - 1) * N2 + n);
        }
    }
    wload<NT2, KS2, KH2, NTL2>(w2b, a.w2, w, lane, KH2);
    chain_epilogue<MT, NT1, N1, NTL1, BWD>(acc1, [&](int mi, int ni) { return e1f[mi][ni]; }, e1z, img1, a.a1, a.z1, nullptr, m0, a.T, w, r, g);
    DLWP_STAMP(4);
    lds_barrier();
    DLWP_STAMP(5);

    // ---- stage 2
    f32x4 acc2[MT][NT2];
    zero_acc<MT, NT2>(acc2);
    mma<MT, NT2, KH2, N1>(acc2, w2a, img1, 0, r, g);
    mma<MT, NT2, KH2, N1>(acc2, w2b, img1, KH2, r, g);
    DLWP_STAMP(6);
    WFrag<NT3, KH3> w3a, w3b;
    f32x4 e3f[MT][NT3];           // bias (+ outer skip)
#pragma unroll
    for (int ni = 0; ni < NT3; ++ni) {
        const int n = 16 * min(w + 8 * ni, NTL3 - 1) + 4 * g;
        f32x4 bb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!BWD) bb = *reinterpret_cast<const f32x4*>(a.b3 + n);
        const float so = a.outer ? 1.f : 0.f;                              // the skip is always loaded (no branch around a load)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
            e3f[mi][ni] = bb + so * *reinterpret_cast<const f32x4*>(a.in + (long long)min(m0 + 16 * mi + r, a.T - 1) * K1 + n);
    }
    wload<NT3, KS3, KH3, NTL3>(w3a, a.w3, w, lane, 0);
    wload<NT3, KS3, KH3, NTL3>(w3b, a.w3, w, lane, KH3);
    chain_epilogue<MT, NT2, N2, NTL2, BWD>(acc2, [&](int, int ni) { return e2b[ni]; }, e2z, img2, a.a2, a.z2, BWD ? a.a2f : nullptr, m0, a.T, w, r, g);
    DLWP_STAMP(7);
    lds_barrier();
    DLWP_STAMP(8);

    // ---- stage 3
    f32x4 acc3[MT][NT3];
    zero_acc<MT, NT3>(acc3);
    mma<MT, NT3, KH3, N2>(acc3, w3a, img2, 0, r, g);
    mma<MT, NT3, KH3, N2>(acc3, w3b, img2, KH3, r, g);
    DLWP_STAMP(9);
#pragma unroll
    for (int ni = 0; ni < NT3; ++ni) {
        const int tile = w + 8 * ni, n = 16 * tile + 4 * g;
        if (tile < NTL3) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                const long long m = m0 + 16 * mi + r;
                if (m < a.T) *reinterpret_cast<f32x4*>(a.out + m * N3 + n) = acc3[mi][ni] + e3f[mi][ni];
            }
        }
    }
    DLWP_STAMP(10);
#ifdef DLWP_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DLWP_STAMP(11);
#endif
}

template <int C, int HD, int MT, bool BWD>
int chain_launch(const ChainDev& a, hipStream_t s) {
    constexpr int K1 = C, N1 = BWD ? HD : C, N2 = BWD ? C : HD, N3 = C;
    const size_t lds = (size_t)16 * MT * (K1 + N1 + N2) * sizeof(__bf16);
    auto kern = mlp_chain_kernel<K1, N1, N2, N3, MT, BWD>;
    int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(kern), lds, "mlp_chain");
    if (rc) return rc;
    ChainDev b = a;
    b.rot = dlwp_tune_or("CHAIN_ROT", 1);
    hipLaunchKernelGGL(kern, dim3(ceil_div(a.T, 16 * MT)), dim3(512), lds, s, b);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

template <bool BWD>
int chain_dispatch(const ChainDev& a, int C, int HD, hipStream_t s) {
    if (C == 256 && HD == 512) return chain_launch<256, 512, 2, BWD>(a, s);
    if (C == 128 && HD == 256) return chain_launch<128, 256, 2, BWD>(a, s);
    if (C == 64 && HD == 128) return chain_launch<64, 128, 2, BWD>(a, s);
    dlwp_set_error("mlp_chain: no kernel for C = %d, hidden = %d (dlwp_mlp_chain_supported)", C, HD);
    return DLWP_E_UNSUPPORTED;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// an absent bias is read as zeros from here (the kernel's bias loads are unconditional); sized for the widest compiled stage
__device__ __attribute__((aligned(16))) float g_chain_zero_bias[512];
const float* zero_bias() {
    // per-device address, resolved on every call (hipGetSymbolAddress does not enqueue anything: capture-safe)
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_chain_zero_bias)) != hipSuccess) return nullptr;
    return static_cast<const float*>(p);
}

}  // namespace

extern "C" int dlwp_mlp_chain_supported(int C, int hidden) {
    return (C == 256 && hidden == 512) || (C == 128 && hidden == 256) || (C == 64 && hidden == 128);
}

extern "C" int dlwp_mlp_chain_pack(const float* W, int rows, int cols, int transpose, void* image, void* stream) {
    DLWP_REQUIRE(W && image, DLWP_E_INVALID, "mlp_chain_pack: null pointer");
    // the image has `rows` x `cols` of W' = W (transpose 0: W is [rows][cols]) or W^T (transpose 1: W is [cols][rows])
    DLWP_REQUIRE(rows > 0 && cols > 0 && rows % 16 == 0 && cols % 32 == 0, DLWP_E_INVALID,
                 "mlp_chain_pack: image rows %d must be a multiple of 16 and columns %d of 32", rows, cols);
    DLWP_REQUIRE(aligned16(image), DLWP_E_INVALID, "mlp_chain_pack: the image must be 16-byte aligned");
    const long long total = (long long)(rows / 16) * (cols / 32) * 64;
    hipLaunchKernelGGL(chain_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W,
                       transpose ? rows : cols, rows, cols, transpose, reinterpret_cast<__bf16*>(image));
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_sfno_tail_pack_many(const float* const* ws, const float* const* w1, const float* const* w2, int n, int C, int hidden,
                                        void* const* images, void* stream) {
    DLWP_REQUIRE(ws && w1 && w2 && images && n >= 1 && n <= PACK_MANY, DLWP_E_INVALID, "sfno_tail_pack: %d block tails (1..%d) / null table", n, PACK_MANY);
    DLWP_REQUIRE(C > 0 && hidden > 0 && C % 32 == 0 && hidden % 32 == 0, DLWP_E_INVALID,
                 "sfno_tail_pack: C = %d and hidden = %d must be multiples of 32", C, hidden);
    PackSixMany many{};
    long long most = 0;
    for (int b = 0; b < n; ++b) {
        DLWP_REQUIRE(ws[b] && w1[b] && w2[b] && images[b] && aligned16(images[b]), DLWP_E_INVALID, "sfno_tail_pack: block %d: null / unaligned pointer", b);
        PackSix& a = many.b[b];
        const float* W[6] = {ws[b], w1[b], w2[b], w2[b], w1[b], ws[b]};
        const int rows[6] = {C, hidden, C, hidden, C, C}, cols[6] = {C, C, hidden, C, hidden, C}, tr[6] = {0, 0, 0, 1, 1, 1};
        for (int i = 0; i < 6; ++i) {
            a.W[i] = W[i];
            a.rows[i] = rows[i];
            a.cols[i] = cols[i];
            a.tr[i] = tr[i];
            a.ld[i] = tr[i] ? rows[i] : cols[i];
            a.img[i] = reinterpret_cast<__bf16*>(images[b]) + (long long)i * C * hidden;
            most = std::max(most, (long long)(rows[i] / 16) * (cols[i] / 32) * 64);
        }
    }
    hipLaunchKernelGGL(chain_pack6_kernel, dim3((unsigned)((most + 255) / 256), 6, n), dim3(256), 0, (hipStream_t)stream, many);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_sfno_tail_pack(const float* ws, const float* w1, const float* w2, int C, int hidden, void* images, void* stream) {
    DLWP_REQUIRE(ws && w1 && w2 && images, DLWP_E_INVALID, "sfno_tail_pack: null pointer");
    return dlwp_sfno_tail_pack_many(&ws, &w1, &w2, 1, C, hidden, &images, stream);
}

extern "C" int dlwp_sfno_tail_fwd(const dlwp_sfno_tail_fwd_args* p, void* stream) {
    DLWP_REQUIRE(p, DLWP_E_INVALID, "sfno_tail_fwd: null arguments");
    DLWP_REQUIRE(dlwp_mlp_chain_supported(p->C, p->hidden), DLWP_E_UNSUPPORTED, "sfno_tail_fwd: no kernel for C = %d, hidden = %d",
                 p->C, p->hidden);
    DLWP_REQUIRE(p->T > 0 && p->x && p->y && p->ws_img && p->w1_img && p->w2_img && p->z0 && p->t && p->z1 && p->h && p->out,
                 DLWP_E_INVALID, "sfno_tail_fwd: null tensor");
    const void* ptrs[] = {p->x, p->y, p->ws_img, p->w1_img, p->w2_img, p->z0, p->t, p->z1, p->h, p->out, p->x_lp, p->bs, p->b1, p->b2};
    for (const void* q : ptrs) DLWP_REQUIRE(aligned16(q), DLWP_E_INVALID, "sfno_tail_fwd: tensors must be 16-byte aligned");
    ChainDev a{};
    a.in = p->x;
    a.in_lp = reinterpret_cast<__bf16*>(p->x_lp);
    a.w1 = reinterpret_cast<const __bf16*>(p->ws_img);
    a.w2 = reinterpret_cast<const __bf16*>(p->w1_img);
    a.w3 = reinterpret_cast<const __bf16*>(p->w2_img);
    const float* zb = zero_bias();
    DLWP_REQUIRE(zb, DLWP_E_HIP, "sfno_tail_fwd: hipGetSymbolAddress failed");
    a.b1 = p->bs ? p->bs : zb;
    a.b2 = p->b1 ? p->b1 : zb;
    a.b3 = p->b2 ? p->b2 : zb;
    a.res1 = p->y;
    a.res1_bf16 = p->y_bf16 != 0;
    a.z1 = reinterpret_cast<__bf16*>(p->z0);
    a.a1 = reinterpret_cast<__bf16*>(p->t);
    a.z2 = reinterpret_cast<__bf16*>(p->z1);
    a.a2 = reinterpret_cast<__bf16*>(p->h);
    a.out = p->out;
    a.T = p->T;
    a.outer = p->outer;
    return chain_dispatch<false>(a, p->C, p->hidden, (hipStream_t)stream);
}

extern "C" int dlwp_sfno_tail_bwd(const dlwp_sfno_tail_bwd_args* p, void* stream) {
    DLWP_REQUIRE(p, DLWP_E_INVALID, "sfno_tail_bwd: null arguments");
    DLWP_REQUIRE(dlwp_mlp_chain_supported(p->C, p->hidden), DLWP_E_UNSUPPORTED, "sfno_tail_bwd: no kernel for C = %d, hidden = %d",
                 p->C, p->hidden);
    DLWP_REQUIRE(p->T > 0 && p->g && p->w2t_img && p->w1t_img && p->wst_img && p->z1 && p->z0 && p->gh && p->gt_lp && p->gx,
                 DLWP_E_INVALID, "sfno_tail_bwd: null tensor");
    const void* ptrs[] = {p->g, p->g_lp, p->w2t_img, p->w1t_img, p->wst_img, p->z1, p->z0, p->gh, p->gt, p->gt_lp, p->gx};
    for (const void* q : ptrs) DLWP_REQUIRE(aligned16(q), DLWP_E_INVALID, "sfno_tail_bwd: tensors must be 16-byte aligned");
    ChainDev a{};
    a.in = p->g;
    a.in_lp = reinterpret_cast<__bf16*>(p->g_lp);
    a.w1 = reinterpret_cast<const __bf16*>(p->w2t_img);
    a.w2 = reinterpret_cast<const __bf16*>(p->w1t_img);
    a.w3 = reinterpret_cast<const __bf16*>(p->wst_img);
    a.zin1 = reinterpret_cast<const __bf16*>(p->z1);
    a.zin2 = reinterpret_cast<const __bf16*>(p->z0);
    a.a1 = reinterpret_cast<__bf16*>(p->gh);
    a.a2 = reinterpret_cast<__bf16*>(p->gt_lp);
    a.a2f = p->gt;
    a.out = p->gx;
    a.T = p->T;
    a.outer = p->outer;
    return chain_dispatch<true>(a, p->C, p->hidden, (hipStream_t)stream);
}

#ifdef DLWP_STAMPS
extern "C" int dlwp_debug_stamps_chain(unsigned long long* host_out) {
    DLWP_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dlwp_stamps), sizeof(unsigned long long) * 32));
    return DLWP_OK;
}
#endif

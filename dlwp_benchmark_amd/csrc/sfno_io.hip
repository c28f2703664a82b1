// SFNO encoder and decoder as ONE launch per direction each (round 4): frame assembly, both 1x1-convolution layers, position
// embedding / big skip / residual frame and the NCHW <-> token layout changes inside the kernel.
//
// Reference: torch_harmonics' SphericalFourierNeuralOperatorNet as dlwpbench builds it (src/dlwpbench/models/fno/fno.py:183-200;
// SURVEY.md App. A-2) inside SFNO2DModule.forward's rollout (fno.py:217-259, clean form unet.py:64-111):
//   x_t  = cat(constants[:, 0], prescribed[:, t-1], prognostic frame)                 [B, Cin, H, W]
//   enc  : t0 = W2e GELU(W1e x_t + b1e) (+ pos_embed)                                   1x1 convolutions, embed E
//   ...  blocks ...
//   dec  : y  = W2d GELU(Wd [t ; x_t] + bd)   (big skip: the network input is concatenated to the block output)
//   out  = prognostic frame + y                                                         (the rollout's residual connection)
// As token GEMMs this was 2 + 2 products forward and 4 + 4 backward per lead time plus ~15 ATen launches (cat, permute copy, zero
// padding of the 10 / 266-wide rows, the position embedding expanded to the batch, the big-skip cat and their backward slices
// and sums): 0.84 ms of the 3.9 ms C3 step.  Here (E = embed, Cin <= 32 input channels, Cout <= 16 output channels):
//   encode_fwd  gather the Cin planes of 32 tokens -> [32][32] bf16 image; h = GELU(tok W1e^T + b1e); t0 = h W2e^T + pos
//   encode_bwd  gh = (g W2e) * GELU'(z); g_tok = gh W1e (+ the decoder's g_tok); the prognostic channels of g_tok (+ the
//               gradient that reached the frame through the decoder's residual) are scattered to the frame gradient [B, Cg, H, W]
//   decode_fwd  image [t | tok] (E + 32 wide); h = GELU([t | tok] Wd^T + bd); y = h W2d^T; out[b, c, h, w] = frame + y
//   decode_bwd  gather g_out's planes; gh = (g W2d) * GELU'(z); [g_t | g_tok] = gh Wd
// Kernel structure = csrc/mlp_chain.hip's (32 tokens per workgroup, eight waves split the output features, weights L2 ->
// registers from fragment-order images, products transposed so that a lane owns four consecutive features of one token, the
// hidden layer handed over in a swizzled LDS image); the bf16 copies that the weight-gradient products need (tok, h, t / g, gh)
// leave from the kernel.  The narrow dimensions are zero-padded inside the weight images (dlwp_sfno_io_pack), never in HBM tensors.
#include <algorithm>
#include <cstdlib>
#include "chain_frag.hip.h"
#include "dlwpmi_internal.h"

namespace {

using namespace chainfrag;

constexpr int KP = DLWP_SFNO_IO_KP;      // padded width of a frame-token row (Cin <= KP)
constexpr int NP = 16;                   // padded number of output channels (one MFMA tile)
constexpr int MT = 2, ROWS = 16 * MT;    // tokens per workgroup
enum { ENC_FWD = 0, ENC_BWD = 1, DEC_FWD = 2, DEC_BWD = 3 };

struct IoDev {
    const float* src[3];          // gather sources, NCHW planes of sample 0 (ENC_FWD: constants | prescribed | prognostic; DEC_BWD: g_out)
    long long src_bs[3];          // their batch strides (floats)
    int src_c[3];                 // their channel counts (0 = absent)
    int HW, T;
    const float* in;              // ENC_BWD: g [T][E]; DEC_FWD: t [T][E]
    const __bf16* in2;            // DEC_FWD: tok_lp [T][KP]
    __bf16* in_lp;                // bf16 copy of the stage-1 input rows (ENC_FWD, DEC_BWD: [T][KP]; ENC_BWD, DEC_FWD: [T][E])
    const __bf16 *w1, *w2;        // fragment-order images (zero-padded)
    const float* b1;              // forward bias [E]
    const __bf16* zin;            // backward: stored pre-activation [T][E]
    __bf16 *z, *a;                // forward: z, h [T][E]; backward: a = gh [T][E]
    const float* pos;             // ENC_FWD: [HW][E], nullable
    float* out;                   // ENC_FWD: t0 [T][E]; DEC_BWD: g_t [T][E]
    float* out2;                  // DEC_BWD: g_tok [T][KP]
    float* dst;                   // NCHW scatter target (DEC_FWD: out frame; ENC_BWD: frame gradient)
    long long dst_bs;
    int dst_c, dst_c0;            // channels scattered: token features dst_c0 .. dst_c0 + dst_c - 1
    const float* add1;            // NCHW addend with dst's channels (DEC_FWD: the frame; ENC_BWD: gradient through the residual), nullable
    long long add1_bs;
    const float* add2;            // ENC_BWD: [T][KP] addend (the decoder's g_tok), nullable
    int rot;
};

template <int E, int MODE>
__global__ __launch_bounds__(512) void sfno_io_kernel(IoDev a) {
    constexpr bool BWD = MODE == ENC_BWD || MODE == DEC_BWD;
    constexpr bool GATHER = MODE == ENC_FWD || MODE == DEC_BWD;
    constexpr int K1 = GATHER ? KP : (MODE == DEC_FWD ? E + KP : E);
    constexpr int N1 = E;
    constexpr int N2 = MODE == ENC_FWD ? E : (MODE == ENC_BWD ? KP : (MODE == DEC_FWD ? NP : E + KP));
    constexpr int KS1 = K1 / 32, KS2 = N1 / 32;
    constexpr int NTL1 = N1 / 16, NTL2 = N2 / 16;
    constexpr int NT1 = (NTL1 + 7) / 8, NT2 = (NTL2 + 7) / 8;
    extern __shared__ __attribute__((aligned(16))) float io_smem[];
    __bf16* img0 = reinterpret_cast<__bf16*>(io_smem);          // [ROWS][K1]
    __bf16* img1 = img0 + ROWS * K1;                            // [ROWS][N1]
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
    const int w = ((tid >> 6) + (a.rot ? (int)(blockIdx.x >> 3) : 0)) & 7;
    const int m0 = blockIdx.x * ROWS;

    // ---- requests, oldest first: stage-1 input, both stages' weights, epilogue operands
    float gv[2] = {0.f, 0.f};                                   // GATHER: channels (tid >> 5) and (tid >> 5) + 16 of token tid & 31
    constexpr int XU = GATHER ? 1 : (ROWS * E / 4 + 511) / 512;
    float4 xv[XU];
    bf16x8 tokv = {};
    if constexpr (GATHER) {
        const int m = min(m0 + (tid & 31), a.T - 1), b = m / a.HW, p = m - b * a.HW;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int ch = (tid >> 5) + 16 * i, s = 0;
            while (s < 3 && ch >= a.src_c[s]) ch -= a.src_c[s++];
            // (a branch around this load is harmless: nothing older is in flight)
            if (s < 3) gv[i] = a.src[s][b * a.src_bs[s] + (long long)ch * a.HW + p];
        }
    } else {
#pragma unroll
        for (int i = 0; i < XU; ++i) {
            const int u = min(tid + 512 * i, ROWS * E / 4 - 1), row = u / (E / 4), c4 = u - row * (E / 4);
            xv[i] = *reinterpret_cast<const float4*>(a.in + (long long)min(m0 + row, a.T - 1) * E + 4 * c4);
        }
        if constexpr (MODE == DEC_FWD) {
            const int t = min(tid, ROWS * KP / 8 - 1), row = t / (KP / 8), ch = t - row * (KP / 8);
            tokv = *reinterpret_cast<const bf16x8*>(a.in2 + (long long)min(m0 + row, a.T - 1) * KP + 8 * ch);
        }
    }
    WFrag<NT1, KS1> wf1;
    wload<NT1, KS1, KS1, NTL1>(wf1, a.w1, w, lane, 0);
    f32x4 e1b[NT1];
    bf16x4 e1z[MT][NT1];
#pragma unroll
    for (int ni = 0; ni < NT1; ++ni) {
        const int n = 16 * min(w + 8 * ni, NTL1 - 1) + 4 * g;
        e1b[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (!BWD) e1b[ni] = *reinterpret_cast<const f32x4*>(a.b1 + n);
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            e1z[mi][ni] = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            if constexpr (BWD) e1z[mi][ni] = *reinterpret_cast<const bf16x4*>(a.zin + (long long)min(m0 + 16 * mi + r, a.T - 1) * N1 + n);
        }
    }
    WFrag<NT2, KS2> wf2;
    wload<NT2, KS2, KS2, NTL2>(wf2, a.w2, w, lane, 0);

    // ---- stage-1 input image
    if constexpr (GATHER) {
        const int row = tid & 31;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = (tid >> 5) + 16 * i;
            const __bf16 v = (__bf16)gv[i];
            img0[row * KP + 8 * ((ch >> 3) ^ (row & 3)) + (ch & 7)] = v;
            if (a.in_lp && m0 + row < a.T) a.in_lp[(long long)(m0 + row) * KP + ch] = v;
        }
    } else {
#pragma unroll
        for (int i = 0; i < XU; ++i) {
            const int u = tid + 512 * i;
            if (u < ROWS * E / 4) {
                const int row = u / (E / 4), k = 4 * (u - row * (E / 4));
                const float v[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
                const bf16x4 b = to_bf4(v);
                img_store<K1>(img0, row, k, b);
                if (a.in_lp && m0 + row < a.T) *reinterpret_cast<bf16x4*>(a.in_lp + (long long)(m0 + row) * E + k) = b;
            }
        }
        if constexpr (MODE == DEC_FWD) {
            if (tid < ROWS * KP / 8) {
                const int row = tid / (KP / 8), c = E / 8 + (tid - row * (KP / 8));
                *reinterpret_cast<bf16x8*>(img0 + row * K1 + 8 * (c ^ (row & cmask<K1>(c)))) = tokv;
            }
        }
    }
    lds_barrier();

    // ---- stage 1
    f32x4 acc1[MT][NT1];
    zero_acc<MT, NT1>(acc1);
    mma<MT, NT1, KS1, K1>(acc1, wf1, img0, 0, r, g);
    // (round 5) the block tail's epilogue: 16-byte pieces, and the forward stores GELU' (in the array called z) for the backward
    // launch to multiply by
    chain_epilogue<MT, NT1, N1, NTL1, BWD>(acc1, [&](int, int ni) { return e1b[ni]; }, e1z, img1, a.a, a.z, nullptr, m0, a.T, w, r, g);
    lds_barrier();

    // ---- stage 2
    f32x4 acc2[MT][NT2];
    zero_acc<MT, NT2>(acc2);
    mma<MT, NT2, KS2, N1>(acc2, wf2, img1, 0, r, g);
#pragma unroll
    for (int ni = 0; ni < NT2; ++ni) {
        const int tile = w + 8 * ni, n = 16 * tile + 4 * g;
        if (tile < NTL2) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                const long long m = m0 + 16 * mi + r;
                if (m >= a.T) continue;
                if constexpr (MODE == ENC_FWD) {
                    f32x4 v = acc2[mi][ni];
                    if (a.pos) v += *reinterpret_cast<const f32x4*>(a.pos + (m % a.HW) * E + n);
                    *reinterpret_cast<f32x4*>(a.out + m * E + n) = v;
                } else if constexpr (MODE == DEC_BWD) {
                    if (n < E) *reinterpret_cast<f32x4*>(a.out + m * E + n) = acc2[mi][ni];
                    else *reinterpret_cast<f32x4*>(a.out2 + m * KP + (n - E)) = acc2[mi][ni];
                } else {                                    // NCHW scatter of features dst_c0 .. dst_c0 + dst_c - 1
                    f32x4 v = acc2[mi][ni];
                    if constexpr (MODE == ENC_BWD) {
                        if (a.add2) v += *reinterpret_cast<const f32x4*>(a.add2 + m * KP + n);
                    }
                    const long long b = m / a.HW, p = m - b * a.HW;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = n + q - a.dst_c0;
                        if (c >= 0 && c < a.dst_c) {
                            float o = v[q];
                            if (a.add1) o += a.add1[b * a.add1_bs + (long long)c * a.HW + p];
                            a.dst[b * a.dst_bs + (long long)c * a.HW + p] = o;
                        }
                    }
                }
            }
        }
    }
}

template <int E, int MODE>
int io_launch(const IoDev& a_in, hipStream_t s) {
    constexpr bool GATHER = MODE == ENC_FWD || MODE == DEC_BWD;
    constexpr int K1 = GATHER ? KP : (MODE == DEC_FWD ? E + KP : E);
    const size_t lds = (size_t)ROWS * (K1 + E) * sizeof(__bf16);
    auto kern = sfno_io_kernel<E, MODE>;
    if (int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(kern), lds, "sfno_io")) return rc;
    IoDev a = a_in;
    a.rot = dlwp_tune_or("CHAIN_ROT", 1);
    hipLaunchKernelGGL(kern, dim3(ceil_div(a.T, ROWS)), dim3(512), lds, s, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

template <int MODE>
int io_dispatch(const IoDev& a, int E, hipStream_t s) {
    if (E == 256) return io_launch<256, MODE>(a, s);
    if (E == 128) return io_launch<128, MODE>(a, s);
    if (E == 64) return io_launch<64, MODE>(a, s);
    dlwp_set_error("sfno_io: no kernel for embed_dim %d (dlwp_sfno_io_supported)", E);
    return DLWP_E_UNSUPPORTED;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

__device__ __attribute__((aligned(16))) float g_io_zero_bias[512];
const float* zero_bias() {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_io_zero_bias)) != hipSuccess) return nullptr;
    return static_cast<const float*>(p);
}

// zero-padded fragment-order images: image i describes W'_i [rows_i][cols_i], W' = W or W^T, zero outside W''s real extent
struct PackPad { const float* W[8]; int ld[8], rr[8], rc[8], rows[8], cols[8], tr[8]; __bf16* img[8]; };
__global__ __launch_bounds__(256) void io_pack_kernel(PackPad a) {
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
    const int ld = a.ld[i], rr = a.rr[i], rc = a.rc[i];       // real rows / columns of W'
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int col = k0 + e;
        float x = 0.f;
        if (row < rr && col < rc) x = a.tr[i] ? W[(long long)col * ld + row] : W[(long long)row * ld + col];
        v[e] = (__bf16)x;
    }
    *reinterpret_cast<bf16x8*>(a.img[i] + idx * 8) = v;
}

}  // namespace

extern "C" int dlwp_sfno_io_supported(int E, int in_chans, int out_chans) {
    return (E == 256 || E == 128 || E == 64) && in_chans >= 1 && in_chans <= KP && out_chans >= 1 && out_chans <= NP;
}

extern "C" long long dlwp_sfno_io_image_elems(int E) { return (long long)(E + KP) * E; }

extern "C" int dlwp_sfno_io_pack(const float* enc_w1, const float* enc_w2, const float* dec_w1, const float* dec_w2, int E, int in_chans,
                                 int out_chans, int big_skip, void* images, void* stream) {
    DLWP_REQUIRE(enc_w1 && enc_w2 && dec_w1 && dec_w2 && images && aligned16(images), DLWP_E_INVALID, "sfno_io_pack: null / unaligned pointer");
    DLWP_REQUIRE(dlwp_sfno_io_supported(E, in_chans, out_chans), DLWP_E_UNSUPPORTED, "sfno_io_pack: embed %d, %d -> %d channels unsupported",
                 E, in_chans, out_chans);
    const int dk = E + (big_skip ? in_chans : 0);              // real width of the decoder's first layer
    PackPad a{};
    //                 enc fwd 1   enc fwd 2  enc bwd 1  enc bwd 2   dec fwd 1  dec fwd 2   dec bwd 1   dec bwd 2
    const float* W[8] = {enc_w1,    enc_w2,    enc_w2,    enc_w1,     dec_w1,    dec_w2,     dec_w2,     dec_w1};
    const int ld[8] =   {in_chans,  E,         E,         in_chans,   dk,        E,          E,          dk};
    const int tr[8] =   {0,         0,         1,         1,          0,         0,          1,          1};
    const int rr[8] =   {E,         E,         E,         in_chans,   E,         out_chans,  E,          dk};
    const int rc[8] =   {in_chans,  E,         E,         E,          dk,        E,          out_chans,  E};
    const int rows[8] = {E,         E,         E,         KP,         E,         NP,         E,          E + KP};
    const int cols[8] = {KP,        E,         E,         E,          E + KP,    E,          KP,         E};
    long long most = 0;
    for (int i = 0; i < 8; ++i) {
        a.W[i] = W[i]; a.ld[i] = ld[i]; a.tr[i] = tr[i]; a.rr[i] = rr[i]; a.rc[i] = rc[i]; a.rows[i] = rows[i]; a.cols[i] = cols[i];
        a.img[i] = reinterpret_cast<__bf16*>(images) + i * dlwp_sfno_io_image_elems(E);
        most = std::max(most, (long long)(rows[i] / 16) * (cols[i] / 32) * 64);
    }
    hipLaunchKernelGGL(io_pack_kernel, dim3((unsigned)((most + 255) / 256), 8), dim3(256), 0, (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

static int io_common(const dlwp_sfno_io_args* p, const char* who) {
    DLWP_REQUIRE(p, DLWP_E_INVALID, "%s: null arguments", who);
    DLWP_REQUIRE(p->T > 0 && p->HW > 0 && p->T % p->HW == 0, DLWP_E_INVALID, "%s: %d tokens must be whole samples of %d", who, p->T, p->HW);
    DLWP_REQUIRE(p->w1_img && p->w2_img && aligned16(p->w1_img) && aligned16(p->w2_img), DLWP_E_INVALID, "%s: weight images", who);
    DLWP_REQUIRE(p->E == 256 || p->E == 128 || p->E == 64, DLWP_E_UNSUPPORTED, "%s: embed_dim %d has no kernel", who, p->E);
    return DLWP_OK;
}

extern "C" int dlwp_sfno_encode_fwd(const dlwp_sfno_io_args* p, void* stream) {
    if (int rc = io_common(p, "sfno_encode_fwd")) return rc;
    int cin = 0;
    for (int s = 0; s < 3; ++s) {
        DLWP_REQUIRE(p->src_c[s] >= 0 && (p->src_c[s] == 0 || p->src[s]), DLWP_E_INVALID, "sfno_encode_fwd: source %d", s);
        cin += p->src_c[s];
    }
    DLWP_REQUIRE(cin >= 1 && cin <= KP && p->tok_lp && p->z && p->h && p->tokens, DLWP_E_INVALID, "sfno_encode_fwd: %d input channels / null tensor", cin);
    DLWP_REQUIRE(aligned16(p->tok_lp) && aligned16(p->z) && aligned16(p->h) && aligned16(p->tokens) && aligned16(p->bias) && aligned16(p->pos),
                 DLWP_E_INVALID, "sfno_encode_fwd: tensors must be 16-byte aligned");
    IoDev a{};
    for (int s = 0; s < 3; ++s) { a.src[s] = p->src[s]; a.src_bs[s] = p->src_bs[s]; a.src_c[s] = p->src_c[s]; }
    a.HW = p->HW; a.T = p->T;
    a.in_lp = reinterpret_cast<__bf16*>(p->tok_lp);
    a.w1 = reinterpret_cast<const __bf16*>(p->w1_img); a.w2 = reinterpret_cast<const __bf16*>(p->w2_img);
    a.b1 = p->bias ? p->bias : zero_bias();
    DLWP_REQUIRE(a.b1, DLWP_E_HIP, "sfno_encode_fwd: hipGetSymbolAddress failed");
    a.z = reinterpret_cast<__bf16*>(p->z); a.a = reinterpret_cast<__bf16*>(p->h);
    a.pos = p->pos;
    a.out = p->tokens;
    return io_dispatch<ENC_FWD>(a, p->E, (hipStream_t)stream);
}

extern "C" int dlwp_sfno_encode_bwd(const dlwp_sfno_io_args* p, void* stream) {
    if (int rc = io_common(p, "sfno_encode_bwd")) return rc;
    DLWP_REQUIRE(p->tokens && p->z && p->h, DLWP_E_INVALID, "sfno_encode_bwd: null tensor");
    DLWP_REQUIRE(aligned16(p->tokens) && aligned16(p->tokens_lp) && aligned16(p->z) && aligned16(p->h) && aligned16(p->tok_grad),
                 DLWP_E_INVALID, "sfno_encode_bwd: tensors must be 16-byte aligned");
    DLWP_REQUIRE(!p->frame || (p->frame_c >= 1 && p->frame_c0 >= 0 && p->frame_c0 + p->frame_c <= KP), DLWP_E_INVALID,
                 "sfno_encode_bwd: frame channels %d .. +%d", p->frame_c0, p->frame_c);
    IoDev a{};
    a.HW = p->HW; a.T = p->T;
    a.in = p->tokens;                                          // g [T][E]
    a.in_lp = reinterpret_cast<__bf16*>(p->tokens_lp);
    a.w1 = reinterpret_cast<const __bf16*>(p->w1_img); a.w2 = reinterpret_cast<const __bf16*>(p->w2_img);
    a.zin = reinterpret_cast<const __bf16*>(p->z);
    a.a = reinterpret_cast<__bf16*>(p->h);                     // gh
    a.dst = p->frame; a.dst_bs = p->frame_bs; a.dst_c = p->frame ? p->frame_c : 0; a.dst_c0 = p->frame_c0;
    a.add1 = p->frame_add; a.add1_bs = p->frame_add_bs;
    a.add2 = p->tok_grad;
    return io_dispatch<ENC_BWD>(a, p->E, (hipStream_t)stream);
}

extern "C" int dlwp_sfno_decode_fwd(const dlwp_sfno_io_args* p, void* stream) {
    if (int rc = io_common(p, "sfno_decode_fwd")) return rc;
    DLWP_REQUIRE(p->tokens && p->tokens_lp && p->tok_lp && p->z && p->h && p->frame && p->frame_c >= 1 && p->frame_c <= NP, DLWP_E_INVALID,
                 "sfno_decode_fwd: null tensor / %d output channels", p->frame_c);
    DLWP_REQUIRE(aligned16(p->tokens) && aligned16(p->tokens_lp) && aligned16(p->tok_lp) && aligned16(p->z) && aligned16(p->h) && aligned16(p->bias),
                 DLWP_E_INVALID, "sfno_decode_fwd: tensors must be 16-byte aligned");
    IoDev a{};
    a.HW = p->HW; a.T = p->T;
    a.in = p->tokens;
    a.in2 = reinterpret_cast<const __bf16*>(p->tok_lp);
    a.in_lp = reinterpret_cast<__bf16*>(p->tokens_lp);
    a.w1 = reinterpret_cast<const __bf16*>(p->w1_img); a.w2 = reinterpret_cast<const __bf16*>(p->w2_img);
    a.b1 = p->bias ? p->bias : zero_bias();
    DLWP_REQUIRE(a.b1, DLWP_E_HIP, "sfno_decode_fwd: hipGetSymbolAddress failed");
    a.z = reinterpret_cast<__bf16*>(p->z); a.a = reinterpret_cast<__bf16*>(p->h);
    a.dst = p->frame; a.dst_bs = p->frame_bs; a.dst_c = p->frame_c; a.dst_c0 = 0;
    a.add1 = p->frame_add; a.add1_bs = p->frame_add_bs;
    return io_dispatch<DEC_FWD>(a, p->E, (hipStream_t)stream);
}

extern "C" int dlwp_sfno_decode_bwd(const dlwp_sfno_io_args* p, void* stream) {
    if (int rc = io_common(p, "sfno_decode_bwd")) return rc;
    DLWP_REQUIRE(p->src[0] && p->src_c[0] >= 1 && p->src_c[0] <= NP && p->tok_lp && p->z && p->h && p->tokens && p->tok_grad, DLWP_E_INVALID,
                 "sfno_decode_bwd: null tensor / %d output channels", p->src_c[0]);
    DLWP_REQUIRE(aligned16(p->tok_lp) && aligned16(p->z) && aligned16(p->h) && aligned16(p->tokens) && aligned16(p->tok_grad), DLWP_E_INVALID,
                 "sfno_decode_bwd: tensors must be 16-byte aligned");
    IoDev a{};
    a.src[0] = p->src[0]; a.src_bs[0] = p->src_bs[0]; a.src_c[0] = p->src_c[0];      // g_out planes
    a.HW = p->HW; a.T = p->T;
    a.in_lp = reinterpret_cast<__bf16*>(p->tok_lp);            // bf16 copy of the gathered g_out rows [T][KP]
    a.w1 = reinterpret_cast<const __bf16*>(p->w1_img); a.w2 = reinterpret_cast<const __bf16*>(p->w2_img);
    a.zin = reinterpret_cast<const __bf16*>(p->z);
    a.a = reinterpret_cast<__bf16*>(p->h);                     // gh
    a.out = p->tokens;                                         // g_t
    a.out2 = p->tok_grad;                                      // g_tok
    return io_dispatch<DEC_BWD>(a, p->E, (hipStream_t)stream);
}

// Token MLP  y = fc2(GELU(fc1 x)) (+ residual)  and its input-gradient chain as ONE launch per direction for WIDE hidden layers
// (round 4): the hidden activation never exists as a GEMM operand in HBM.
//
// Reference: Mlp.forward of the AFNO / Swin / Pangu blocks (/root/reference/src/nsbench/models/fourcastnet/fourcastnet.py:40-56,
// swintransformer/swin_transformer.py:25-48; dlwpbench twins): fc1 -> GELU -> fc2, hidden = mlp_ratio * dim.
//   forward   z = x W1^T + b1,  h = GELU(z),  y = h W2^T + b2 (+ residual)          x, y [T, E];  z, h [T, Hd]
//   backward  gh = (g W2) * GELU'(z),  gx = gh W1                                    (gW1 = gh^T x, gW2 = g^T h stay GEMMs)
// As two GEMM launches the FourCastNet-scale layer (T = 16200, E = 768, Hd = 3072) costs 144 + 94 us forward: the first
// product's GELU epilogue has as many VALU issue slots as the product has MFMA slots, and the [T, Hd] hidden tensor is written
// and read back as an operand.  Here a workgroup (8 waves) owns 64 tokens and walks the hidden layer in chunks of 128:
//   stage A   z_j [64 x 128] = x . W1[chunk j]^T     (wave w: feature tiles w, w + 4, all four token tiles; K = E)
//   epilogue  + bias, GELU (backward: * GELU'(z_j)), the bf16 copies z_j / h_j (gh_j) that the other pass and the weight
//             gradients need leave from registers, h_j also goes to a double-buffered LDS image
//   stage B   y [64 x E] += h_j . W2[:, chunk j]^T   (wave w: feature tiles w, w + 4, ...; K = 128) in registers
// with ONE workgroup barrier per chunk.  Four waves, one per SIMD, 512 registers each: 192 output accumulators + a four-slot
// ring of six-fragment weight groups (18 KB in flight per wave) streamed L2 -> registers from fragment-order images
// (dlwp_mlp_chain_pack); products are transposed (weights = A operand, tokens = B operand) as in csrc/mlp_chain.hip.  Cost model per workgroup at E = 768: 9.4 MB of weights (the L2 -> CU stream is the roof:
// 253 workgroups x 9.4 MB = 2.4 GB per launch), 4608 MFMAs per wave (61 us of matrix pipe at two waves per SIMD).
#include <algorithm>
#include <cstdlib>
#include "chain_frag.hip.h"
#include "dlwpmi_internal.h"

namespace {

using namespace chainfrag;

constexpr int MT = 4, ROWS = 16 * MT;        // tokens per workgroup
constexpr int HC = 128;                      // hidden features per chunk
constexpr int NW = 4;                        // waves per workgroup: one per SIMD, 512 registers each (accumulators 224, weight ring 96, fragments)
constexpr int GF = 6, RING = 4;              // fragments per ring group, groups in the ring: 18 KB of weights in flight per wave (an
                                             // eight-slot ring spills: 248 bytes of scratch per lane)

struct StreamDev {
    const void* x;               // forward: x [T][E] (bf16 or fp32); backward: g [T][E] fp32
    __bf16* x_lp;                // bf16 copy of an fp32 input (the weight-gradient operand); nullable
    const __bf16 *wa, *wb;       // stage-A image (matrix [Hd][E]) and stage-B image (matrix [E][Hd])
    const float *ba, *bb;        // forward: b1 [Hd], b2 [E] (zeros when absent)
    const __bf16* zin;           // backward: z [T][Hd]
    __bf16 *z, *h;               // forward: z, h [T][Hd]; backward: h = gh
    const float* res;            // forward: residual [T][E] fp32, nullable
    void* y;                     // [T][E]: fp32, or bf16 when y_bf16
    int T, Hd, x_bf16, y_bf16, rot;
};

struct Group { bf16x8 f[GF]; };

template <int E, bool BWD>
__global__ __launch_bounds__(64 * NW) void mlp_stream_kernel(StreamDev a) {
    constexpr int KSA = E / 32;              // k-steps of stage A
    constexpr int NGA = KSA / GF;            // ring groups per stage-A feature tile
    constexpr int TA = HC / 16 / NW;         // stage-A feature tiles per wave and chunk
    constexpr int NTB = E / 16 / NW;         // stage-B feature tiles per wave
    constexpr int PB = NTB / GF;             // ring groups per stage-B k-step
    constexpr int KSB = HC / 32;             // k-steps of stage B per chunk
    static_assert(KSA % GF == 0 && NTB % GF == 0 && HC % (16 * NW) == 0, "stage shapes must tile into ring groups");
    constexpr int NGAT = TA * NGA, NG = NGAT + KSB * PB;        // ring groups per chunk
    static_assert(NG % RING == 0, "the ring slot of a group must not depend on the chunk");
    extern __shared__ __attribute__((aligned(16))) float st_smem[];
    __bf16* ximg = reinterpret_cast<__bf16*>(st_smem);          // [ROWS][E]
    __bf16* himg = ximg + ROWS * E;                             // [2][ROWS][HC]
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
    const int w = ((tid >> 6) + (a.rot ? (int)(blockIdx.x >> 3) : 0)) % NW;
    const int m0 = blockIdx.x * ROWS, nchunk = a.Hd / HC, KSBT = a.Hd / 32;       // KSBT: k-steps of the whole stage-B matrix

    // ring group `gi` of chunk `j`: stage A walks the k-steps of tiles 8 j + w + NW ta of the A image six at a time, stage B takes
    // k-step 4 j + kk of six of this wave's tiles w + NW i of the B image
    auto request = [&](Group& gq, int j, int gi) {
        if (gi < NGAT) {
            const int ta = gi / NGA, ga = gi - ta * NGA;
            const long long f0 = (long long)(8 * j + w + NW * ta) * KSA + ga * GF;
#pragma unroll
            for (int i = 0; i < GF; ++i) gq.f[i] = *reinterpret_cast<const bf16x8*>(a.wa + ((f0 + i) * 64 + lane) * 8);
        } else {
            const int gb = gi - NGAT, kk = KSB * j + gb / PB, part = gb % PB;
#pragma unroll
            for (int i = 0; i < GF; ++i)
                gq.f[i] = *reinterpret_cast<const bf16x8*>(a.wb + (((long long)(w + NW * (part * GF + i)) * KSBT + kk) * 64 + lane) * 8);
        }
    };
    Group ring[RING];
    // ---- input tile -> bf16 image
    if (a.x_bf16) {
        const __bf16* X = static_cast<const __bf16*>(a.x);
        constexpr int CPR = E / 8;
        for (int u = tid; u < ROWS * CPR; u += 64 * NW) {
            const int row = u / CPR, c = u - row * CPR;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(X + (long long)min(m0 + row, a.T - 1) * E + 8 * c);
            *reinterpret_cast<bf16x8*>(ximg + row * E + 8 * (c ^ (row & cmask<E>(c)))) = v;
        }
    } else {
        const float* X = static_cast<const float*>(a.x);
        for (int u = tid; u < ROWS * E / 4; u += 64 * NW) {
            const int row = u / (E / 4), k = 4 * (u - row * (E / 4));
            const float4 v4 = *reinterpret_cast<const float4*>(X + (long long)min(m0 + row, a.T - 1) * E + k);
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
            const bf16x4 b = to_bf4(v);
            img_store<E>(ximg, row, k, b);
            if (a.x_lp && m0 + row < a.T) *reinterpret_cast<bf16x4*>(a.x_lp + (long long)(m0 + row) * E + k) = b;
        }
    }
    // the first RING - 1 groups travel while the image lands
#pragma unroll
    for (int gi = 0; gi < RING - 1; ++gi) request(ring[gi], gi / NG, gi % NG);
    f32x4 acc[MT][NTB];
    zero_acc<MT, NTB>(acc);
    lds_barrier();

    for (int j = 0; j < nchunk; ++j) {
        __bf16* hb = himg + (j & 1) * ROWS * HC;
        f32x4 za[MT][TA];
        zero_acc<MT, TA>(za);
        f32x4 bias[TA];
        bf16x4 zv[MT][TA];
#pragma unroll
        for (int ta = 0; ta < TA; ++ta) {
            const int na = HC * j + 16 * (w + NW * ta) + 4 * g;       // this lane's four hidden features of tile ta
            if (BWD) {
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
                    zv[mi][ta] = *reinterpret_cast<const bf16x4*>(a.zin + (long long)min(m0 + 16 * mi + r, a.T - 1) * a.Hd + na);
            } else {
                bias[ta] = *reinterpret_cast<const f32x4*>(a.ba + na);
            }
        }
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            // keep RING - 1 groups in flight: the group RING - 1 ahead of this one goes into the slot consumed last
            {
                // (unconditional: a branch around loads makes the compiler wait vmcnt(0) at the join; past the last chunk the last
                // chunk's group is fetched again and never used)
                const int nx = gi + RING - 1, jn = min(j + nx / NG, nchunk - 1), gn = nx % NG;
                request(ring[nx % RING], jn, gn);
            }
            const Group& cur = ring[gi % RING];
            __builtin_amdgcn_sched_barrier(0);                        // a group's LDS reads stay inside the group (register pressure)
            if (gi < NGAT) {                                           // stage A: six k-steps of feature tile ta
                const int ta = gi / NGA, ga = gi - ta * NGA;
                // token fragments of k-step i + 1 are read while the MFMAs of k-step i run (one wave per SIMD: nobody else hides the
                // LDS latency)
                bf16x8 tf[GF + 1][MT];
                auto xfrag = [&](bf16x8 (&t)[MT], int kk) {
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi) {
                        const int row = 16 * mi + r, c0 = 4 * kk + g, c = c0 ^ (row & cmask<E>(c0));
                        t[mi] = *reinterpret_cast<const bf16x8*>(ximg + row * E + 8 * c);
                    }
                };
                xfrag(tf[0], ga * GF);
                xfrag(tf[1], ga * GF + 1);
#pragma unroll
                for (int i = 0; i < GF; ++i) {
                    if (i + 2 < GF) xfrag(tf[i + 2], ga * GF + i + 2);
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi)
                        za[mi][ta] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.f[i], tf[i][mi], za[mi][ta], 0, 0, 0);
                }
                if (gi == NGAT - 1) {                                  // epilogue A, then the chunk's one barrier
#pragma unroll
                    for (int tb = 0; tb < TA; ++tb) {
                        const int nl = 16 * (w + NW * tb) + 4 * g, na = HC * j + nl;
#pragma unroll
                        for (int mi = 0; mi < MT; ++mi) {
                            const int row = 16 * mi + r;
                            const long long m = m0 + row;
                            float v[4], act[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                if (BWD) {
                                    act[q] = za[mi][tb][q] * gelu_grad_f((float)zv[mi][tb][q]);
                                } else {
                                    v[q] = za[mi][tb][q] + bias[tb][q];
                                    act[q] = gelu_f(v[q]);
                                }
                            }
                            const bf16x4 ab = to_bf4(act);
                            img_store<HC>(hb, row, nl, ab);
                            if (m < a.T) {
                                *reinterpret_cast<bf16x4*>(a.h + m * a.Hd + na) = ab;
                                if (!BWD) *reinterpret_cast<bf16x4*>(a.z + m * a.Hd + na) = to_bf4(v);
                            }
                        }
                    }
                    lds_barrier();
                }
            } else {                                                   // stage B: one k-step of the chunk, six of this wave's tiles
                const int gb = gi - NGAT, kk = gb / PB, part = gb % PB;
                bf16x8 tf[MT];
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    const int row = 16 * mi + r, c0 = 4 * kk + g, c = c0 ^ (row & cmask<HC>(c0));
                    tf[mi] = *reinterpret_cast<const bf16x8*>(hb + row * HC + 8 * c);
                }
#pragma unroll
                for (int i = 0; i < GF; ++i)
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi)
                        acc[mi][part * GF + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.f[i], tf[mi], acc[mi][part * GF + i], 0, 0, 0);
            }
        }
    }
    // ---- output: features 16 (w + NW i) + 4 g .. + 3 of token 16 mi + r
#pragma unroll
    for (int i = 0; i < NTB; ++i) {
        const int n = 16 * (w + NW * i) + 4 * g;
        f32x4 bb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!BWD) bb = *reinterpret_cast<const f32x4*>(a.bb + n);
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const long long m = m0 + 16 * mi + r;
            if (m >= a.T) continue;
            f32x4 v = acc[mi][i] + bb;
            if (!BWD && a.res) v += *reinterpret_cast<const f32x4*>(a.res + m * E + n);
            if (a.y_bf16) {
                const float vv[4] = {v[0], v[1], v[2], v[3]};
                *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(a.y) + m * E + n) = to_bf4(vv);
            } else {
                *reinterpret_cast<f32x4*>(static_cast<float*>(a.y) + m * E + n) = v;
            }
        }
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

__device__ __attribute__((aligned(16))) float g_stream_zero_bias[8192];
const float* zero_bias() {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_stream_zero_bias)) != hipSuccess) return nullptr;
    return static_cast<const float*>(p);
}

template <bool BWD>
int stream_launch(const StreamDev& a_in, int E, hipStream_t s) {
    StreamDev a = a_in;
    a.rot = dlwp_tune_or("CHAIN_ROT", 1);
    const dim3 grid(ceil_div(a.T, ROWS));
    if (E == 768) {
        const size_t lds = (size_t)ROWS * (768 + 2 * HC) * sizeof(__bf16);
        auto kern = mlp_stream_kernel<768, BWD>;
        if (int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(kern), lds, "mlp_stream")) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, s, a);
    } else if (E == 384) {
        const size_t lds = (size_t)ROWS * (384 + 2 * HC) * sizeof(__bf16);
        auto kern = mlp_stream_kernel<384, BWD>;
        if (int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(kern), lds, "mlp_stream")) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, s, a);
    } else {
        dlwp_set_error("mlp_stream: no kernel for width %d (dlwp_mlp_stream_supported)", E);
        return DLWP_E_UNSUPPORTED;
    }
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

}  // namespace

extern "C" int dlwp_mlp_stream_supported(int E, int hidden) {
    return (E == 768 || E == 384) && hidden >= HC && hidden % HC == 0 && hidden <= 8192;
}

// the four images of one MLP in one launch: forward W1 [Hd][E], W2 [E][Hd]; backward W2^T [Hd][E], W1^T [E][Hd]
struct PackFour { const float* W[4]; int ld[4], rows[4], cols[4], tr[4]; __bf16* img[4]; };
__global__ __launch_bounds__(256) void stream_pack_kernel(PackFour a) {
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

extern "C" int dlwp_mlp_stream_pack(const float* w1, const float* w2, int E, int hidden, void* images, void* stream) {
    DLWP_REQUIRE(w1 && w2 && images && aligned16(images), DLWP_E_INVALID, "mlp_stream_pack: null / unaligned pointer");
    DLWP_REQUIRE(dlwp_mlp_stream_supported(E, hidden), DLWP_E_UNSUPPORTED, "mlp_stream_pack: widths %d -> %d unsupported", E, hidden);
    PackFour a{};
    const float* W[4] = {w1, w2, w2, w1};
    const int rows[4] = {hidden, E, hidden, E}, cols[4] = {E, hidden, E, hidden}, tr[4] = {0, 0, 1, 1};
    for (int i = 0; i < 4; ++i) {
        a.W[i] = W[i]; a.rows[i] = rows[i]; a.cols[i] = cols[i]; a.tr[i] = tr[i];
        a.ld[i] = tr[i] ? rows[i] : cols[i];
        a.img[i] = static_cast<__bf16*>(images) + (long long)i * E * hidden;
    }
    const long long per = (long long)(E / 16) * (hidden / 32) * 64;          // fragments x lanes of one image (all four have E * hidden elements)
    hipLaunchKernelGGL(stream_pack_kernel, dim3((unsigned)((per + 255) / 256), 4), dim3(256), 0, (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_mlp_stream_fwd(const void* x, int x_bf16, void* x_lp, const void* w1_img, const float* b1, const void* w2_img, const float* b2,
                                   const float* residual, void* z, void* h, float* y, int T, int E, int hidden, void* stream) {
    DLWP_REQUIRE(x && w1_img && w2_img && z && h && y && T > 0, DLWP_E_INVALID, "mlp_stream_fwd: null pointer / empty batch");
    DLWP_REQUIRE(dlwp_mlp_stream_supported(E, hidden), DLWP_E_UNSUPPORTED, "mlp_stream_fwd: widths %d -> %d unsupported", E, hidden);
    const void* ptrs[] = {x, x_lp, w1_img, b1, w2_img, b2, residual, z, h, y};
    for (const void* q : ptrs) DLWP_REQUIRE(aligned16(q), DLWP_E_INVALID, "mlp_stream_fwd: tensors must be 16-byte aligned");
    const float* zb = zero_bias();
    DLWP_REQUIRE(zb, DLWP_E_HIP, "mlp_stream_fwd: hipGetSymbolAddress failed");
    StreamDev a{};
    a.x = x; a.x_lp = static_cast<__bf16*>(x_lp);
    a.wa = static_cast<const __bf16*>(w1_img); a.wb = static_cast<const __bf16*>(w2_img);
    a.ba = b1 ? b1 : zb; a.bb = b2 ? b2 : zb;
    a.z = static_cast<__bf16*>(z); a.h = static_cast<__bf16*>(h);
    a.res = residual; a.y = y;
    a.T = T; a.Hd = hidden; a.x_bf16 = x_bf16 != 0; a.y_bf16 = 0;
    return stream_launch<false>(a, E, (hipStream_t)stream);
}

extern "C" int dlwp_mlp_stream_bwd(const float* g, void* g_lp, const void* w2t_img, const void* w1t_img, const void* z, void* gh, void* gx,
                                   int gx_bf16, int T, int E, int hidden, void* stream) {
    DLWP_REQUIRE(g && w2t_img && w1t_img && z && gh && gx && T > 0, DLWP_E_INVALID, "mlp_stream_bwd: null pointer / empty batch");
    DLWP_REQUIRE(dlwp_mlp_stream_supported(E, hidden), DLWP_E_UNSUPPORTED, "mlp_stream_bwd: widths %d -> %d unsupported", E, hidden);
    const void* ptrs[] = {g, g_lp, w2t_img, w1t_img, z, gh, gx};
    for (const void* q : ptrs) DLWP_REQUIRE(aligned16(q), DLWP_E_INVALID, "mlp_stream_bwd: tensors must be 16-byte aligned");
    StreamDev a{};
    a.x = g; a.x_lp = static_cast<__bf16*>(g_lp);
    a.wa = static_cast<const __bf16*>(w2t_img); a.wb = static_cast<const __bf16*>(w1t_img);
    a.zin = static_cast<const __bf16*>(z);
    a.h = static_cast<__bf16*>(gh);
    a.y = gx;
    a.T = T; a.Hd = hidden; a.x_bf16 = 0; a.y_bf16 = gx_bf16 != 0;
    return stream_launch<true>(a, E, (hipStream_t)stream);
}

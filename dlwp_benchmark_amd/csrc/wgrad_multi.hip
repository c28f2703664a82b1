// Weight gradients of token layers over SEVERAL applications of the same weights in one launch (round 4).
//
// Reference: torch.nn.Linear / 1x1-convolution backward, gW += g^T x and gb += sum_t g, as autograd accumulates it over the
// lead times of a rollout (the training loops at /root/reference/src/dlwpbench/scripts/train.py:116-141 back-propagate through
// every lead time of SFNO2DModule.forward, src/dlwpbench/models/fno/fno.py:217-259, so each layer's weight gradient is a sum of
// one product per lead time).
//
// The per-lead-time products of the SFNO block tail (8192 tokens deep, 256 x 256 ... 512 x 256 outputs) are latency-bound
// split-K launches: 29 us for three of them on the grouped register-staged kernel, 0.58 ms per C3 step.  A rollout applies a
// block's weights once per lead time, so the backward pass of a block can hand its (g, x) pairs to a list and let the LAST one
// run ONE product per weight over the concatenated token axis:  gW += sum_s g_s^T x_s  =  [g_0; g_1; ...]^T [x_0; x_1; ...].
// Here: up to four such products (one launch covers the three layers of a tail), each over up to eight segments of T tokens;
// all operands are bf16 arrays [T][cols] (the copies the forward / backward chain kernels leave, csrc/mlp_chain.hip).
// Kernel = the sliced LDS-DMA weight-gradient kernel of csrc/token_ops.hip (gemm_glds_tn_kernel: 128 x 128 tiles, both operand
// tiles [64 k][128] images filled by global_load_lds_dwordx4 with the chunk swizzle on the source address, fragments through
// ds_read_b64_tr_b16, two stages with the next step's DMAs in flight across raw barriers) with the K slices walking the
// segments; every slice writes its partial tile to a CALLER-PROVIDED slab with plain stores and a second launch adds the
// slices in order (bit-reproducible; no float atomics on the weight gradients).  The bias gradients (column sums of g) ride on
// the A fragments as in that kernel.
#include <algorithm>
#include "common.hip.h"
#include "dlwpmi_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int GT = 128, KD = 64, MAXP = DLWP_WGRAD_MAX_PRODUCTS, MAXS = DLWP_WGRAD_MAX_SEGMENTS;

struct WgDev {
    const __bf16* A[MAXP][MAXS];      // g segments [T][M]
    const __bf16* B[MAXP][MAXS];      // x segments [T][N]
    float* rowsum[MAXP];              // += sum_t g[t][m] (nullable)
    float* C[MAXP];                   // [M][N] fp32, accumulated into by the reduction launch
    int M[MAXP], N[MAXP], ntn[MAXP], overwrite[MAXP];
    int tile0[MAXP + 1];              // first tile of each product in the 1-D tile grid
    long long out_off[MAXP + 1];      // element offset of each product inside one slab slice
    float* slab;                      // [slices][out_off[nprod]]
    int nprod, T, kchunk, sps;        // sps = K slices per segment
    int nslices;                      // segments x sps
};

__global__ __launch_bounds__(256) void wgrad_multi_kernel(WgDev a) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    __bf16* lds = reinterpret_cast<__bf16*>(wsm);          // [2 stages][A | B][KD][128]
    constexpr int TILE = GT * KD, NI = KD / 16;
    const int lane = lane_id(), w = wave_id(), r = lane & 15, g = lane >> 4;
    // Speed only: the tiles of ONE K slice read the same token rows of up to six operand arrays (each 128-column piece by 2 - 4 of
    // them), so a slice's tiles run on one XCD (equal blockIdx % 8 under round-robin placement), back to back: the rows come from
    // that XCD's L2 after their first fetch.  Dealt over the XCDs (round 4) every tile fetched its own copy across the fabric:
    // 2.4 x the unique bytes at the C3 shapes.
    int tile_id, slice;
    {
        const int nt = a.tile0[a.nprod], id = blockIdx.x;
        if (a.nslices % 8 == 0) {
            const int xcd = id & 7, j = id >> 3;
            tile_id = j % nt;
            slice = (j / nt) * 8 + xcd;
        } else {
            tile_id = id % nt;
            slice = id / nt;
        }
    }
    int p = 0;
    while (p + 1 < a.nprod && tile_id >= a.tile0[p + 1]) ++p;
    const int local = tile_id - a.tile0[p], ntn = a.ntn[p];
    const int mt = local / ntn, nt_ = local - mt * ntn;
    const int m0 = mt * GT, n0 = nt_ * GT, M = a.M[p], N = a.N[p];
    const int seg = slice / a.sps, zz = slice - seg * a.sps;
    const int kbeg = zz * a.kchunk, kend = min(a.T, kbeg + a.kchunk);
    const __bf16* A = a.A[p][seg];
    const __bf16* B = a.B[p][seg];
    int krow[NI], acol[NI], bcol[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = (4 * i + w) * 64 + lane, kr = q >> 4, c = (q & 15) ^ (2 * (kr & 3));
        krow[i] = kr;
        acol[i] = min(m0 + 8 * c, M - 8);
        bcol[i] = min(n0 + 8 * c, N - 8);
    }
    auto issue = [&](int stage, int k0) {
        __bf16* As = lds + stage * 2 * TILE;
        __bf16* Bs = As + TILE;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const long long kk = min(k0 + krow[i], a.T - 1);
            lds_dma16(A + kk * M + acol[i], As + (4 * i + w) * 512);      // (inline assembly: see common.hip.h -- the builtin form made
            lds_dma16(B + kk * N + bcol[i], Bs + (4 * i + w) * 512);      // the compiler wait vmcnt(0) before this step's fragment reads)
        }
    };
    const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float rsum[4] = {0.f, 0.f, 0.f, 0.f};
    const bool has_rsum = a.rowsum[p] && !(w & 1);
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
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // this step's eight DMAs have landed, the next step's stay in flight
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
                const int kl = kend - (k0 + 32 * kk + 8 * g);        // this lane's fragment elements e >= kl lie past the slice
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (e >= kl) af[i][e] = (__bf16)0.f;
            }
            if (has_rsum && kt % ntn == nt_) {      // the column tiles of a row panel share the A tile: each takes every ntn-th step
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
            if (g == 0 && m < M) atomic_add_f32(&a.rowsum[p][m], v);
        }
    }
    float* out = a.slab + (long long)slice * a.out_off[a.nprod] + a.out_off[p];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + wm + i * 16 + 4 * g + q, n = n0 + wn + j * 16 + r;
                if (m < M && n < N) out[(long long)m * N + n] = acc[i][j][q];
            }
}

// C_p (+)= sum over the slices in a FIXED order (bit-reproducible): the 256 threads of a workgroup are 64 float4 units x 4 slice groups;
// group q adds the slices z = q, q + 4, ... in order, the four partial sums meet in LDS and are added 0 + 1 + 2 + 3.  (One thread per
// unit walking every slice -- round 4 -- was a serial chain of `slices` dependent loads on a grid of ~100 workgroups: 19 us for the 56
// slices x 0.44 MB of a Swin block, 12 us for a Pangu block's 8 x 7 MB.)
constexpr int RED_SG = 4;
__global__ __launch_bounds__(256) void wgrad_multi_reduce_kernel(WgDev a, int slices) {
    __shared__ f32x4 part[RED_SG - 1][64];
    const long long total4 = a.out_off[a.nprod] / 4, plane = a.out_off[a.nprod];
    const int u = threadIdx.x & 63, q = threadIdx.x >> 6;
    for (long long e0 = (long long)blockIdx.x * 64; e0 < total4; e0 += (long long)gridDim.x * 64) {      // workgroup-uniform trip count
        const long long e = e0 + u, off = 4 * e;
        const bool ok = e < total4;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok)
            for (int z = q; z < slices; z += RED_SG) v += *reinterpret_cast<const f32x4*>(a.slab + z * plane + off);
        if (q) part[q - 1][u] = v;
        __syncthreads();
        if (q == 0 && ok) {
#pragma unroll
            for (int i = 0; i < RED_SG - 1; ++i) v += part[i][u];
            int p = 0;
            while (p + 1 < a.nprod && off >= a.out_off[p + 1]) ++p;
            float* cp = a.C[p] + (off - a.out_off[p]);
            if (!a.overwrite[p]) v += *reinterpret_cast<const f32x4*>(cp);
            *reinterpret_cast<f32x4*>(cp) = v;
        }
        __syncthreads();
    }
}

int plan(const dlwp_wgrad_seg_product* p, int nprod, int nseg, int T, WgDev* dev, int* slices) {
    DLWP_REQUIRE(p && nprod >= 1 && nprod <= MAXP && nseg >= 1 && nseg <= MAXS && T >= 1, DLWP_E_INVALID,
                 "wgrad_segments: %d products (1..%d), %d segments (1..%d), %d tokens", nprod, MAXP, nseg, MAXS, T);
    WgDev a{};
    a.nprod = nprod;
    a.T = T;
    int tiles = 0;
    long long off = 0;
    for (int i = 0; i < nprod; ++i) {
        DLWP_REQUIRE(p[i].N >= 8 && p[i].K >= 8 && p[i].N % 8 == 0 && p[i].K % 8 == 0 && p[i].gw, DLWP_E_INVALID,
                     "wgrad_segments: product %d: widths %d x %d must be multiples of 8 (16-byte bf16 rows) and gw non-NULL", i, p[i].N, p[i].K);
        DLWP_REQUIRE((reinterpret_cast<uintptr_t>(p[i].gw) & 15) == 0, DLWP_E_INVALID, "wgrad_segments: gw must be 16-byte aligned");
        a.M[i] = p[i].N;
        a.N[i] = p[i].K;
        a.ntn[i] = ceil_div(p[i].K, GT);
        a.tile0[i] = tiles;
        a.out_off[i] = off;
        tiles += a.ntn[i] * ceil_div(p[i].N, GT);
        off += (long long)p[i].N * p[i].K;
        a.C[i] = p[i].gw;
        a.overwrite[i] = p[i].overwrite != 0;
        a.rowsum[i] = p[i].gb;
        for (int s = 0; s < nseg; ++s) {
            DLWP_REQUIRE(p[i].g[s] && p[i].x[s] && (reinterpret_cast<uintptr_t>(p[i].g[s]) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(p[i].x[s]) & 15) == 0, DLWP_E_INVALID,
                         "wgrad_segments: product %d segment %d: NULL or unaligned operand", i, s);
            a.A[i][s] = reinterpret_cast<const __bf16*>(p[i].g[s]);
            a.B[i][s] = reinterpret_cast<const __bf16*>(p[i].x[s]);
        }
    }
    a.tile0[nprod] = tiles;
    a.out_off[nprod] = off;
    // K slices per segment: the grid should fill the resident slots (two 64 KB workgroups per CU) once; at least eight K-steps per slice
    // (round 6, in the step: 384 against round 5's 512 -- Swin C4 395.5 -> 400.3 samples/s, Pangu C4 and SFNO C3 unchanged; 256 / 768 / 1024 lose 1.5 - 2.5 %)
    const int slots = dlwp_tune_or("WGRAD_MULTI_WGS", 384);
    int sps = std::max(1, std::min((slots + tiles * nseg / 2) / (tiles * nseg), std::max(1, T / (8 * KD))));
    {   // a slice count that is a multiple of 8 lets the kernel keep a slice's tiles on one XCD: round sps up where that costs little
        int q = 8;
        for (int d = 8; d >= 1; d >>= 1)
            if (nseg % d == 0) { q = 8 / d; break; }
        const int up = ceil_div(sps, q) * q;
        if (up <= 2 * sps && T / up >= 2 * KD) sps = up;
    }
    a.kchunk = ceil_div(ceil_div(T, sps), KD) * KD;
    sps = ceil_div(T, a.kchunk);
    a.sps = sps;
    *slices = nseg * sps;
    DLWP_REQUIRE(*slices <= 65535, DLWP_E_INVALID, "wgrad_segments: too many slices");
    *dev = a;
    return DLWP_OK;
}

}  // namespace

extern "C" size_t dlwp_wgrad_segments_workspace_bytes(const dlwp_wgrad_seg_product* p, int nprod, int nseg, int T) {
    WgDev a;
    int slices = 0;
    if (plan(p, nprod, nseg, T, &a, &slices) != DLWP_OK) return 0;
    return sizeof(float) * (size_t)slices * (size_t)a.out_off[nprod];
}

extern "C" int dlwp_wgrad_segments(const dlwp_wgrad_seg_product* p, int nprod, int nseg, int T, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    WgDev a;
    int slices = 0;
    if (int rc = plan(p, nprod, nseg, T, &a, &slices)) return rc;
    const size_t need = sizeof(float) * (size_t)slices * (size_t)a.out_off[nprod];
    DLWP_REQUIRE(workspace && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0, DLWP_E_INVALID,
                 "wgrad_segments: workspace of %zu bytes needed (dlwp_wgrad_segments_workspace_bytes), %zu given", need, workspace_bytes);
    a.slab = static_cast<float*>(workspace);
    const size_t lds = (size_t)2 * 2 * GT * KD * sizeof(__bf16);
    if (int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(wgrad_multi_kernel), lds, "wgrad_multi")) return rc;
    const hipStream_t s = (hipStream_t)stream;
    a.nslices = slices;
    double pflops = 0, pbytes = 0;         // live accounting: gW_i = g_i^T x_i over nseg x T tokens; operands bf16, read once
    for (int i = 0; i < nprod; ++i) {
        pflops += 2.0 * p[i].N * p[i].K * (double)T * nseg;
        pbytes += 2.0 * (p[i].N + p[i].K) * (double)T * nseg;
    }
    {
        dlwp_prof_scope prof(s, pflops, pbytes + 4.0 * slices * a.out_off[nprod], "wgrad_multi_kernel");
        hipLaunchKernelGGL(wgrad_multi_kernel, dim3(a.tile0[nprod] * slices), dim3(256), lds, s, a);
    }
    DLWP_LAUNCH_CHECK();
    const long long units = a.out_off[nprod] / 4;
    dlwp_prof_scope prof2(s, (double)slices * a.out_off[nprod], 4.0 * (slices + 1) * a.out_off[nprod], "wgrad_multi_reduce_kernel");
    hipLaunchKernelGGL(wgrad_multi_reduce_kernel, dim3((unsigned)std::min<long long>((units + 63) / 64, 4096)), dim3(256), 0, s, a, slices);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

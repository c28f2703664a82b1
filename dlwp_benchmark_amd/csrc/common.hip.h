// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of libdlwpmi.
// wave = 64 lanes everywhere; MFMA helpers use the exact-f32 v_mfma_f32_16x16x4_f32.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- host side error plumbing (thread-local last error, C-ABI returns negative codes)
#define DLWP_OK 0
#define DLWP_E_INVALID (-1)
#define DLWP_E_HIP (-2)
#define DLWP_E_UNSUPPORTED (-3)
#define DLWP_E_NOMEM (-4)

void dlwp_set_error(const char* fmt, ...);
// raise a kernel's dynamic-LDS limit once (cached per kernel, so steady-state launches and
// graph capture never call hipFuncSetAttribute)
int dlwp_ensure_lds(const void* kernel, size_t bytes, const char* what);
// Zero fills as ordinary kernels.  hipMemsetAsync / hipMemset2DAsync nodes captured into a hipGraph were seen to write a
// garbage pattern instead of zero on later replays once other allocations had happened in the process (ROCm 7.2,
// intermittent; DESIGN.md §4 "lessons"), so nothing on a capturable path uses the runtime's memset.
int dlwp_zero_f32(float* p, long long n, void* stream);
int dlwp_zero_2d_f32(float* p, long long ld, int rows, int cols, void* stream);

#define DLWP_HIP(call)                                                              \
    do {                                                                            \
        hipError_t e__ = (call);                                                    \
        if (e__ != hipSuccess) {                                                    \
            dlwp_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call,            \
                           hipGetErrorString(e__));                                 \
            return DLWP_E_HIP;                                                      \
        }                                                                           \
    } while (0)

#define DLWP_REQUIRE(cond, code, ...)                                               \
    do {                                                                            \
        if (!(cond)) {                                                              \
            dlwp_set_error(__VA_ARGS__);                                            \
            return (code);                                                          \
        }                                                                           \
    } while (0)

#define DLWP_LAUNCH_CHECK() DLWP_HIP(hipGetLastError())

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// ---- device helpers
#define WAVE 64

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// Workgroup barrier for hand-offs through LDS: waits for this wave's LDS traffic only.  __syncthreads() also waits for vmcnt(0), i.e. for
// every global load still in flight and every global STORE the wave has issued (a store round trip is ~1 us under load): kernels whose
// barriers only order LDS accesses use this one.  (Not for data exchanged through GLOBAL memory inside a workgroup.)
// DLWP_PLAIN_BARRIERS builds fall back to __syncthreads() (A/B measurements).
__device__ __forceinline__ void lds_barrier() {
#ifdef DLWP_PLAIN_BARRIERS
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// ---- LDS-DMA the compiler does not see (round 5).  hipcc knows that __builtin_amdgcn_global_load_lds writes LDS and, with no
// alias information for dynamic shared memory, puts `s_waitcnt vmcnt(0)` in front of the FIRST LDS read that follows one -- in
// every double-buffered kernel here that is the read of the OTHER buffer, so the "next step's DMAs in flight" of the counted
// vmcnt waits never were (ISA of wgrad_multi_kernel, dhconv_apply2_kernel, gemm_glds*: vmcnt(N) by hand, s_barrier, then the
// compiler's vmcnt(0)).  This form issues the same instruction from inline assembly: lane i's 16 bytes at `gptr` go to LDS byte
// address lds_base + 16 i (lds_base wave-uniform, M0 holds it).  The caller orders its LDS reads behind the data with an explicit
// `s_waitcnt vmcnt(N)` (+ barrier for other waves' pieces), as it already does.  The compiler's own vmcnt counts for register
// loads stay correct: vector-memory operations retire in order, so operations it does not know about can only make a wait
// longer, never shorter.
__device__ __forceinline__ void lds_dma16(const void* gptr, const void* lds_base) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)lds_base);
    unsigned keep;          // M0 is a reserved register for the compiler (not allowed on a clobber list): saved and restored here
    // (s_nop: an SALU write of M0 needs one wait state before an instruction that reads it; the hazard recogniser does not look
    // inside inline assembly)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(m0v), "v"(gptr)
                 : "memory");
}

// D[m=4g+j][n=r] += sum_k A[m=r][k=g] * B[k=g][n=r]   (r = lane&15, g = lane>>4, j = reg)
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// One "permuted-k" chunk of 16: lane (r,g) supplies A[m=r][k0+4g+s] in a[s] and
// B[k0+4g+s][n=r] in b[s]; the four MFMAs together sum k0..k0+15.
__device__ __forceinline__ f32x4 mfma16_chunk(const f32x4 a, const f32x4 b, f32x4 c) {
    c = mfma16(a[0], b[0], c);
    c = mfma16(a[1], b[1], c);
    c = mfma16(a[2], b[2], c);
    c = mfma16(a[3], b[3], c);
    return c;
}

// The same chunk with bf16 operands (fp32 accumulation): v_mfma_f32_16x16x16_bf16 takes exactly the fragment layout of the
// permuted-k chunk -- lane (r, g) supplies A[m=r][k=4g..4g+3] and B[k=4g..4g+3][n=r] -- so ONE instruction (8 cycles) replaces the
// four fp32 ones (4 x 32 cycles).  Used where the run asks for bf16 matrix arithmetic (dlwp_set_gemm_precision(1)): window
// attention's Q K^T, P V and their backward products; softmax statistics, bias and accumulators stay fp32.
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <bool BF>
__device__ __forceinline__ f32x4 mfma16_chunk_p(const f32x4 a, const f32x4 b, f32x4 c) {
    if constexpr (BF) {
        typedef __bf16 bh4 __attribute__((ext_vector_type(4)));
        const bh4 ah = {(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3]};
        const bh4 bh = {(__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, ah), __builtin_bit_cast(s16x4, bh), c, 0, 0, 0);
    } else {
        return mfma16_chunk(a, b, c);
    }
}

// Exact (erf) GELU and its derivative, evaluated together in ~17 VALU instructions.
// erfc via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, fp32 rounding level) with one v_rcp_f32
// and one v_exp_f32: libm erff costs ~100 issue slots per element and made the MLP kernels
// VALU-bound (profiles/ r01 notes).  Phi(z) = 0.5 + copysign(0.5 - q, z), q = 0.5*erfc(|z|/sqrt2)
// (no cancellation in the negative tail).
__device__ __forceinline__ void gelu_both(float z, float& act, float& dact) {
    const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(z), 0.3275911f * 0.70710678118654752440f, 1.0f));
    const float e = __builtin_amdgcn_exp2f(z * z * (-0.5f * 1.44269504088896340736f));  // exp(-z^2/2)
    // 0.5 * (a1 t + a2 t^2 + a3 t^3 + a4 t^4 + a5 t^5)
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f),
                                                0.5f * 1.421413741f), 0.5f * -0.284496736f), 0.5f * 0.254829592f);
    const float q = poly * e;
    const float cdf = 0.5f + copysignf(0.5f - q, z);
    act = z * cdf;
    dact = fmaf(z * e, 0.39894228040143267794f, cdf);
}
// The same on two elements at once: the polynomial and the products are written on 2-vectors so that they compile to the packed
// fp32 forms (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of arithmetic per issue slot); the reciprocal and the exponential
// stay one instruction per element.  Bit-identical to gelu_both per element (same operations in the same order).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_both2(f32x2 z, f32x2& act, f32x2& dact) {
    const f32x2 az = f32x2{fabsf(z[0]), fabsf(z[1])};
    const f32x2 u = __builtin_elementwise_fma(az, f32x2{0.3275911f * 0.70710678118654752440f, 0.3275911f * 0.70710678118654752440f}, f32x2{1.0f, 1.0f});
    const f32x2 t = f32x2{__builtin_amdgcn_rcpf(u[0]), __builtin_amdgcn_rcpf(u[1])};
    const f32x2 zz = z * z * f32x2{-0.5f * 1.44269504088896340736f, -0.5f * 1.44269504088896340736f};
    const f32x2 e = f32x2{__builtin_amdgcn_exp2f(zz[0]), __builtin_amdgcn_exp2f(zz[1])};
    auto c2 = [](float c) { return f32x2{c, c}; };
    f32x2 p = __builtin_elementwise_fma(t, c2(0.5f * 1.061405429f), c2(0.5f * -1.453152027f));
    p = __builtin_elementwise_fma(t, p, c2(0.5f * 1.421413741f));
    p = __builtin_elementwise_fma(t, p, c2(0.5f * -0.284496736f));
    p = __builtin_elementwise_fma(t, p, c2(0.5f * 0.254829592f));
    const f32x2 q = t * p * e;
    const f32x2 h = c2(0.5f) - q;
    const f32x2 cdf = c2(0.5f) + f32x2{copysignf(h[0], z[0]), copysignf(h[1], z[1])};
    act = z * cdf;
    dact = __builtin_elementwise_fma(z * e, c2(0.39894228040143267794f), cdf);
}
// four consecutive accumulator elements
__device__ __forceinline__ void gelu_both4(const f32x4 z, f32x4& act, f32x4& dact) {
    f32x2 a0, d0, a1, d1;
    gelu_both2(f32x2{z[0], z[1]}, a0, d0);
    gelu_both2(f32x2{z[2], z[3]}, a1, d1);
    act = f32x4{a0[0], a0[1], a1[0], a1[1]};
    dact = f32x4{d0[0], d0[1], d1[0], d1[1]};
}
__device__ __forceinline__ f32x4 gelu4(const f32x4 z) {
    f32x4 a, d;
    gelu_both4(z, a, d);
    return a;
}
__device__ __forceinline__ f32x4 gelu_grad4(const f32x4 z) {
    f32x4 a, d;
    gelu_both4(z, a, d);
    return d;
}
__device__ __forceinline__ float gelu_f(float z) {
    float a, d;
    gelu_both(z, a, d);
    return a;
}
__device__ __forceinline__ float gelu_grad_f(float z) {
    float a, d;
    gelu_both(z, a, d);
    return d;
}

// hardware float atomic add (no CAS loop)
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

// ---- staging helpers: batched (unrolled) 16-byte global loads -> LDS images with row padding
// exact e / d for e*d < 2^32 (all index spaces here are < 2^21): q = umulhi(e, ceil(2^32 / d))
struct FastDiv {
    unsigned mul, d;
};
static inline FastDiv make_fastdiv(int d) {
    FastDiv f;
    f.d = (unsigned)d;
    f.mul = d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d);
    return f;
}
__device__ __forceinline__ int fastdiv(int e, FastDiv f) { return f.d <= 1 ? e : (int)__umulhi((unsigned)e, f.mul); }

// dst[r*ld + c] = src[r*cols + c]  (TR: dst[c*ld + r]); src is a dense row-major [rows][cols] matrix.
template <bool TR>
__device__ __forceinline__ void stage_matrix(float* dst, int ld, const float* __restrict__ src, int rows, int cols,
                                             FastDiv dcols, bool vec_ok) {
    const int n = rows * cols;
    const int n4 = vec_ok ? (n >> 2) : 0;
#pragma unroll 4
    for (int u = threadIdx.x; u < n4; u += blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(src)[u];
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = 4 * u + k;
            const int rr = fastdiv(e, dcols);
            const int cc = e - rr * cols;
            dst[TR ? cc * ld + rr : rr * ld + cc] = vv[k];
        }
    }
#pragma unroll 4
    for (int e = 4 * n4 + threadIdx.x; e < n; e += blockDim.x) {
        const int rr = fastdiv(e, dcols);
        const int cc = e - rr * cols;
        dst[TR ? cc * ld + rr : rr * ld + cc] = src[e];
    }
}

// Split staging: issue() puts up to MAXU 16-byte loads per thread in flight (no LDS traffic yet);
// commit() scatters them into the padded LDS image.  Kernels issue ALL their independent inputs
// first and commit afterwards, so a workgroup pays one global-memory latency instead of one per
// input (in-kernel stamps: the serial staging phases were 40-60% of these short kernels).
template <int MAXU>
struct MatLoad {
    float4 v[MAXU];
    __device__ __forceinline__ void issue(const float* __restrict__ src, int n4, int base = 0) {
#pragma unroll
        for (int k = 0; k < MAXU; ++k) {
            const int u = base + threadIdx.x + k * blockDim.x;
            v[k] = u < n4 ? reinterpret_cast<const float4*>(src)[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    template <bool TR>
    __device__ __forceinline__ void commit(float* dst, int ld, int cols, FastDiv dcols, int n4, int base = 0) const {
#pragma unroll
        for (int k = 0; k < MAXU; ++k) {
            const int u = base + threadIdx.x + k * blockDim.x;
            if (u < n4) {
                const float vv[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = 4 * u + q;
                    const int rr = fastdiv(e, dcols);
                    const int cc = e - rr * cols;
                    dst[TR ? cc * ld + rr : rr * ld + cc] = vv[q];
                }
            }
        }
    }
};
// number of 16-byte units MatLoad handles for a dense [rows][cols] matrix (0 when not 16-byte loadable);
// the remainder (and the unaligned case) goes through stage_matrix_tail.
__device__ __forceinline__ int matload_units(int rows, int cols, bool vec_ok, int maxu) {
    const int n4 = vec_ok ? (rows * cols) >> 2 : 0;
    const int cap = maxu * (int)blockDim.x;
    return n4 < cap ? n4 : cap;
}
template <bool TR>
__device__ __forceinline__ void stage_matrix_tail(float* dst, int ld, const float* __restrict__ src, int rows, int cols,
                                                  FastDiv dcols, int done4) {
    const int n = rows * cols;
#pragma unroll 4
    for (int e = 4 * done4 + threadIdx.x; e < n; e += blockDim.x) {
        const int rr = fastdiv(e, dcols);
        const int cc = e - rr * cols;
        dst[TR ? cc * ld + rr : rr * ld + cc] = src[e];
    }
}

// zero the padding of an LDS image holding a [rows][cols] matrix inside [rows_pad][cols_pad] (row stride ld)
__device__ __forceinline__ void zero_padding(float* dst, int ld, int rows, int cols, int rows_pad, int cols_pad) {
    if (cols_pad > cols)
        for (int rr = threadIdx.x; rr < rows; rr += blockDim.x)
            for (int cc = cols; cc < cols_pad; ++cc) dst[rr * ld + cc] = 0.f;
    for (int rr = rows; rr < rows_pad; ++rr)
        for (int cc = threadIdx.x; cc < cols_pad; cc += blockDim.x) dst[rr * ld + cc] = 0.f;
}

// ---- in-kernel phase stamps (diagnostic build only: -DDLWP_STAMPS; never in the shipped library)
#ifdef DLWP_STAMPS
__device__ unsigned long long g_dlwp_stamps[32];
#define DLWP_STAMP(i)                                                                         \
    do {                                                                                      \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                            \
            unsigned long long t__;                                                           \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");      \
            g_dlwp_stamps[i] = t__;                                                           \
        }                                                                                     \
    } while (0)
// the same for a workgroup chosen by the caller (wg = a workgroup-uniform predicate): thread 0 of that workgroup records
#define DLWP_STAMP_IF(wg, i)                                                                  \
    do {                                                                                      \
        if ((wg) && threadIdx.x == 0) {                                                       \
            unsigned long long t__;                                                           \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");      \
            g_dlwp_stamps[i] = t__;                                                           \
        }                                                                                     \
    } while (0)
// span of a whole launch on the constant-rate 100 MHz clock (comparable across XCDs): earliest first instruction and
// latest last instruction over ALL workgroups (slots: begin = min, end = max)
__device__ unsigned long long g_dlwp_span[2] = {~0ull, 0ull};
#define DLWP_SPAN_BEGIN()                                                                     \
    do {                                                                                      \
        if (threadIdx.x == 0) {                                                               \
            unsigned long long t__;                                                           \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");  \
            atomicMin(&g_dlwp_span[0], t__);                                                  \
        }                                                                                     \
    } while (0)
#define DLWP_SPAN_END()                                                                       \
    do {                                                                                      \
        if ((threadIdx.x & 63) == 0) {                                                        \
            unsigned long long t__;                                                           \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");  \
            atomicMax(&g_dlwp_span[1], t__);                                                  \
        }                                                                                     \
    } while (0)
// per-wave variant: lane 0 of every wave of workgroup 0 records slot base + wave
#define DLWP_STAMP_WAVE(base)                                                                 \
    do {                                                                                      \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {                                     \
            unsigned long long t__;                                                           \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");      \
            g_dlwp_stamps[(base) + (threadIdx.x >> 6)] = t__;                                 \
        }                                                                                     \
    } while (0)
#else
#define DLWP_STAMP(i)
#define DLWP_STAMP_IF(wg, i)
#define DLWP_STAMP_WAVE(base)
#define DLWP_SPAN_BEGIN()
#define DLWP_SPAN_END()
#endif

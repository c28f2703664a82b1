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
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

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

__device__ __forceinline__ float gelu_f(float z) {
    return 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_grad_f(float z) {
    const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * z * z);
    return cdf + z * pdf;
}

// hardware float atomic add (no CAS loop)
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

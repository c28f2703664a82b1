// Does the exact-f32 MFMA (v_mfma_f32_16x16x4_f32) share the SIMD's issue / FMA lanes with ordinary VALU work, and does the bf16
// MFMA (v_mfma_f32_16x16x16_bf16_1k)?  One 512-thread workgroup per CU = two waves per SIMD: waves 0-3 run a matrix loop,
// waves 4-7 a VALU loop (independent fma chains); cycles of each alone and of both together.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu_coexec.hip -o tools/micro/mfma_valu_coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int KIND>   // 0: f32 MFMA, 1: bf16 MFMA
__global__ __launch_bounds__(512) void k(float* sink, unsigned long long* cyc, int n_mfma, int n_valu, int mode) {
    const int w = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_readcyclecounter();
    float out = 0.f;
    if (w < 4) {
        if (mode & 1) {
            f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
            float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
            s16x4 ah = {(short)threadIdx.x, 1, 2, 3}, bh = {4, 5, 6, (short)threadIdx.x};
            for (int i = 0; i < n_mfma; ++i) {
                if (KIND == 0) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
                } else {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, c3, 0, 0, 0);
                }
            }
            out = c0[0] + c1[1] + c2[2] + c3[3];
        }
    } else {
        if (mode & 2) {
            float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
            const float m = 1.0001f, a = 0.5f;
            for (int i = 0; i < n_valu; ++i) {
                x0 = fmaf(x0, m, a); x1 = fmaf(x1, m, a); x2 = fmaf(x2, m, a); x3 = fmaf(x3, m, a);
                x4 = fmaf(x4, m, a); x5 = fmaf(x5, m, a); x6 = fmaf(x6, m, a); x7 = fmaf(x7, m, a);
            }
            out = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    sink[blockIdx.x * 512 + threadIdx.x] = out;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float* sink; unsigned long long* cyc;
    hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&cyc, 8);
    const int n_mfma = 2000, n_valu = 4000;     // 8000 MFMAs per wave; 32000 fma per wave
    for (int kind = 0; kind < 2; ++kind)
        for (int mode = 1; mode <= 3; ++mode) {
            unsigned long long c = 0;
            for (int rep = 0; rep < 3; ++rep) {
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, sink, cyc, n_mfma, n_valu, mode);
                else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, sink, cyc, n_mfma, n_valu, mode);
                hipDeviceSynchronize();
                hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            }
            printf("%s  %s: %8llu cycles  (%d MFMAs per matrix wave = %.1f cycles each if alone; %d fma per VALU wave = %.2f cycles each if alone)\n",
                   kind == 0 ? "v_mfma_f32_16x16x4_f32     " : "v_mfma_f32_16x16x16_bf16_1k",
                   mode == 1 ? "matrix waves only" : mode == 2 ? "VALU waves only  " : "both             ", c, 4 * n_mfma, (double)c / (4 * n_mfma),
                   8 * n_valu, (double)c / (8 * n_valu));
        }
    return 0;
}

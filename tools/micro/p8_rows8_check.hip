// Standalone check of the cross-lane transposition used by gemm_p8_kernel's wide epilogue (csrc/token_ops.hip, p8_rows8):
//   hipcc --offload-arch=gfx950 -O3 tools/micro/p8_rows8_check.hip -o tools/micro/p8_rows8_check && tools/micro/p8_rows8_check
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
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
__global__ void k(float* out) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int J = 0; J < 4; ++J) {
        const float b = (float)(r * 1000 + 16 * J + 4 * g);
        acc[J] = f32x4{b, b + 1.f, b + 2.f, b + 3.f};
    }
    f32x4 ya[2], yb[2];
    p8_rows8(acc, ya, yb);
#define OUT(SEL, VEC, HALF) out[(SEL * 8 + 4 * HALF + 0) * 64 + lane] = VEC[HALF][0]; out[(SEL * 8 + 4 * HALF + 1) * 64 + lane] = VEC[HALF][1]; \
                     out[(SEL * 8 + 4 * HALF + 2) * 64 + lane] = VEC[HALF][2]; out[(SEL * 8 + 4 * HALF + 3) * 64 + lane] = VEC[HALF][3]
    OUT(0, ya, 0); OUT(0, ya, 1); OUT(1, yb, 0); OUT(1, yb, 1);
}
int main() {
    float* d;
    hipMalloc(&d, sizeof(float) * 16 * 64);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[16 * 64];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int r = lane & 15, g = lane >> 4, col = 32 * (r >> 3) + 16 * (g & 1) + 4 * (g & ~1);
        for (int y = 0; y < 2; ++y)
            for (int e = 0; e < 8; ++e) {
                const int row = (y ? 8 : 0) + (r & 7), want = row * 1000 + col + e, got = (int)h[(y * 8 + e) * 64 + lane];
                if (want != got) {
                    if (bad < 16) printf("lane %d (r %d g %d) y%c[%d]: want %d got %d\n", lane, r, g, y ? 'b' : 'a', e, want, got);
                    ++bad;
                }
            }
    }
    printf("p8_rows8: %d mismatches\n", bad);
    return bad != 0;
}

// Semantics check of ds_read_b64_tr_b16 (builtin __builtin_amdgcn_ds_read_tr16_b64_v4bf16) on gfx950:
// per 16-lane group, lane 4q+p passes the address of block row q, columns 4p..4p+3; lane i receives column i of the four rows.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/tr_read.hip -o tools/micro/tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    __shared__ __attribute__((aligned(16))) __bf16 S[64 * 72];
    for (int i = threadIdx.x; i < 64 * 72; i += 64) S[i] = (__bf16)(float)((i / 72) * 16 + (i % 72) % 16);   // value = 16*row + col%16 (exact in bf16 for row < 16)
    __syncthreads();
    const int lane = threadIdx.x, grp = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    auto* ptr = (__attribute__((address_space(3))) bf16x4*)(&S[(4 * grp + q) * 72 + 4 * p]);
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(ptr);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (float)v[e];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4); float h[256];
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 4; ++e) {
        const int grp = lane >> 4, i = lane & 15;
        const float expect = (4 * grp + e) * 16 + i;      // row 4 grp + e, column i
        if (h[lane * 4 + e] != expect) { if (bad < 8) printf("lane %d elem %d: got %g expect %g\n", lane, e, h[lane * 4 + e], expect); ++bad; }
    }
    printf("ds_read_b64_tr_b16 semantics %s (%d mismatches)\n", bad ? "DIFFER" : "as documented", bad);
    return bad != 0;
}

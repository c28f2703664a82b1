// Microbenchmark: LDS atomic-add throughput on gfx950 (float vs u32 vs u64), random and strided addresses.
// Build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/micro/lds_atomics.hip -o tools/micro/lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>   // 0 f32, 1 u32, 2 u64, 3 plain store, 4 read-modify-write (non atomic)
__global__ __launch_bounds__(256) void k(const int* __restrict__ idx, unsigned long long* cycles, float* sink, int iters) {
    __shared__ unsigned long long buf[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = 0;
    int my[16];
    for (int u = 0; u < 16; ++u) my[u] = idx[threadIdx.x * 16 + u];
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int a = (my[u] + it) & 4095;
            if (MODE == 0) atomicAdd(reinterpret_cast<float*>(buf) + a, 1.5f);
            if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(buf) + a, 3u);
            if (MODE == 2) atomicAdd(buf + a, 3ull);
            if (MODE == 3) reinterpret_cast<float*>(buf)[a] = 1.5f;
            if (MODE == 4) reinterpret_cast<float*>(buf)[a] += 1.5f;
        }
    }
    __syncthreads();
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = reinterpret_cast<float*>(buf)[threadIdx.x];
}

int main() {
    int* h = (int*)malloc(4096 * sizeof(int));
    int *d; unsigned long long* c; float* s;
    hipMalloc(&d, 4096 * sizeof(int)); hipMalloc(&c, 8); hipMalloc(&s, 256 * 256 * 4);
    const char* names[5] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "ds_write_b32", "read+add+write"};
    for (int pattern = 0; pattern < 3; ++pattern) {
        for (int i = 0; i < 4096; ++i) h[i] = pattern == 0 ? rand() & 4095 : pattern == 1 ? (i / 16 + (i % 16) * 256) & 4095 : (i % 16) * 7;
        hipMemcpy(d, h, 4096 * sizeof(int), hipMemcpyHostToDevice);
        for (int mode = 0; mode < 5; ++mode) {
            const int iters = 64;
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, c, s, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, c, s, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, d, c, s, iters);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, d, c, s, iters);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, d, c, s, iters);
                hipDeviceSynchronize();
            }
            unsigned long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
            const double ops = 256.0 * 16 * iters;
            printf("pattern %d (%s) %-16s: %8llu cycles for %.0f lane-ops per workgroup -> %.2f lane-ops/clk/CU\n", pattern,
                   pattern == 0 ? "random" : pattern == 1 ? "lane-consecutive" : "16 hot addresses", names[mode], cy, ops, ops / cy);
        }
    }
    return 0;
}

// Microbenchmark for the headline FNO step: what does the rows -> modes -> rows seam of a spectral block cost when it is kept
// INSIDE one launch (persistent row workgroups + mode workgroups exchanging x1 / spec through write-through stores, sharded
// arrival counters and sc1 loads; cdna_hip_programming.md Guideline 16) against the same data movement as two dependent
// launches per block (what libdlwpmi does today: fno_spatial_kernel / fno_mix_kernel)?
// Geometry = the headline config: B 4, H 64, C 32, 12 x 7 kept modes.  256 row workgroups (one image row each) publish
// 1792 B of x1 per block; 84 mode workgroups read one x1 column (64 KB), publish 1 KB of spec; every row workgroup reads its
// sample's spec (21.5 KB).  Arithmetic is replaced by a spin of a given number of cycles; every handed-off word is CHECKED.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/handoff_probe.hip -o tools/micro/handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int B = 4, H = 64, C = 32, M1 = 12, M2C = 7, NMODE = M1 * M2C, NROW = B * H;
constexpr int ROW_F4 = M2C * C / 2;          // float4 per x1 row (7 * 32 complex = 112 float4)
constexpr int MODE_F4 = C / 2;               // float4 per (sample, mode) of spec (32 complex = 16 float4)
constexpr int NT = 512;
constexpr int SHARDS = 8, SHARD_STRIDE = 32; // one 128-byte line per shard
constexpr unsigned SPIN_LIMIT = 4000000;

typedef __attribute__((address_space(1))) unsigned gu32;

struct Args {
    float4* x1;     // [NL][NROW][ROW_F4]
    float4* spec;   // [NL][B][NMODE][MODE_F4]
    const float4* wts;   // [NMODE][512] (8 KB per mode, emulates the weight slice)
    unsigned* cntA; // [NL][SHARDS][SHARD_STRIDE]
    unsigned* cntB;
    unsigned* err;  // [0] mismatches, [1] timeouts, [2..] first mismatch record
    unsigned long long* stamps;
    float* sink;
    int NL, row_spin, mode_spin;
    unsigned salt;
};

__device__ __forceinline__ float4 val(unsigned l, unsigned idx, unsigned salt, unsigned kind) {
    const unsigned u = (l * 2654435761u) ^ (idx * 40503u + kind * 977u) ^ (salt * 69069u);
    return make_float4(__uint_as_float((u & 0x007fffffu) | 0x3f800000u), __uint_as_float(((u >> 3) & 0x007fffffu) | 0x3f800000u),
                       __uint_as_float(((u >> 7) & 0x007fffffu) | 0x3f800000u), (float)(idx & 1023));
}
__device__ __forceinline__ bool same(const float4 a, const float4 b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }
__device__ __forceinline__ void record(unsigned* err, unsigned where, unsigned l, unsigned idx, float4 got, float4 want) {
    if (atomicAdd(err + 2, 1u) == 0) {
        err[3] = where; err[4] = l; err[5] = idx;
        err[6] = __float_as_uint(got.x); err[7] = __float_as_uint(want.x); err[8] = __float_as_uint(got.w); err[9] = __float_as_uint(want.w);
    }
}

__device__ __forceinline__ void spin(int cycles) {
    if (cycles <= 0) return;
    const unsigned long long t0 = __builtin_readcyclecounter();
    while ((long long)(__builtin_readcyclecounter() - t0) < cycles) __builtin_amdgcn_s_sleep(1);
}

// write-through 16-byte store / L1-bypassing 16-byte load (sc1): the hand-off forms of Guideline 16, R1
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_sc1(float4* p, float4 v) {
    const f32x4 x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ f32x4 load_sc1(const float4* p) {     // the result may be READ only behind drain_loaded()
    f32x4 x;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(x) : "v"(p) : "memory");
    return x;
}
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// hipcc does not model the loads inside an asm statement (cdna_hip_programming.md 5.7): the wait must CARRY the loaded registers,
// otherwise their uses may be scheduled above it
__device__ __forceinline__ void drain_loaded(f32x4& v) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory"); }
__device__ __forceinline__ f32x4 as_f32x4(const float4 v) { return f32x4{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ float4 as_float4(const f32x4 v) { return make_float4(v.x, v.y, v.z, v.w); }

// one wave polls the SHARDS counters (lane s < SHARDS reads shard s) until each holds its expected count
__device__ __forceinline__ bool wait_counts(unsigned* cnt, unsigned expect_lo, unsigned expect_hi, int n_hi, unsigned* err) {
    const int lane = threadIdx.x & 63;
    const unsigned want = lane < n_hi ? expect_hi : expect_lo;
    for (unsigned spins = 0;; ++spins) {
        unsigned v = want;
        if (lane < SHARDS) v = __hip_atomic_load((gu32*)(cnt + lane * SHARD_STRIDE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v >= want)) return true;
        if (spins > SPIN_LIMIT) {
            if (lane == 0) atomicAdd(err + 1, 1u);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

template <bool SC1>
__global__ __launch_bounds__(NT, 4) void persistent_kernel(Args a) {   // 4 waves per SIMD = two workgroups per CU: all 340 resident
    const int tid = threadIdx.x, w = tid >> 6;
    unsigned bad = 0;
    __shared__ int ok_s;
    if (blockIdx.x == 0 && tid == 0) a.stamps[0] = __builtin_amdgcn_s_memrealtime();
    if ((int)blockIdx.x < NMODE) {
        // ---- mode workgroup (kx, j)
        const int m = blockIdx.x, kx = m / M1;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int l = 0; l < a.NL; ++l) {
            const float4 wv = a.wts[m * NT + tid];          // weight slice: independent of the hand-off, issued before the wait
            if (w == 0) {
                const bool ok = wait_counts(a.cntA + (size_t)l * SHARDS * SHARD_STRIDE, NROW / SHARDS, NROW / SHARDS, 0, a.err);
                if (tid == 0) ok_s = ok;
                if (!SC1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                drain();
            }
            __syncthreads();
            if (!ok_s) break;
            const float4* xl = a.x1 + (size_t)l * NROW * ROW_F4;
            float4 v[8];
            f32x4 vr[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int u = tid + NT * q, row = u / MODE_F4, e = u % MODE_F4;      // 256 rows x 16 float4 of column kx
                const float4* p = xl + (size_t)row * ROW_F4 + kx * MODE_F4 + e;
                if (SC1) vr[q] = load_sc1(p);
                else v[q] = *p;
            }
            if (SC1) {
#pragma unroll
                for (int q = 0; q < 8; ++q) drain_loaded(vr[q]);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = as_float4(vr[q]);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int u = tid + NT * q, row = u / MODE_F4, e = u % MODE_F4;
                { const float4 wv_ = val(l, row * ROW_F4 + kx * MODE_F4 + e, a.salt, 0); if (!same(v[q], wv_)) { ++bad; record(a.err, 0, l, row * ROW_F4 + kx * MODE_F4 + e, v[q], wv_); } }
                acc.x += v[q].x;
            }
            acc.y += wv.x;
            spin(a.mode_spin);
            float4* sl = a.spec + (size_t)l * B * NMODE * MODE_F4;
            if (tid < B * MODE_F4) {
                const int b = tid / MODE_F4, e = tid % MODE_F4, idx = (b * NMODE + m) * MODE_F4 + e;
                store_sc1(sl + idx, val(l, idx, a.salt, 1));
            }
            drain();
            __syncthreads();
            if (tid == 0)
                __hip_atomic_fetch_add((gu32*)(a.cntB + ((size_t)l * SHARDS + (m & 7)) * SHARD_STRIDE), 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        }
        a.sink[blockIdx.x * NT + tid] = acc.x + acc.y;
    } else {
        // ---- row workgroup (b, h)
        const int r = blockIdx.x - NMODE, b = r / H;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int l = 0; l < a.NL; ++l) {
            spin(a.row_spin);
            float4* xl = a.x1 + (size_t)l * NROW * ROW_F4;
            if (tid < ROW_F4) store_sc1(xl + (size_t)r * ROW_F4 + tid, val(l, r * ROW_F4 + tid, a.salt, 0));
            drain();
            __syncthreads();
            if (tid == 0)
                __hip_atomic_fetch_add((gu32*)(a.cntA + ((size_t)l * SHARDS + (r & 7)) * SHARD_STRIDE), 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            if (w == 0) {
                // 84 modes over 8 shards: shards 0..3 receive 11 arrivals, 4..7 receive 10
                const bool ok = wait_counts(a.cntB + (size_t)l * SHARDS * SHARD_STRIDE, NMODE / SHARDS, NMODE / SHARDS + 1, NMODE % SHARDS, a.err);
                if (tid == 0) ok_s = ok;
                if (!SC1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                drain();
            }
            __syncthreads();
            if (!ok_s) break;
            const float4* sl = a.spec + (size_t)l * B * NMODE * MODE_F4 + (size_t)b * NMODE * MODE_F4;
            float4 v[3];
            f32x4 vr[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int u = tid + NT * q;
                const float4* p = sl + (u < NMODE * MODE_F4 ? u : 0);
                if (SC1) vr[q] = load_sc1(p);
                else v[q] = *p;
            }
            if (SC1) {
#pragma unroll
                for (int q = 0; q < 3; ++q) drain_loaded(vr[q]);
#pragma unroll
                for (int q = 0; q < 3; ++q) v[q] = as_float4(vr[q]);
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int u = tid + NT * q;
                if (u < NMODE * MODE_F4) { const float4 wv_ = val(l, b * NMODE * MODE_F4 + u, a.salt, 1); if (!same(v[q], wv_)) { ++bad; record(a.err, 1, l, b * NMODE * MODE_F4 + u, v[q], wv_); } }
                acc.x += v[q].x;
            }
        }
        a.sink[blockIdx.x * NT + tid] = acc.x;
    }
    if (bad) atomicAdd(a.err, bad);
    if (blockIdx.x == gridDim.x - 1 && tid == 0) a.stamps[1] = __builtin_amdgcn_s_memrealtime();
}

// ---- the same data movement as dependent launches (plain stores / loads; the kernel boundary is the hand-off)
__global__ __launch_bounds__(256) void rows_kernel(Args a, int l) {
    const int tid = threadIdx.x, r = blockIdx.x, b = r / H;
    unsigned bad = 0;
    float acc = 0.f;
    if (l > 0) {
        const float4* sl = a.spec + (size_t)(l - 1) * B * NMODE * MODE_F4 + (size_t)b * NMODE * MODE_F4;
        float4 v[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) v[q] = sl[min(tid + 256 * q, NMODE * MODE_F4 - 1)];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int u = tid + 256 * q;
            if (u < NMODE * MODE_F4) { const float4 wv_ = val(l - 1, b * NMODE * MODE_F4 + u, a.salt, 1); if (!same(v[q], wv_)) { ++bad; record(a.err, 2, l - 1, b * NMODE * MODE_F4 + u, v[q], wv_); } }
            acc += v[q].x;
        }
    }
    spin(a.row_spin);
    if (l < a.NL) {
        float4* xl = a.x1 + (size_t)l * NROW * ROW_F4;
        if (tid < ROW_F4) xl[(size_t)r * ROW_F4 + tid] = val(l, r * ROW_F4 + tid, a.salt, 0);
    }
    a.sink[blockIdx.x * 256 + tid] = acc;
    if (bad) atomicAdd(a.err, bad);
}
__global__ __launch_bounds__(NT) void modes_kernel(Args a, int l) {
    const int tid = threadIdx.x, m = blockIdx.x, kx = m / M1;
    unsigned bad = 0;
    const float4 wv = a.wts[m * NT + tid];
    const float4* xl = a.x1 + (size_t)l * NROW * ROW_F4;
    float4 v[8];
    float acc = wv.x;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int u = tid + NT * q, row = u / MODE_F4, e = u % MODE_F4;
        v[q] = xl[(size_t)row * ROW_F4 + kx * MODE_F4 + e];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int u = tid + NT * q, row = u / MODE_F4, e = u % MODE_F4;
        { const float4 wv_ = val(l, row * ROW_F4 + kx * MODE_F4 + e, a.salt, 0); if (!same(v[q], wv_)) { ++bad; record(a.err, 0, l, row * ROW_F4 + kx * MODE_F4 + e, v[q], wv_); } }
        acc += v[q].x;
    }
    spin(a.mode_spin);
    float4* sl = a.spec + (size_t)l * B * NMODE * MODE_F4;
    if (tid < B * MODE_F4) {
        const int b = tid / MODE_F4, e = tid % MODE_F4, idx = (b * NMODE + m) * MODE_F4 + e;
        sl[idx] = val(l, idx, a.salt, 1);
    }
    a.sink[blockIdx.x * NT + tid] = acc;
    if (bad) atomicAdd(a.err, bad);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int NL = 8, reps = 30;
    Args a{};
    a.NL = NL;
    CK(hipMalloc(&a.x1, sizeof(float4) * NL * NROW * ROW_F4));
    CK(hipMalloc(&a.spec, sizeof(float4) * NL * B * NMODE * MODE_F4));
    float4* wts; CK(hipMalloc(&wts, sizeof(float4) * NMODE * NT)); CK(hipMemset(wts, 0, sizeof(float4) * NMODE * NT)); a.wts = wts;
    const size_t cnt_bytes = sizeof(unsigned) * NL * SHARDS * SHARD_STRIDE;
    unsigned* cnt; CK(hipMalloc(&cnt, 2 * cnt_bytes)); a.cntA = cnt; a.cntB = cnt + NL * SHARDS * SHARD_STRIDE;
    CK(hipMalloc(&a.err, 64)); CK(hipMemset(a.err, 0, 64));
    CK(hipMalloc(&a.stamps, 16));
    CK(hipMalloc(&a.sink, sizeof(float) * (NMODE + NROW) * NT));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned salt = 1;
    // poison the payload buffers so that a stale read can never look right
    CK(hipMemset(a.x1, 0xff, sizeof(float4) * NL * NROW * ROW_F4)); CK(hipMemset(a.spec, 0xff, sizeof(float4) * NL * B * NMODE * MODE_F4));
    const int spins[][2] = {{0, 0}, {2500, 1200}, {5000, 2400}};
    for (const auto& sp : spins) {
        a.row_spin = sp[0]; a.mode_spin = sp[1];
        for (int variant = 0; variant < 2; ++variant) {       // 0: sc1 loads, no fence; 1: plain loads behind an agent acquire
            float ms_tot = 0.f; double span = 0;
            for (int rep = 0; rep < reps + 3; ++rep) {
                a.salt = ++salt;
                CK(hipMemsetAsync(cnt, 0, 2 * cnt_bytes, s));
                CK(hipEventRecord(e0, s));
                if (variant == 0) hipLaunchKernelGGL(persistent_kernel<true>, dim3(NMODE + NROW), dim3(NT), 0, s, a);
                else hipLaunchKernelGGL(persistent_kernel<false>, dim3(NMODE + NROW), dim3(NT), 0, s, a);
                CK(hipEventRecord(e1, s));
                CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                unsigned long long st[2]; CK(hipMemcpy(st, a.stamps, 16, hipMemcpyDeviceToHost));
                if (rep >= 3) { ms_tot += ms; span += (double)(st[1] - st[0]) * 0.01; }
            }
            unsigned err[2]; CK(hipMemcpy(err, a.err, 8, hipMemcpyDeviceToHost));
            printf("persistent (%s) spins row %5d / mode %5d cycles: %7.2f us per launch of %d blocks = %6.2f us per block; in-kernel span %6.2f us "
                   "per block; mismatches %u timeouts %u\n", variant == 0 ? "sc1 loads     " : "acquire fence ", sp[0], sp[1],
                   ms_tot * 1e3 / reps, NL, ms_tot * 1e3 / reps / NL, span / reps / NL, err[0], err[1]);
            if (err[0]) { unsigned rec[16]; CK(hipMemcpy(rec, a.err, 64, hipMemcpyDeviceToHost));
                printf("  first mismatch: site %u layer %u idx %u got.x %08x want.x %08x got.w %08x want.w %08x\n", rec[3], rec[4], rec[5], rec[6], rec[7], rec[8], rec[9]); }
            if (err[1]) { printf("TIMEOUT: aborting\n"); return 2; }
        }
        // baseline: 2 launches per block, captured in a graph
        a.salt = ++salt;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l <= NL; ++l) {
            hipLaunchKernelGGL(rows_kernel, dim3(NROW), dim3(256), 0, s, a, l);
            if (l < NL) hipLaunchKernelGGL(modes_kernel, dim3(NMODE), dim3(NT), 0, s, a, l);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float ms_tot = 0.f;
        for (int rep = 0; rep < reps + 3; ++rep) {
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 3) ms_tot += ms;
        }
        unsigned err[2]; CK(hipMemcpy(err, a.err, 8, hipMemcpyDeviceToHost));
        printf("launches   (graph, %2d kernels) spins row %5d / mode %5d cycles: %7.2f us per replay = %6.2f us per block; mismatches %u\n",
               2 * NL + 1, sp[0], sp[1], ms_tot * 1e3 / reps, ms_tot * 1e3 / reps / NL, err[0]);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}

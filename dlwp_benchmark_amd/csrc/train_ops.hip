// Loss / optimizer pieces of the training step (reference: nn.MSELoss, clip_grad_norm_,
// torch.optim.Adam at src/nsbench/scripts/train.py:72-74,118-127; dlwpbench train.py:126-139),
// plus the thread-local error string and the contiguous-tensor wrappers of the C ABI.
// All of these are HBM-bound streaming kernels: 16 B per lane, grid-stride, <= 2048 blocks.
#include "common.hip.h"
#include "dlwpmi_internal.h"
#include <algorithm>
#include <string>
#include <mutex>
#include <unordered_map>

static thread_local std::string g_last_error;

void dlwp_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int dlwp_ensure_lds(const void* kernel, size_t bytes, const char* what) {
    static std::mutex mu;
    static std::unordered_map<const void*, size_t> limit;
    if (bytes > 160 * 1024) {
        dlwp_set_error("%s: LDS request %zu exceeds the 160 KiB of a CU", what, bytes);
        return DLWP_E_UNSUPPORTED;
    }
    std::lock_guard<std::mutex> lk(mu);
    auto it = limit.find(kernel);
    if (it != limit.end() && it->second >= bytes) return DLWP_OK;
    DLWP_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    limit[kernel] = bytes;
    return DLWP_OK;
}

extern "C" const char* dlwp_last_error(void) { return g_last_error.c_str(); }
extern "C" int dlwp_version(void) { return DLWPMI_VERSION; }

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// block-wide sum -> one atomic per block
__device__ __forceinline__ void block_atomic_sum(float v, float* out) {
    __shared__ float part[4];
    v = wave_sum(v);
    if (lane_id() == 0) part[wave_id()] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomic_add_f32(out, part[0] + part[1] + part[2] + part[3]);
}

__global__ __launch_bounds__(256) void sqerr_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    long long n, float scale, float* out) {
    float acc = 0.f;
    const long long n4 = n / 4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 x = reinterpret_cast<const float4*>(a)[i];
        const float4 y = reinterpret_cast<const float4*>(b)[i];
        const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
        acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float d = a[n4 * 4 + threadIdx.x] - b[n4 * 4 + threadIdx.x];
        acc += d * d;
    }
    block_atomic_sum(acc * scale, out);
}

// loss += mean((a-b)^2), g = 2 (a-b) / n  (torch.nn.MSELoss forward + backward in one pass)
__global__ __launch_bounds__(256) void mse_grad_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       long long n, float inv_n, float* out, float* __restrict__ g) {
    float acc = 0.f;
    const long long stride = (long long)gridDim.x * blockDim.x, t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool vec = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(g)) & 15) == 0;
    const long long n4 = vec ? n / 4 : 0;
    const float s = 2.f * inv_n;
    for (long long i = t0; i < n4; i += stride) {
        const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
        const float4 d = make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w);
        acc += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        reinterpret_cast<float4*>(g)[i] = make_float4(s * d.x, s * d.y, s * d.z, s * d.w);
    }
    for (long long i = 4 * n4 + t0; i < n; i += stride) {
        const float d = a[i] - b[i];
        acc += d * d;
        g[i] = s * d;
    }
    block_atomic_sum(acc * inv_n, out);
}

// Evaluation metrics (nsbench/scripts/evaluate.py:232-257, dlwpbench/scripts/evaluate.py:494-546): error moments per
// group g of a [B][G][H][W] pair, optionally weighted per row h (latitude weights) and relative to a climatology c:
//   m[0][g] = sum w (o-t)^2   m[1][g] = sum w |o-t|   m[2][g] = sum w (o-c)(t-c)   m[3][g] = sum w (o-c)^2   m[4][g] = sum w (t-c)^2
__global__ __launch_bounds__(256) void error_moments_kernel(const float* __restrict__ o, const float* __restrict__ t,
                                                            const float* __restrict__ c, const float* __restrict__ roww, int B,
                                                            int G, int H, int W, float* m) {
    const int g = blockIdx.x;
    const long long hw = (long long)H * W, per = (long long)B * hw;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
    for (long long e = (long long)blockIdx.y * 256 + threadIdx.x; e < per; e += (long long)gridDim.y * 256) {
        const long long b = e / hw, r = e - b * hw;
        const long long idx = (b * G + g) * hw + r;
        const float w = roww ? roww[r / W] : 1.f;
        const float ov = o[idx], tv = t[idx], d = ov - tv;
        a0 += w * d * d;
        a1 += w * fabsf(d);
        if (c) {
            const float oc = ov - c[idx], tc = tv - c[idx];
            a2 += w * oc * tc;
            a3 += w * oc * oc;
            a4 += w * tc * tc;
        }
    }
    block_atomic_sum(a0, m + g);
    __syncthreads();                       // block_atomic_sum reuses one LDS scratch
    block_atomic_sum(a1, m + G + g);
    if (c) {
        __syncthreads();
        block_atomic_sum(a2, m + 2 * G + g);
        __syncthreads();
        block_atomic_sum(a3, m + 3 * G + g);
        __syncthreads();
        block_atomic_sum(a4, m + 4 * G + g);
    }
}

__global__ __launch_bounds__(256) void zero_kernel(float* __restrict__ p, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = 0.f;
}

__global__ __launch_bounds__(256) void zero_2d_kernel(float* __restrict__ p, long long ld, int rows, int cols) {
    const long long n = (long long)rows * cols, stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const long long r = i / cols;
        p[r * ld + (i - r * cols)] = 0.f;
    }
}

// The squared norm behind gradient clipping must come out BIT-IDENTICAL on every data-parallel rank (the ranks hold the same
// all-reduced gradient; a last-bit difference in the clipping coefficient makes the replicas drift apart): float atomics add
// the block partials in arrival order, so the partials go to a scratch array and a one-block kernel adds them in index order.
// (One scratch per device: calls on different streams at the same time would share it; the training paths use one stream.)
constexpr int SUMSQ_MAX_BLOCKS = 4096;
__device__ float g_sumsq_partial[SUMSQ_MAX_BLOCKS];
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, float* part_out) {
    float acc = 0.f;
    const long long n4 = n / 4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 x = reinterpret_cast<const float4*>(g)[i];
        acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float d = g[n4 * 4 + threadIdx.x];
        acc += d * d;
    }
    __shared__ float part[4];
    acc = wave_sum(acc);
    if (lane_id() == 0) part[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part_out[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
__global__ __launch_bounds__(256) void sumsq_finish_kernel(const float* __restrict__ part_in, int nblocks, float* out) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) acc += part_in[i];      // fixed order per thread
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {                                    // fixed tree
        if ((int)threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out += red[0];
}

__global__ __launch_bounds__(256) void clip_scale_kernel(float* g, long long n, const float* sumsq,
                                                         float grad_scale, float max_norm) {
    const float total = sqrtf(*sumsq) * fabsf(grad_scale);
    float coef = max_norm / (total + 1e-6f);
    coef = coef < 1.f ? coef : 1.f;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] *= coef;
}

// sumsq != nullptr: the gradient is clipped on the way in (torch.nn.utils.clip_grad_norm_'s coefficient from the squared norm
// dlwp_sumsq left there) -- the separate read-modify-write pass of dlwp_clip_scale over the gradient buffer is not needed
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   const int* step_ptr, long long n, float lr, float beta1,
                                                   float beta2, float eps, float grad_scale, int zero_grad,
                                                   const float* __restrict__ sumsq, float max_norm) {
    if (sumsq) {
        const float total = sqrtf(*sumsq) * fabsf(grad_scale);
        const float coef = max_norm / (total + 1e-6f);
        grad_scale *= coef < 1.f ? coef : 1.f;
    }
    const int step = *step_ptr + 1;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = 1.f - powf(beta2, (float)step);
    const float step_size = lr / bc1;
    const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i] * grad_scale;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p[i] -= step_size * (mi / denom);
        if (zero_grad) g[i] = 0.f;
    }
}

// The counter is advanced by its own one-thread launch: an in-kernel "last workgroup increments" ticket costs one
// same-address atomic per workgroup (~25 ns each, serialised: ~50 us for the 2048-workgroup grid, measured -0.8 % on the
// headline step), ten times the launch it saves.
__global__ void step_inc_kernel(int* step) { *step += 1; }

// Start of a fused rollout step in one launch: loss = 0, g_out = 0, and the frames the rollout returns unchanged
// (out[b][0 : row] = x[b][0 : row], nsbench fno.py:240-243) -- three tiny launches otherwise.
__global__ __launch_bounds__(256) void rollout_prep_kernel(float* loss, float* __restrict__ g_out, long long n,
                                                           float* __restrict__ out, const float* __restrict__ x,
                                                           long long out_bs, long long x_bs, long long row, int B) {
    const long long stride = (long long)gridDim.x * blockDim.x, t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t0 == 0 && loss) *loss = 0.f;
    if (g_out)
        for (long long i = t0; i < n; i += stride) g_out[i] = 0.f;
    if (row > 0)
        for (long long i = t0; i < row * B; i += stride) {
            const long long b = i / row, r = i - b * row;
            out[b * out_bs + r] = x[b * x_bs + r];
        }
}

int stream_grid(long long n) {
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace

extern "C" int dlwp_sqerr_sum(const float* a, const float* b, long long n, float scale, float* loss_out,
                              void* stream) {
    DLWP_REQUIRE(a && b && loss_out && n >= 0, DLWP_E_INVALID, "sqerr_sum: NULL argument");
    if (n == 0) return DLWP_OK;
    hipLaunchKernelGGL(sqerr_kernel, dim3(stream_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, a, b, n,
                       scale, loss_out);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_mse_fwd_bwd(const float* pred, const float* target, long long n, float* loss_out, float* grad,
                                void* stream) {
    DLWP_REQUIRE(pred && target && loss_out && grad && n > 0, DLWP_E_INVALID, "mse_fwd_bwd: NULL argument or n <= 0");
    // one same-address atomic per workgroup (~25 ns each, serialised): at most 512 workgroups of >= 8192 elements (16-byte
    // accesses) -- the 2048-block scalar grid spent 29 us on a 2.6 MB loss (C3 at batch 16)
    const int blocks = (int)std::max<long long>(1, std::min<long long>(512, n / 8192));
    hipLaunchKernelGGL(mse_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, target, n,
                       1.0f / (float)n, loss_out, grad);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_error_moments(const float* out, const float* target, const float* climatology, const float* row_weights,
                                  int B, int G, int H, int W, float* moments, void* stream) {
    DLWP_REQUIRE(out && target && moments && B > 0 && G > 0 && H > 0 && W > 0, DLWP_E_INVALID, "error_moments: bad argument");
    DLWP_REQUIRE(G <= 65535, DLWP_E_UNSUPPORTED, "error_moments: more than 65535 groups");
    const long long per = (long long)B * H * W;
    const int ny = (int)std::max<long long>(1, std::min<long long>(64, per / 4096));
    hipLaunchKernelGGL(error_moments_kernel, dim3(G, ny), dim3(256), 0, (hipStream_t)stream, out, target, climatology,
                       row_weights, B, G, H, W, moments);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

int dlwp_zero_f32(float* p, long long n, void* stream) {
    if (n <= 0) return DLWP_OK;
    hipLaunchKernelGGL(zero_kernel, dim3((int)std::min<long long>((n + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream, p, n);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

namespace {
struct AddManyDev { dlwp_add2d_desc d[DLWP_ADD2D_MAX]; };
// dst[r][c] += src[r * sr + c * sc] for up to DLWP_ADD2D_MAX small matrices in one launch (blockIdx.y = matrix): the hand-over of
// padded / transposed temporary weight gradients to the parameters' gradient slots (SFNO encoder / decoder / position embedding)
__global__ __launch_bounds__(256) void add2d_many_kernel(AddManyDev a) {
    const dlwp_add2d_desc d = a.d[blockIdx.y];
    const long long n = (long long)d.rows * d.cols;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const long long r = e / d.cols, c = e - r * d.cols;
        d.dst[r * d.dst_ld + c] += d.src[r * d.src_rs + c * d.src_cs];
    }
}
}  // namespace

extern "C" int dlwp_add2d_many(const dlwp_add2d_desc* descs, int n, void* stream) {
    DLWP_REQUIRE(descs && n >= 1 && n <= DLWP_ADD2D_MAX, DLWP_E_INVALID, "add2d_many: 1 .. %d matrices", DLWP_ADD2D_MAX);
    AddManyDev a{};
    long long most = 0;
    for (int i = 0; i < n; ++i) {
        DLWP_REQUIRE(descs[i].dst && descs[i].src && descs[i].rows > 0 && descs[i].cols > 0, DLWP_E_INVALID,
                     "add2d_many: matrix %d has a NULL pointer or an empty shape", i);
        a.d[i] = descs[i];
        most = std::max(most, (long long)descs[i].rows * descs[i].cols);
    }
    const dim3 grid((unsigned)std::min<long long>((most + 255) / 256, 1024), (unsigned)n);
    hipLaunchKernelGGL(add2d_many_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

int dlwp_zero_2d_f32(float* p, long long ld, int rows, int cols, void* stream) {
    const long long n = (long long)rows * cols;
    if (n <= 0) return DLWP_OK;
    hipLaunchKernelGGL(zero_2d_kernel, dim3((int)std::min<long long>((n + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream, p, ld,
                       rows, cols);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

namespace {
// out[b][i] = (x ? x[b][i] : 0) + s[b] * t[b][i]; one sample's row is n floats (n % 4 == 0: 16-byte accesses)
__global__ __launch_bounds__(256) void scale_rows_add_kernel(const float* __restrict__ t, const float* __restrict__ s,
                                                             const float* __restrict__ x, float* __restrict__ out,
                                                             long long n4, int vec, int x_shared) {
    const int b = blockIdx.y;
    const float sc = s ? s[b] : 1.f;
    if (x && !x_shared) x += (long long)b * n4 * (vec ? 4 : 1);      // x_shared: one row for every sample (broadcast add)
    const long long stride = (long long)gridDim.x * 256;
    if (vec) {
        const float4* t4 = reinterpret_cast<const float4*>(t) + (long long)b * n4;
        const float4* x4 = x ? reinterpret_cast<const float4*>(x) : nullptr;
        float4* o4 = reinterpret_cast<float4*>(out) + (long long)b * n4;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            float4 v = t4[i];
            float4 r = x4 ? x4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            r.x = fmaf(sc, v.x, r.x); r.y = fmaf(sc, v.y, r.y); r.z = fmaf(sc, v.z, r.z); r.w = fmaf(sc, v.w, r.w);
            o4[i] = r;
        }
    } else {      // n4 = n here
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            const long long o = (long long)b * n4 + i;
            out[o] = fmaf(sc, t[o], x ? x[i] : 0.f);
        }
    }
}
}  // namespace

static int scale_rows_add_impl(const float* t, const float* scale, const float* x, float* out, int B, long long n, int x_shared,
                               void* stream) {
    DLWP_REQUIRE(t && out && B > 0 && n > 0, DLWP_E_INVALID, "scale_rows_add: bad argument");
    DLWP_REQUIRE(B <= 65535, DLWP_E_UNSUPPORTED, "scale_rows_add: at most 65535 rows (got %d)", B);
    const bool vec = n % 4 == 0 && ((uintptr_t)t % 16 == 0) && ((uintptr_t)out % 16 == 0) && (!x || (uintptr_t)x % 16 == 0);
    const long long units = vec ? n / 4 : n;
    long long bx = (units + 255) / 256;
    const long long cap = std::max<long long>(1, 2048 / B);
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL(scale_rows_add_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, (hipStream_t)stream, t, scale, x,
                       out, units, vec ? 1 : 0, x_shared);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_scale_rows_add(const float* t, const float* scale, const float* x, float* out, int B, long long n,
                                   void* stream) {
    return scale_rows_add_impl(t, scale, x, out, B, n, 0, stream);
}

extern "C" int dlwp_add_bcast(const float* t, const float* p, float* out, int B, long long n, void* stream) {
    DLWP_REQUIRE(p, DLWP_E_INVALID, "add_bcast: NULL argument");
    return scale_rows_add_impl(t, nullptr, p, out, B, n, 1, stream);
}

extern "C" int dlwp_sumsq(const float* g, long long n, float* out, void* stream) {
    DLWP_REQUIRE(g && out && n >= 0, DLWP_E_INVALID, "sumsq: NULL argument");
    if (n == 0) return DLWP_OK;
    const int blocks = std::min(stream_grid(n / 4 + 1), SUMSQ_MAX_BLOCKS);
    // the scratch is a __device__ array of the code object: its address is per device, so it is resolved on every call for the
    // CURRENT device (hipGetSymbolAddress enqueues nothing: safe under stream capture)
    float* scratch = nullptr;
    DLWP_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&scratch), HIP_SYMBOL(g_sumsq_partial)));
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, n, scratch);
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, blocks, out);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_clip_scale(float* g, long long n, const float* sumsq, float grad_scale, float max_norm,
                               void* stream) {
    DLWP_REQUIRE(g && sumsq && n >= 0, DLWP_E_INVALID, "clip_scale: NULL argument");
    if (n == 0) return DLWP_OK;
    hipLaunchKernelGGL(clip_scale_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, g, n, sumsq,
                       grad_scale, max_norm);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_adam_step_clipped(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int* step,
                                      long long n, float lr, float beta1, float beta2, float eps, float grad_scale,
                                      int zero_grad, const float* sumsq, float max_norm, void* stream) {
    DLWP_REQUIRE(param && grad && exp_avg && exp_avg_sq && step && n >= 0, DLWP_E_INVALID,
                 "adam_step: NULL argument");
    DLWP_REQUIRE(n > 0, DLWP_E_INVALID, "adam_step: empty parameter buffer");
    DLWP_REQUIRE(!sumsq || max_norm > 0.f, DLWP_E_INVALID, "adam_step_clipped: max_norm must be positive");
    hipLaunchKernelGGL(adam_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       exp_avg, exp_avg_sq, step, n, lr, beta1, beta2, eps, grad_scale, zero_grad, sumsq, max_norm);
    DLWP_LAUNCH_CHECK();
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int* step,
                              long long n, float lr, float beta1, float beta2, float eps, float grad_scale,
                              int zero_grad, void* stream) {
    return dlwp_adam_step_clipped(param, grad, exp_avg, exp_avg_sq, step, n, lr, beta1, beta2, eps, grad_scale, zero_grad,
                                  nullptr, 0.f, stream);
}

int dlwp_rollout_prep(float* loss, float* g_out, long long n, float* out, const float* x, long long out_bs, long long x_bs,
                      long long row, int B, hipStream_t stream) {
    const long long work = std::max<long long>(n, row * B);
    hipLaunchKernelGGL(rollout_prep_kernel, dim3(stream_grid(work)), dim3(256), 0, stream, loss, g_out, n, out, x, out_bs,
                       x_bs, row, B);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

// ---- contiguous-tensor wrappers of the pointwise MLP
extern "C" int dlwp_pwmlp_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                              float* y, int B, int Cin, int Ch, int Cout, int P, void* stream) {
    DLWP_REQUIRE(x && w1 && b1 && w2 && b2 && y, DLWP_E_INVALID, "pwmlp_fwd: NULL argument");
    dlwp_chan_src xs{x, (long long)Cin * P, P, nullptr, nullptr};
    dlwp_chan_dst yd{y, (long long)Cout * P, P, nullptr, nullptr};
    return dlwp_pwmlp_fwd_ex(&xs, w1, b1, w2, b2, &yd, nullptr, B, Cin, Ch, Cout, P, (hipStream_t)stream);
}

extern "C" int dlwp_pwmlp_bwd(const float* x, const float* w1, const float* b1, const float* w2, const float* gy,
                              float* gx, float* gw1, float* gb1, float* gw2, float* gb2, int B, int Cin, int Ch,
                              int Cout, int P, void* stream) {
    DLWP_REQUIRE(x && w1 && b1 && w2 && gy && gw1 && gb1 && gw2 && gb2, DLWP_E_INVALID, "pwmlp_bwd: NULL argument");
    dlwp_chan_src xs{x, (long long)Cin * P, P, nullptr, nullptr};
    dlwp_chan_src gys{gy, (long long)Cout * P, P, nullptr, nullptr};
    dlwp_chan_dst gxd{gx, (long long)Cin * P, P, nullptr, nullptr};
    return dlwp_pwmlp_bwd_ex(&xs, w1, b1, w2, &gys, nullptr, nullptr, 0.f, gx ? &gxd : nullptr, 0, nullptr, gw1, gb1,
                             gw2, gb2, nullptr, 0, B, Cin, Ch, Cout, P, (hipStream_t)stream);
}

// same as dlwp_pwmlp_bwd, but the parameter gradients go to a per-workgroup partial slab (the path the rollout
// trainer uses); fold with dlwp_pwmlp_slab_fold.
extern "C" long long dlwp_pwmlp_slab_floats(int B, int Cin, int Ch, int Cout, int P) {
    return (long long)dlwp_pwmlp_slab_count(B, P) * dlwp_pwmlp_slab_stride(Cin, Ch, Cout);
}
extern "C" int dlwp_pwmlp_bwd_slab(const float* x, const float* w1, const float* b1, const float* w2, const float* gy,
                                   float* gx, float* slab, int slab_accumulate, int B, int Cin, int Ch, int Cout, int P,
                                   void* stream) {
    DLWP_REQUIRE(x && w1 && b1 && w2 && gy && slab, DLWP_E_INVALID, "pwmlp_bwd_slab: NULL argument");
    dlwp_chan_src xs{x, (long long)Cin * P, P, nullptr, nullptr};
    dlwp_chan_src gys{gy, (long long)Cout * P, P, nullptr, nullptr};
    dlwp_chan_dst gxd{gx, (long long)Cin * P, P, nullptr, nullptr};
    return dlwp_pwmlp_bwd_ex(&xs, w1, b1, w2, &gys, nullptr, nullptr, 0.f, gx ? &gxd : nullptr, 0, nullptr, nullptr,
                             nullptr, nullptr, nullptr, slab, slab_accumulate, B, Cin, Ch, Cout, P, (hipStream_t)stream);
}
extern "C" int dlwp_pwmlp_slab_fold(const float* slab, float* gw1, float* gb1, float* gw2, float* gb2, int B, int Cin,
                                    int Ch, int Cout, int P, void* stream) {
    DLWP_REQUIRE(slab && gw1 && gb1 && gw2 && gb2, DLWP_E_INVALID, "pwmlp_slab_fold: NULL argument");
    return dlwp_pwmlp_slab_reduce(slab, dlwp_pwmlp_slab_count(B, P), Cin, Ch, Cout, gw1, gb1, gw2, gb2, (hipStream_t)stream);
}

// ---- debug: a chain of n dependent empty kernels (measures the per-kernel floor of a stream / graph)
namespace { __global__ void null_kernel(int* p) { if (p && threadIdx.x == 1024) *p = 0; } }
extern "C" int dlwp_debug_null_kernels(int n, int blocks, void* stream) {
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(null_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, nullptr);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

// ---- debug: a chain of n dependent kernels whose every wave spins for `cycles` shader cycles, with a chosen grid, block size
// and dynamic LDS size: what a launch costs beyond its body as a function of the number and size of its workgroups
namespace {
__global__ void spin_kernel(int cycles, int* sink) {
    extern __shared__ float spin_lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) spin_lds[0] = 1.f;
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < cycles) __builtin_amdgcn_s_sleep(1);
    if (sink && spin_lds[0] == 3.f) *sink = 1;
}
}  // namespace
extern "C" int dlwp_debug_spin_kernels(int n, int blocks, int threads, int cycles, int lds_bytes, void* stream) {
    DLWP_REQUIRE(n > 0 && blocks > 0 && threads > 0 && threads <= 1024 && cycles >= 0 && cycles < (1 << 24) && lds_bytes >= 4,
                 DLWP_E_INVALID, "debug_spin_kernels: bad argument");
    int rc = dlwp_ensure_lds(reinterpret_cast<const void*>(spin_kernel), (size_t)lds_bytes, "spin");
    if (rc) return rc;
    for (int i = 0; i < n; ++i)
        hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(threads), (size_t)lds_bytes, (hipStream_t)stream, cycles, nullptr);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

// ---- debug: effective shader clock during a short kernel: out[0]=delta s_memtime, out[1]=delta s_memrealtime (100 MHz)
namespace {
__global__ void clock_probe_kernel(unsigned long long* out, int iters) {
    float x = threadIdx.x * 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    if (x == 12345.678f) out[2] = 1;
}
}
extern "C" int dlwp_debug_clock_probe(unsigned long long* out, int iters, int blocks, void* stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

// The sliding input window of the autoregressive rollouts, advanced by ONE kernel per lead time (and one in the backward pass).
//
// Reference: every rollout driver rebuilds its input window from Python lists on each step --
//   nsbench  AFNONet.forward / SwinTransformer.forward (src/nsbench/models/fourcastnet/fourcastnet.py:262-300,
//            swintransformer/swin_transformer.py:597-640): x_t = cat([x_obs, stack(outs[-k:])]); out = x_t[:, -1] + net(x_t)
//   dlwpbench UNet/SFNONet/... .forward (src/dlwpbench/models/unet/unet.py:64-111): prog_t = cat([prognostic[...], stack(outs)[:, -ctx:]])
// i.e. stack + cat + residual add per step in the forward pass, and in the backward pass one gradient accumulation per consumer
// of every predicted frame (each frame is read by up to ctx later windows).  Here the window is a single tensor:
//   next[b][j] = win[b][j + 1]            (j < ctx - 1)
//   next[b][ctx - 1] = out[b] = win[b][ctx - 1] + delta[b]
// and its adjoint, which also sums the (up to) three gradients that reach `next` / `out` (from the following advance, from the
// network that read the window, from the loss):
//   G = g_adv + g_net;  g_win[b][k] = (k >= 1 ? G[b][k - 1] : 0) + (k == ctx - 1 ? G[b][ctx - 1] + g_out[b] : 0);
//   g_delta[b] = G[b][ctx - 1] + g_out[b].
// A frame is F = D*H*W floats, [D][H][W] row-major.  `delta` arrives either in frame layout or as the patch tokens of a
// linear head, [B][H/ph][W/pw][ph][pw][D] (AFNONet.head, fourcastnet.py:233,296-298), which folds the un-patching permute
// into this kernel.  Pure data movement: HBM-bound, 16-byte accesses when F % 4 == 0 and delta is in frame layout.
#include "common.hip.h"
#include "dlwpmi_internal.h"

namespace {

struct AdvDev {
    const float *win, *delta, *g_adv, *g_net, *g_out;
    float *next, *out, *g_win, *g_delta;
    long long win_bs, net_bs, out_bs, F;
    int B, ctx, layout, D, H, W, ph, pw;
};

// offset of frame element f = (d, y, x) of sample b inside the patch-token tensor [B][H/ph][W/pw][ph][pw][D]
__device__ __forceinline__ long long patch_offset(const AdvDev& a, int b, long long f) {
    const int x = (int)(f % a.W);
    const long long r = f / a.W;
    const int y = (int)(r % a.H), d = (int)(r / a.H);
    const int hh = a.H / a.ph, ww = a.W / a.pw;
    return (((((long long)b * hh + y / a.ph) * ww + x / a.pw) * a.ph + y % a.ph) * a.pw + x % a.pw) * a.D + d;
}

template <int VEC>
__global__ __launch_bounds__(256) void advance_fwd_kernel(AdvDev a) {
    const int b = blockIdx.y;
    const long long n = a.F / VEC;
    const float* wb = a.win + (long long)b * a.win_bs;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long f = i * VEC;
        float v[VEC], dl[VEC];
        if (VEC == 4) {
            *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(wb + (long long)(a.ctx - 1) * a.F + f);
            *reinterpret_cast<f32x4*>(dl) = *reinterpret_cast<const f32x4*>(a.delta + (long long)b * a.F + f);
        } else {
            v[0] = wb[(long long)(a.ctx - 1) * a.F + f];
            dl[0] = a.delta[a.layout ? patch_offset(a, b, f) : (long long)b * a.F + f];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] += dl[k];
        if (VEC == 4) {
            *reinterpret_cast<f32x4*>(a.out + (long long)b * a.F + f) = *reinterpret_cast<const f32x4*>(v);
            if (a.next) *reinterpret_cast<f32x4*>(a.next + ((long long)b * a.ctx + a.ctx - 1) * a.F + f) = *reinterpret_cast<const f32x4*>(v);
        } else {
            a.out[(long long)b * a.F + f] = v[0];
            if (a.next) a.next[((long long)b * a.ctx + a.ctx - 1) * a.F + f] = v[0];
        }
        if (a.next)
            for (int j = 0; j + 1 < a.ctx; ++j) {
                if (VEC == 4)
                    *reinterpret_cast<f32x4*>(a.next + ((long long)b * a.ctx + j) * a.F + f) =
                        *reinterpret_cast<const f32x4*>(wb + (long long)(j + 1) * a.F + f);
                else
                    a.next[((long long)b * a.ctx + j) * a.F + f] = wb[(long long)(j + 1) * a.F + f];
            }
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void advance_bwd_kernel(AdvDev a) {
    const int b = blockIdx.y;
    const long long n = a.F / VEC;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long f = i * VEC;
        float prev[VEC];               // G[b][k - 1] while walking k upwards
#pragma unroll
        for (int k = 0; k < VEC; ++k) prev[k] = 0.f;
        for (int k = 0; k < a.ctx; ++k) {
            float g[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) g[q] = 0.f;
            auto add_from = [&](const float* p) {
                if (VEC == 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
                    for (int q = 0; q < 4; ++q) g[q] += t[q];
                } else {
                    g[0] += *p;
                }
            };
            if (a.g_adv) add_from(a.g_adv + ((long long)b * a.ctx + k) * a.F + f);
            if (a.g_net) add_from(a.g_net + (long long)b * a.net_bs + (long long)k * a.F + f);
            float w[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) w[q] = prev[q];
            if (k == a.ctx - 1) {
                if (a.g_out) add_from(a.g_out + (long long)b * a.out_bs + f);      // g now holds G[ctx-1] + g_out = g_delta
#pragma unroll
                for (int q = 0; q < VEC; ++q) w[q] += g[q];
                if (VEC == 4) *reinterpret_cast<f32x4*>(a.g_delta + (long long)b * a.F + f) = *reinterpret_cast<const f32x4*>(g);
                else a.g_delta[a.layout ? patch_offset(a, b, f) : (long long)b * a.F + f] = g[0];
            }
            if (a.g_win) {
                if (VEC == 4) *reinterpret_cast<f32x4*>(a.g_win + ((long long)b * a.ctx + k) * a.F + f) = *reinterpret_cast<const f32x4*>(w);
                else a.g_win[((long long)b * a.ctx + k) * a.F + f] = w[0];
            }
#pragma unroll
            for (int q = 0; q < VEC; ++q) prev[q] = g[q];
        }
    }
}

bool aligned16(const void* p) { return p == nullptr || ((uintptr_t)p % 16) == 0; }

int adv_check(const AdvDev& a, const char* who) {
    DLWP_REQUIRE(a.B > 0 && a.B <= 65535 && a.ctx > 0 && a.F > 0, DLWP_E_INVALID, "%s: B %d, ctx %d, frame %lld", who, a.B, a.ctx, a.F);
    if (a.layout) {
        DLWP_REQUIRE(a.D > 0 && a.H > 0 && a.W > 0 && a.ph > 0 && a.pw > 0 && a.H % a.ph == 0 && a.W % a.pw == 0 &&
                         (long long)a.D * a.H * a.W == a.F,
                     DLWP_E_INVALID, "%s: patch layout D %d H %d W %d patch %d x %d does not match the frame size %lld", who, a.D,
                     a.H, a.W, a.ph, a.pw, a.F);
    }
    return DLWP_OK;
}

dim3 adv_grid(const AdvDev& a, int vec) {
    long long bx = (a.F / vec + 255) / 256;
    const long long cap = std::max<long long>(1, 4096 / a.B);
    return dim3((unsigned)std::min(bx, cap), (unsigned)a.B);
}

}  // namespace

extern "C" int dlwp_window_advance_fwd(const float* win, long long win_batch_stride, const float* delta, float* next, float* out,
                                       int B, int ctx, long long frame, int delta_layout, int D, int H, int W, int ph, int pw,
                                       void* stream) {
    DLWP_REQUIRE(win && delta && out, DLWP_E_INVALID, "window_advance_fwd: NULL argument");
    AdvDev a{};
    a.win = win; a.win_bs = win_batch_stride; a.delta = delta; a.next = next; a.out = out; a.B = B; a.ctx = ctx; a.F = frame;
    a.layout = delta_layout; a.D = D; a.H = H; a.W = W; a.ph = ph; a.pw = pw;
    int rc = adv_check(a, "window_advance_fwd");
    if (rc) return rc;
    DLWP_REQUIRE(win_batch_stride >= (long long)ctx * frame, DLWP_E_INVALID, "window_advance_fwd: batch stride %lld < ctx * frame",
                 win_batch_stride);
    const bool vec = !delta_layout && frame % 4 == 0 && win_batch_stride % 4 == 0 && aligned16(win) && aligned16(delta) &&
                     aligned16(next) && aligned16(out);
    if (vec) hipLaunchKernelGGL(advance_fwd_kernel<4>, adv_grid(a, 4), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(advance_fwd_kernel<1>, adv_grid(a, 1), dim3(256), 0, (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

extern "C" int dlwp_window_advance_bwd(const float* g_next, const float* g_net, long long net_batch_stride, const float* g_out,
                                       long long out_batch_stride, float* g_win, float* g_delta, int B, int ctx, long long frame,
                                       int delta_layout, int D, int H, int W, int ph, int pw, void* stream) {
    DLWP_REQUIRE(g_delta, DLWP_E_INVALID, "window_advance_bwd: NULL g_delta");
    AdvDev a{};
    a.g_adv = g_next; a.g_net = g_net; a.net_bs = net_batch_stride; a.g_out = g_out; a.out_bs = out_batch_stride;
    a.g_win = g_win; a.g_delta = g_delta; a.B = B; a.ctx = ctx; a.F = frame;
    a.layout = delta_layout; a.D = D; a.H = H; a.W = W; a.ph = ph; a.pw = pw;
    int rc = adv_check(a, "window_advance_bwd");
    if (rc) return rc;
    const bool vec = !delta_layout && frame % 4 == 0 && net_batch_stride % 4 == 0 && out_batch_stride % 4 == 0 &&
                     aligned16(g_next) && aligned16(g_net) && aligned16(g_out) && aligned16(g_win) && aligned16(g_delta);
    if (vec) hipLaunchKernelGGL(advance_bwd_kernel<4>, adv_grid(a, 4), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(advance_bwd_kernel<1>, adv_grid(a, 1), dim3(256), 0, (hipStream_t)stream, a);
    DLWP_LAUNCH_CHECK();
    return DLWP_OK;
}

// Fused FNO rollout trainer: autoregressive rollout forward + MSE + BPTT backward as one
// statically planned kernel sequence, replayed as a hipGraph.
//
// Reference semantics restated (file:line under /root/reference/src):
//   nsbench/models/fno/fno.py:217-250  TFNO2DModule.forward  (context window, teacher forcing,
//                                      closed loop; FNOModule :29-41 is the context_size=1 case)
//   nsbench/scripts/train.py:118-122   zero_grad -> model(x, tf) -> MSELoss -> backward
// Differences by design (MI355X-first): the per-step th.stack/th.cat of <= ctx frames is replaced
// by pointer tables into the trajectory buffers (the lifting kernel gathers its input channels
// from observations or earlier predictions in place); all T steps stay on device; activations
// of every net call are kept in HBM (a few MB each, far below 288 GB) for BPTT.
#include "common.hip.h"
#include "dlwpmi_internal.h"
#include <algorithm>
#include <vector>

struct dlwp_fno_trainer {
    dlwp_fno_cfg cfg;
    dlwp_fno_plan* plan = nullptr;
    int ncalls = 0, Cin = 0, T_out = 0;
    long long frame = 0;      // D*H*W (prognostic channels per frame)
    long long traj = 0;       // T*D*H*W           (input trajectory stride per sample)
    long long traj_out = 0;   // T_out*D*H*W       (prediction / target stride per sample)
    const float *constants = nullptr, *prescribed = nullptr;   // dlwp form (borrowed)
    struct CallInfo {         // host-side plan of one net call
        int out_slot;                       // index into out/y/g_out along their time axis
        const float* res; long long res_bs; // residual frame added to the net output (dlwp form), or nullptr
        float* gres; long long gres_bs;     // where the residual path sends its gradient, or nullptr
    };
    std::vector<CallInfo> calls;
    long long act = 0;        // C*H*W per sample
    // trajectory buffers borrowed from the caller (dlwp_fno_trainer_bind_io)
    const float *x = nullptr, *y = nullptr;
    float *out = nullptr, *loss = nullptr;
    // trainer-owned device memory
    float* g_out = nullptr;
    float *h0 = nullptr, *pre = nullptr;     // [ncalls][B,C,H,W], [ncalls][NL][B,C,H,W]
    float2* xhat = nullptr;                  // [ncalls][NL][B,m1,m2c,C]
    float2 *x1 = nullptr, *spec = nullptr;   // work
    float *gA = nullptr, *gB = nullptr;      // work [B,C,H,W]
    float *slab_lift = nullptr, *slab_proj = nullptr;  // per-workgroup parameter-gradient partials
    float* slab_skip = nullptr;                        // [n_layers][B*H][gslab_stride] skip-weight / bias partials
    // wide layers (hidden > 64, fno_wide.hip): dense lifting inputs and the stored hidden activations of both MLPs
    bool wide = false;
    float *xin = nullptr;                    // [ncalls][B][Cin][HW]
    float *zl = nullptr, *al = nullptr;      // [ncalls][B][lifting][HW]   pre-activation / GELU of the lifting MLP
    float *zp = nullptr, *ap = nullptr;      // [ncalls][B][projection][HW]
    float *gz = nullptr;                     // work [B][max(lifting, projection)][HW]
    float *gyb = nullptr, *gxin = nullptr;   // work [B][D][HW], [B][Cin][HW]
    const float** src_tab = nullptr;         // [ncalls][Cin]
    float** gdst_tab = nullptr;              // [ncalls][Cin]
    long long* bstride_tab = nullptr;        // [ncalls][Cin]
    std::vector<const float*> h_src;         // host copies of the tables (which window planes are earlier predictions:
    std::vector<float*> h_gdst;              //  the chained projection -> lifting launches hand those over in LDS)
    std::vector<long long> h_bs;
    // bound by the caller
    float *params = nullptr, *grads = nullptr;
    // graph
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    hipStream_t cap_stream = nullptr;  // capture happens on a private stream (the caller's may be the
                                       // legacy default stream, which cannot be captured)
};

namespace {

long long pad4(long long n) { return (n + 3) & ~3LL; }

// DLWP_FNO_FORM_NS_CONTEXT3D: the block kernels see a (ctx * H) x W image with m0 * m1 row frequencies (plan_create3d)
bool is_ctx3d(const dlwp_fno_cfg& c) { return c.form == DLWP_FNO_FORM_NS_CONTEXT3D; }
int eff_H(const dlwp_fno_cfg& c) { return is_ctx3d(c) ? c.context_size * c.H : c.H; }
int eff_m1(const dlwp_fno_cfg& c) { return is_ctx3d(c) ? c.m0 * c.m1 : c.m1; }
// channels the lifting layer reads / entries of a net call's gather table
long long lift_cin(const dlwp_fno_cfg& c) {
    if (c.form == DLWP_FNO_FORM_DLWP) return (long long)c.constant_channels + (long long)(c.prescribed_channels + c.D) * c.context_size;
    return is_ctx3d(c) ? c.D : (long long)c.D * c.context_size;
}

struct Layout {
    long long lw1, lb1, lw2, lb2, pw1, pb1, pw2, pb2, layers, per_layer, spec_sz, skip_sz, bias_sz, total;
};

Layout make_layout(const dlwp_fno_cfg& c) {
    Layout L{};
    const long long Cin = lift_cin(c);
    long long o = 0;
    L.lw1 = o; o += pad4((long long)c.lifting * Cin);
    L.lb1 = o; o += pad4(c.lifting);
    L.lw2 = o; o += pad4((long long)c.hidden * c.lifting);
    L.lb2 = o; o += pad4(c.hidden);
    L.pw1 = o; o += pad4((long long)c.projection * c.hidden);
    L.pb1 = o; o += pad4(c.projection);
    L.pw2 = o; o += pad4((long long)c.out_channels * c.projection);
    L.pb2 = o; o += pad4(c.out_channels);
    L.layers = o;
    L.spec_sz = (long long)eff_m1(c) * c.m2c * c.hidden * c.hidden * 2;
    L.skip_sz = (long long)c.hidden * c.hidden;
    L.bias_sz = c.hidden;
    L.per_layer = pad4(L.spec_sz) + pad4(L.skip_sz) + pad4(L.bias_sz);
    L.total = o + L.per_layer * c.n_layers;
    return L;
}

template <typename T>
int dmalloc(T** p, size_t n) {
    DLWP_HIP(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
    return DLWP_OK;
}

int check_cfg(const dlwp_fno_cfg& c) {
    DLWP_REQUIRE(c.B > 0 && c.T > 0 && c.D > 0 && c.H > 0 && c.W > 0, DLWP_E_INVALID, "fno_trainer: bad shape");
    DLWP_REQUIRE(c.context_size >= 1, DLWP_E_INVALID, "fno_trainer: context_size must be >= 1");
    DLWP_REQUIRE(c.context_size <= c.T, DLWP_E_INVALID, "fno_trainer: context_size > T");
    DLWP_REQUIRE(c.form == DLWP_FNO_FORM_NS || c.form == DLWP_FNO_FORM_DLWP || c.form == DLWP_FNO_FORM_NS_CONTEXT3D, DLWP_E_INVALID,
                 "fno_trainer: unknown form");
    if (is_ctx3d(c)) {
        DLWP_REQUIRE(c.m0 >= 1 && c.m0 <= c.context_size && c.context_size >= 2, DLWP_E_INVALID,
                     "fno_trainer: context3d form needs 1 <= m0 <= context_size and context_size >= 2");
    }
    if (c.form == DLWP_FNO_FORM_NS || is_ctx3d(c)) {
        DLWP_REQUIRE(c.teacher_forcing_steps >= c.context_size - 1, DLWP_E_UNSUPPORTED,
                     "fno_trainer: teacher_forcing_steps < context_size-1 (the reference slices x with a "
                     "negative start there, fno.py:236) is not supported");
        DLWP_REQUIRE(c.constant_channels == 0 && c.prescribed_channels == 0, DLWP_E_INVALID,
                     "fno_trainer: constant/prescribed channels only exist in the dlwp form");
    } else {
        DLWP_REQUIRE(c.context_size < c.T, DLWP_E_INVALID, "fno_trainer: dlwp form needs T > context_size");
        DLWP_REQUIRE(c.constant_channels >= 0 && c.prescribed_channels >= 0, DLWP_E_INVALID, "fno_trainer: bad channels");
    }
    DLWP_REQUIRE(c.n_layers >= 1 && c.hidden > 0 && c.lifting > 0 && c.projection > 0, DLWP_E_INVALID,
                 "fno_trainer: bad widths");
    DLWP_REQUIRE(c.out_channels == c.D, DLWP_E_INVALID, "fno_trainer: out_channels must equal D");
    DLWP_REQUIRE((c.H * c.W) % 4 == 0, DLWP_E_UNSUPPORTED, "fno_trainer: H*W must be a multiple of 4");
    return DLWP_OK;
}

struct P {  // resolved parameter pointers inside a flat buffer
    float *lw1, *lb1, *lw2, *lb2, *pw1, *pb1, *pw2, *pb2;
    float* base; Layout L;
    float* spec(int l) const { return base + L.layers + L.per_layer * l; }
    float* skip(int l) const { return spec(l) + pad4(L.spec_sz); }
    float* bias(int l) const { return skip(l) + pad4(L.skip_sz); }
};
P resolve(float* base, const Layout& L) {
    return P{base + L.lw1, base + L.lb1, base + L.lw2, base + L.lb2, base + L.pw1, base + L.pb1,
             base + L.pw2, base + L.pb2, base, L};
}

// fused: the caller is the fused train step -- the loss / g_out zero fills ride with the copy of the untouched frames
int enqueue_forward(dlwp_fno_trainer* tr, bool keep, hipStream_t s, bool fused = false) {
    const dlwp_fno_cfg& c = tr->cfg;
    const Layout L = make_layout(c);
    const P w = resolve(tr->params, L);
    const int HW = c.H * c.W, C = c.hidden, NL = c.n_layers, ctx = c.context_size;
    const bool v3d = is_ctx3d(c);
    const int P = v3d ? ctx * HW : HW, LCin = (int)lift_cin(c);      // pixels of a hidden field; channels the lifting layer reads
    const long long actB = (long long)c.B * tr->act;
    const long long xhatB = (long long)c.B * eff_m1(c) * c.m2c * C;
    int rc;
    const bool copy_head = (c.form == DLWP_FNO_FORM_NS || v3d) && ctx > 1;
    if (fused) {
        if ((rc = dlwp_rollout_prep(tr->loss, tr->g_out, (long long)c.B * tr->traj_out, tr->out, tr->x, tr->traj_out, tr->traj,
                                    copy_head ? (long long)(ctx - 1) * tr->frame : 0, c.B, s))) return rc;
    } else if (copy_head) {
        // steps before the context is full return the latest observation (fno.py:240-243)
        DLWP_HIP(hipMemcpy2DAsync(tr->out, tr->traj_out * sizeof(float), tr->x, tr->traj * sizeof(float),
                                  (size_t)(ctx - 1) * tr->frame * sizeof(float), c.B, hipMemcpyDeviceToDevice, s));
    }
    bool lifted = false;             // the lifting MLP of call k has already run (chained behind call k - 1's projection)
    for (int k = 0; k < tr->ncalls; ++k) {
        const dlwp_fno_trainer::CallInfo& ci = tr->calls[k];
        const int kk = keep ? k : 0;  // evaluation reuses slot 0
        float* h0 = tr->h0 + kk * actB;
        float* pre = tr->pre + (long long)kk * NL * actB;
        float2* xhat = tr->xhat + (long long)kk * NL * xhatB;
        if (tr->wide) {
            // channel-blocked kernels + channels-first GEMM MLPs (fno_wide.hip); same dataflow as below
            // (3-D context form: the gather table holds D * ctx frame planes = the [D][ctx * HW] volume the lifting layer
            // reads as D channels of P = ctx * HW pixels; the projection runs on the LAST time slice only, fno.py:96)
            const long long xinB = (long long)c.B * tr->Cin * HW, zlB = (long long)c.B * c.lifting * P,
                            zpB = (long long)c.B * c.projection * HW;
            float* xin = tr->xin + kk * xinB;
            if ((rc = dlwp_gather_channels(tr->src_tab + (long long)k * tr->Cin, tr->bstride_tab + (long long)k * tr->Cin, xin,
                                           c.B, tr->Cin, HW, s))) return rc;
            if ((rc = dlwp_cfmlp_fwd(xin, (long long)tr->Cin * HW, w.lw1, w.lb1, w.lw2, w.lb2, h0, tr->act, nullptr, 0,
                                     tr->zl + kk * zlB, tr->al + kk * zlB, c.B, LCin, c.lifting, C, P, s))) return rc;
            if ((rc = dlwp_fno_rows_dft(tr->plan, h0, 0, 0, tr->x1, c.B, s))) return rc;
            for (int l = 0; l < NL; ++l) {
                if ((rc = dlwp_fno_mix_fwd(tr->plan, tr->x1, reinterpret_cast<const float2*>(w.spec(l)),
                                           xhat + l * xhatB, tr->spec, c.B, s))) return rc;
                dlwp_fno_spatial_args a{};
                a.tin = l == 0 ? h0 : pre + (l - 1) * actB;
                a.act_tin = l > 0;
                a.spec = tr->spec; a.wskip = w.skip(l); a.bias = w.bias(l);
                a.out = pre + l * actB;
                a.x1_out = l < NL - 1 ? tr->x1 : nullptr;
                a.x1_act = 1;
                a.B = c.B;
                if ((rc = dlwp_fno_spatial(tr->plan, &a, s))) return rc;
            }
            if ((rc = dlwp_cfmlp_fwd(pre + (NL - 1) * actB + (P - HW), tr->act, w.pw1, w.pb1, w.pw2, w.pb2,
                                     tr->out + (long long)ci.out_slot * tr->frame, tr->traj_out, ci.res, ci.res_bs,
                                     tr->zp + kk * zpB, tr->ap + kk * zpB, c.B, C, c.projection, c.out_channels, HW, s, P))) return rc;
            continue;
        }
        // lifting MLP with the first block's W-axis DFT in its epilogue where the shapes allow (one launch less per net
        // call); otherwise the callee runs the separate rows kernel.  From the second call on it has normally already run,
        // chained behind the previous call's projection (below).
        if (!lifted) {
            dlwp_chan_src xs{nullptr, 0, 0, tr->src_tab + (long long)k * tr->Cin, tr->bstride_tab + (long long)k * tr->Cin};
            dlwp_chan_dst h0d{h0, tr->act, HW, nullptr, nullptr};
            if ((rc = dlwp_pwmlp_fwd_rows_ex(&xs, w.lw1, w.lb1, w.lw2, w.lb2, &h0d, nullptr, c.B, tr->Cin, c.lifting, C, HW,
                                             tr->plan, tr->x1, s))) return rc;
        }
        lifted = false;
        for (int l = 0; l < NL; ++l) {
            if ((rc = dlwp_fno_mix_fwd(tr->plan, tr->x1, reinterpret_cast<const float2*>(w.spec(l)),
                                       xhat + l * xhatB, tr->spec, c.B, s))) return rc;
            dlwp_fno_spatial_args a{};
            a.tin = l == 0 ? h0 : pre + (l - 1) * actB;
            a.act_tin = l > 0;
            a.spec = tr->spec; a.wskip = w.skip(l); a.bias = w.bias(l);
            a.out = pre + l * actB;
            a.x1_out = l < NL - 1 ? tr->x1 : nullptr;
            a.x1_act = 1;
            a.B = c.B;
            if ((rc = dlwp_fno_spatial(tr->plan, &a, s))) return rc;
        }
        dlwp_chan_src ps{pre + (NL - 1) * actB, tr->act, HW, nullptr, nullptr};
        float* oplane = tr->out + (long long)ci.out_slot * tr->frame;
        dlwp_chan_dst od{oplane, tr->traj_out, HW, nullptr, nullptr};
        dlwp_chan_src rs{ci.res, ci.res_bs, HW, nullptr, nullptr};  // out = last frame + net (dlwp fno.py:103)
        if (k + 1 < tr->ncalls && c.out_channels <= 8) {
            // projection of this call chained with the lifting of the next: the frame(s) just produced go to the next window in LDS
            signed char next_ch[8];
            for (int o = 0; o < 8; ++o) next_ch[o] = -1;
            const size_t t1 = (size_t)(k + 1) * tr->Cin;
            for (int o = 0; o < c.out_channels; ++o)
                for (int ch = 0; ch < tr->Cin; ++ch)
                    if (tr->h_src[t1 + ch] == oplane + (long long)o * HW && tr->h_bs[t1 + ch] == tr->traj_out) next_ch[o] = (signed char)ch;
            const int k1 = keep ? k + 1 : 0;
            dlwp_chan_src xs1{nullptr, 0, 0, tr->src_tab + (long long)(k + 1) * tr->Cin, tr->bstride_tab + (long long)(k + 1) * tr->Cin};
            dlwp_chan_dst h0d1{tr->h0 + k1 * actB, tr->act, HW, nullptr, nullptr};
            rc = dlwp_pwmlp_fwd_chain_ex(&ps, w.pw1, w.pb1, w.pw2, w.pb2, &od, ci.res ? &rs : nullptr, C, c.projection, c.out_channels,
                                         &xs1, w.lw1, w.lb1, w.lw2, w.lb2, &h0d1, tr->Cin, c.lifting, C, next_ch, c.B, HW,
                                         tr->plan, tr->x1, s);
            if (rc == DLWP_OK) { lifted = true; continue; }
            if (rc != DLWP_E_UNSUPPORTED) return rc;
        }
        if ((rc = dlwp_pwmlp_fwd_ex(&ps, w.pw1, w.pb1, w.pw2, w.pb2, &od, ci.res ? &rs : nullptr, c.B, C, c.projection,
                                    c.out_channels, HW, s))) return rc;
    }
    return DLWP_OK;
}

int enqueue_loss_backward(dlwp_fno_trainer* tr, const float* grad_out, hipStream_t s, bool fused = false) {
    const dlwp_fno_cfg& c = tr->cfg;
    const Layout L = make_layout(c);
    const P w = resolve(tr->params, L);
    const P g = resolve(tr->grads, L);
    const int HW = c.H * c.W, C = c.hidden, NL = c.n_layers;
    const bool v3d = is_ctx3d(c);
    const int P = v3d ? c.context_size * HW : HW, LCin = (int)lift_cin(c);
    const long long actB = (long long)c.B * tr->act;
    const long long xhatB = (long long)c.B * eff_m1(c) * c.m2c * C;
    const long long n = (long long)c.B * tr->traj_out;
    int rc;
    float mse_scale = 0.f;
    if (grad_out) {
        // arbitrary upstream gradient d loss / d out (autograd path of the nn.Module)
        DLWP_HIP(hipMemcpyAsync(tr->g_out, grad_out, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    } else {
        // fused nn.MSELoss(reduction="mean") against the trainer's target buffer
        if (!fused && (rc = dlwp_zero_f32(tr->loss, 1, s))) return rc;
        if ((rc = dlwp_sqerr_sum(tr->out, tr->y, n, 1.0f / (float)n, tr->loss, s))) return rc;
        if (!fused && (rc = dlwp_zero_f32(tr->g_out, n, s))) return rc;
        mse_scale = 2.0f / (float)n;
    }
    for (int k = tr->ncalls - 1; k >= 0; --k) {
        const dlwp_fno_trainer::CallInfo& ci = tr->calls[k];
        const long long oofs = (long long)ci.out_slot * tr->frame;
        float* h0 = tr->h0 + k * actB;
        float* pre = tr->pre + (long long)k * NL * actB;
        float2* xhat = tr->xhat + (long long)k * NL * xhatB;
        float *gcur = tr->gA, *gnext = tr->gB;
        if (tr->wide) {
            const long long xinB = (long long)c.B * tr->Cin * HW, zlB = (long long)c.B * c.lifting * P,
                            zpB = (long long)c.B * c.projection * HW, CP = (long long)c.out_channels * HW;
            // upstream of the projection = accumulated closed-loop gradient + d MSE / d out[t]; identity path of the residual
            if ((rc = dlwp_proj_gy(tr->g_out + oofs, grad_out ? nullptr : tr->out + oofs, grad_out ? nullptr : tr->y + oofs,
                                   mse_scale, tr->gyb, ci.gres, tr->traj_out, ci.gres_bs, CP, c.B, s))) return rc;
            // 3-D context form: only the last time slice of the volume reaches the projection -- the rest of its gradient is zero
            if (v3d && (rc = dlwp_zero_f32(gcur, actB, s))) return rc;
            if ((rc = dlwp_cfmlp_bwd(pre + (NL - 1) * actB + (P - HW), tr->act, w.pw1, w.pw2, tr->gyb, CP, tr->zp + k * zpB,
                                     tr->ap + k * zpB, gcur + (P - HW), tr->act, tr->gz, g.pw1, g.pb1, g.pw2, g.pb2, c.B, C,
                                     c.projection, c.out_channels, HW, s, P, P))) return rc;
            if ((rc = dlwp_fno_rows_dft(tr->plan, gcur, 0, 1, tr->x1, c.B, s))) return rc;
            for (int l = NL - 1; l >= 0; --l) {
                if ((rc = dlwp_fno_mix_bwd(tr->plan, tr->x1, reinterpret_cast<const float2*>(w.spec(l)),
                                           xhat + l * xhatB, tr->spec, reinterpret_cast<float2*>(g.spec(l)), c.B, s))) return rc;
                dlwp_fno_spatial_args a{};
                a.tin = gcur; a.spec = tr->spec; a.wskip = w.skip(l); a.transpose_w = 1;
                a.pprev = l == 0 ? h0 : pre + (l - 1) * actB;
                a.act_prev = l > 0;
                a.out = gnext;
                a.x1_out = l > 0 ? tr->x1 : nullptr;
                a.x1_adjoint = 1;
                a.g_wskip = g.skip(l); a.g_bias = g.bias(l);
                a.inverse_adjoint = 1;
                a.B = c.B;
                if ((rc = dlwp_fno_spatial(tr->plan, &a, s))) return rc;
                float* tmp = gcur; gcur = gnext; gnext = tmp;
            }
            if ((rc = dlwp_cfmlp_bwd(tr->xin + k * xinB, (long long)tr->Cin * HW, w.lw1, w.lw2, gcur, tr->act, tr->zl + k * zlB,
                                     tr->al + k * zlB, tr->gxin, (long long)tr->Cin * HW, tr->gz, g.lw1, g.lb1, g.lw2, g.lb2, c.B,
                                     LCin, c.lifting, C, P, s))) return rc;
            if ((rc = dlwp_scatter_add_channels(tr->gdst_tab + (long long)k * tr->Cin, tr->bstride_tab + (long long)k * tr->Cin,
                                                tr->gxin, c.B, tr->Cin, HW, s))) return rc;
            continue;
        }
        // projection backward; upstream = accumulated closed-loop gradient + d MSE / d out[t]
        dlwp_chan_src ps{pre + (NL - 1) * actB, tr->act, HW, nullptr, nullptr};
        dlwp_chan_src gy{tr->g_out + oofs, tr->traj_out, HW, nullptr, nullptr};
        dlwp_chan_src pred{tr->out + oofs, tr->traj_out, HW, nullptr, nullptr};
        dlwp_chan_src targ{tr->y + oofs, tr->traj_out, HW, nullptr, nullptr};
        dlwp_chan_dst gx{gcur, tr->act, HW, nullptr, nullptr};
        dlwp_chan_dst gr{ci.gres, ci.gres_bs, HW, nullptr, nullptr};  // identity path of the residual
        // the adjoint W-axis DFT of the finished gradient rows rides in the epilogue where the shapes allow (one launch
        // less per net call); otherwise the callee runs the separate rows kernel
        if ((rc = dlwp_pwmlp_bwd_rows_ex(&ps, w.pw1, w.pb1, w.pw2, &gy, grad_out ? nullptr : &pred,
                                         grad_out ? nullptr : &targ, mse_scale, &gx, 0, ci.gres ? &gr : nullptr, g.pw1,
                                         g.pb1, g.pw2, g.pb2, tr->slab_proj, k != tr->ncalls - 1, c.B, C, c.projection,
                                         c.out_channels, HW, tr->plan, tr->x1, s))) return rc;
        for (int l = NL - 1; l >= 0; --l) {
            if ((rc = dlwp_fno_mix_bwd(tr->plan, tr->x1, reinterpret_cast<const float2*>(w.spec(l)),
                                       xhat + l * xhatB, tr->spec, reinterpret_cast<float2*>(g.spec(l)), c.B, s))) return rc;
            dlwp_fno_spatial_args a{};
            a.tin = gcur; a.spec = tr->spec; a.wskip = w.skip(l); a.transpose_w = 1;
            a.pprev = l == 0 ? h0 : pre + (l - 1) * actB;
            a.act_prev = l > 0;
            a.out = gnext;
            a.x1_out = l > 0 ? tr->x1 : nullptr;
            a.x1_adjoint = 1;
            a.gslab = tr->slab_skip + (long long)l * c.B * c.H * dlwp_fno_gslab_stride(C);
            a.gslab_accumulate = k != tr->ncalls - 1;
            a.inverse_adjoint = 1;
            a.B = c.B;
            if ((rc = dlwp_fno_spatial(tr->plan, &a, s))) return rc;
            float* tmp = gcur; gcur = gnext; gnext = tmp;
        }
        // lifting backward; input-channel gradients flow into the predictions that fed this step
        dlwp_chan_src xs{nullptr, 0, 0, tr->src_tab + (long long)k * tr->Cin, tr->bstride_tab + (long long)k * tr->Cin};
        dlwp_chan_src gh0{gcur, tr->act, HW, nullptr, nullptr};
        dlwp_chan_dst gxd{nullptr, 0, 0, tr->gdst_tab + (long long)k * tr->Cin, tr->bstride_tab + (long long)k * tr->Cin};
        if ((rc = dlwp_pwmlp_bwd_ex(&xs, w.lw1, w.lb1, w.lw2, &gh0, nullptr, nullptr, 0.f, &gxd, 1, nullptr, g.lw1, g.lb1,
                                    g.lw2, g.lb2, tr->slab_lift, k != tr->ncalls - 1, c.B, tr->Cin, c.lifting, C,
                                    HW, s))) return rc;
    }
    if (tr->wide) return DLWP_OK;       // the GEMMs accumulated straight into the gradient buffer
    // fold the per-workgroup partial slabs of all net calls into the gradient buffer: one launch when the job table holds
    // them all (n_layers + 2 jobs), otherwise layer by layer
    const int nslab = dlwp_pwmlp_slab_count(c.B, HW);
    std::vector<dlwp_fold_job> jobs;
    for (int l = 0; l < NL; ++l) {
        dlwp_fold_job q{};
        q.slab = tr->slab_skip + (long long)l * c.B * c.H * dlwp_fno_gslab_stride(C);
        q.nslab = c.B * c.H; q.stride = dlwp_fno_gslab_stride(C);
        q.d1 = g.skip(l); q.n1 = (long long)C * C; q.d2 = g.bias(l); q.n2 = C;
        jobs.push_back(q);
    }
    dlwp_fold_job qp{}, ql{};
    qp.slab = tr->slab_proj; qp.nslab = nslab; qp.pwmlp = 1; qp.Cin = C; qp.Ch = c.projection; qp.Cout = c.out_channels;
    qp.d1 = g.pw1; qp.d2 = g.pb1; qp.d3 = g.pw2; qp.d4 = g.pb2;
    ql.slab = tr->slab_lift; ql.nslab = nslab; ql.pwmlp = 1; ql.Cin = tr->Cin; ql.Ch = c.lifting; ql.Cout = C;
    ql.d1 = g.lw1; ql.d2 = g.lb1; ql.d3 = g.lw2; ql.d4 = g.lb2;
    jobs.push_back(qp);
    jobs.push_back(ql);
    for (size_t k = 0; k < jobs.size(); k += 8)
        if ((rc = dlwp_fold_slabs(jobs.data() + k, (int)std::min<size_t>(8, jobs.size() - k), s))) return rc;
    return DLWP_OK;
}

}  // namespace

extern "C" long long dlwp_fno_param_offset(const dlwp_fno_cfg* cfg, int kind, int layer, long long* size) {
    const dlwp_fno_cfg& c = *cfg;
    const Layout L = make_layout(c);
    const long long Cin = lift_cin(c);
    long long off = -1, sz = 0;
    switch (kind) {
        case -1: off = L.total; sz = L.total; break;
        case DLWP_FNO_P_LIFT_W1: off = L.lw1; sz = (long long)c.lifting * Cin; break;
        case DLWP_FNO_P_LIFT_B1: off = L.lb1; sz = c.lifting; break;
        case DLWP_FNO_P_LIFT_W2: off = L.lw2; sz = (long long)c.hidden * c.lifting; break;
        case DLWP_FNO_P_LIFT_B2: off = L.lb2; sz = c.hidden; break;
        case DLWP_FNO_P_PROJ_W1: off = L.pw1; sz = (long long)c.projection * c.hidden; break;
        case DLWP_FNO_P_PROJ_B1: off = L.pb1; sz = c.projection; break;
        case DLWP_FNO_P_PROJ_W2: off = L.pw2; sz = (long long)c.out_channels * c.projection; break;
        case DLWP_FNO_P_PROJ_B2: off = L.pb2; sz = c.out_channels; break;
        case DLWP_FNO_P_SPEC_W: off = L.layers + L.per_layer * layer; sz = L.spec_sz; break;
        case DLWP_FNO_P_SKIP_W: off = L.layers + L.per_layer * layer + pad4(L.spec_sz); sz = L.skip_sz; break;
        case DLWP_FNO_P_SPEC_B: off = L.layers + L.per_layer * layer + pad4(L.spec_sz) + pad4(L.skip_sz); sz = L.bias_sz; break;
        default: break;
    }
    if (size) *size = sz;
    return off;
}

extern "C" int dlwp_fno_trainer_create(const dlwp_fno_cfg* cfg, dlwp_fno_trainer** out) {
    DLWP_REQUIRE(cfg && out, DLWP_E_INVALID, "fno_trainer_create: NULL argument");
    int rc = check_cfg(*cfg);
    if (rc) return rc;
    dlwp_fno_trainer* tr = new dlwp_fno_trainer();
    tr->cfg = *cfg;
    const dlwp_fno_cfg& c = tr->cfg;
    if (is_ctx3d(c)) rc = dlwp_fno_plan_create3d(c.hidden, c.context_size, c.H, c.W, c.m0, c.m1, c.m2c, &tr->plan);
    else rc = dlwp_fno_plan_create(c.hidden, c.H, c.W, c.m1, c.m2c, &tr->plan);
    if (rc) { delete tr; return rc; }
    const bool dlwp = c.form == DLWP_FNO_FORM_DLWP;
    tr->ncalls = dlwp ? c.T - c.context_size : c.T - (c.context_size - 1);
    tr->T_out = dlwp ? c.T - c.context_size : c.T;
    // entries of a net call's channel gather table: every (channel, frame) plane the lifting layer reads
    tr->Cin = dlwp ? c.constant_channels + (c.prescribed_channels + c.D) * c.context_size : c.D * c.context_size;
    tr->frame = (long long)c.D * c.H * c.W;
    tr->traj = tr->frame * c.T;
    tr->traj_out = tr->frame * tr->T_out;
    tr->act = (long long)c.hidden * eff_H(c) * c.W;
    const size_t n = (size_t)c.B * tr->traj_out, actB = (size_t)c.B * tr->act;
    const size_t xhatB = (size_t)c.B * eff_m1(c) * c.m2c * c.hidden;
    tr->wide = dlwp_fno_is_wide(tr->plan);
    if (tr->wide) {
        const size_t HW = (size_t)c.H * c.W, nc = (size_t)tr->ncalls, Pl = (size_t)eff_H(c) * c.W;
        if ((rc = dmalloc(&tr->xin, nc * c.B * tr->Cin * HW)) || (rc = dmalloc(&tr->zl, nc * c.B * c.lifting * Pl)) ||
            (rc = dmalloc(&tr->al, nc * c.B * c.lifting * Pl)) || (rc = dmalloc(&tr->zp, nc * c.B * c.projection * HW)) ||
            (rc = dmalloc(&tr->ap, nc * c.B * c.projection * HW)) ||
            (rc = dmalloc(&tr->gz, (size_t)c.B * std::max((size_t)c.lifting * Pl, (size_t)c.projection * HW))) ||
            (rc = dmalloc(&tr->gyb, (size_t)c.B * c.out_channels * HW)) || (rc = dmalloc(&tr->gxin, (size_t)c.B * tr->Cin * HW))) {
            dlwp_fno_trainer_destroy(tr);
            return rc;
        }
    }
    if ((rc = dmalloc(&tr->g_out, n)) ||
        (rc = dmalloc(&tr->h0, actB * tr->ncalls)) || (rc = dmalloc(&tr->pre, actB * tr->ncalls * c.n_layers)) ||
        (rc = dmalloc(&tr->xhat, xhatB * tr->ncalls * c.n_layers)) ||
        (rc = dmalloc(&tr->x1, (size_t)c.B * eff_H(c) * c.m2c * c.hidden)) || (rc = dmalloc(&tr->spec, xhatB)) ||
        (rc = dmalloc(&tr->gA, actB)) || (rc = dmalloc(&tr->gB, actB)) ||
        (!tr->wide && ((rc = dmalloc(&tr->slab_skip, (size_t)c.n_layers * c.B * c.H * dlwp_fno_gslab_stride(c.hidden))) ||
                       (rc = dmalloc(&tr->slab_lift, (size_t)dlwp_pwmlp_slab_count(c.B, c.H * c.W) *
                                                         dlwp_pwmlp_slab_stride(tr->Cin, c.lifting, c.hidden))) ||
                       (rc = dmalloc(&tr->slab_proj, (size_t)dlwp_pwmlp_slab_count(c.B, c.H * c.W) *
                                                         dlwp_pwmlp_slab_stride(c.hidden, c.projection, c.out_channels))))) ||
        (rc = dmalloc(&tr->src_tab, (size_t)tr->ncalls * tr->Cin)) ||
        (rc = dmalloc(&tr->gdst_tab, (size_t)tr->ncalls * tr->Cin)) || (rc = dmalloc(&tr->bstride_tab, (size_t)tr->ncalls * tr->Cin))) {
        dlwp_fno_trainer_destroy(tr);
        return rc;
    }
    *out = tr;
    return DLWP_OK;
}

extern "C" int dlwp_fno_trainer_bind_io(dlwp_fno_trainer* tr, const float* x, const float* y, float* out_buf,
                                        float* loss) {
    DLWP_REQUIRE(tr && x && out_buf, DLWP_E_INVALID, "fno_trainer_bind_io: x and out are required");
    if (tr->graph_exec) {
        (void)hipGraphExecDestroy(tr->graph_exec);
        (void)hipGraphDestroy(tr->graph);
        tr->graph_exec = nullptr;
        tr->graph = nullptr;
    }
    tr->x = x; tr->y = y; tr->out = out_buf; tr->loss = loss;
    const dlwp_fno_cfg& c = tr->cfg;
    std::vector<const float*> src((size_t)tr->ncalls * tr->Cin);
    std::vector<float*> gdst((size_t)tr->ncalls * tr->Cin);
    std::vector<long long> bs((size_t)tr->ncalls * tr->Cin);
    tr->calls.assign(tr->ncalls, dlwp_fno_trainer::CallInfo{0, nullptr, 0, nullptr, 0});
    const long long HW = (long long)c.H * c.W;
    const int ctx = c.context_size;
    if (c.form == DLWP_FNO_FORM_NS || is_ctx3d(c)) {
        // window frame j of step t comes from the observations while j < tf, from the prediction out[j-1]
        // afterwards (nsbench fno.py:228-237; the context form :77-87 builds the same window and transposes it to
        // [B, D, ctx, H, W]: plane order (d, frame) instead of (frame, d))
        for (int k = 0; k < tr->ncalls; ++k) {
            const int t = ctx - 1 + k;
            tr->calls[k].out_slot = t;
            for (int sl = 0; sl < ctx; ++sl) {
                const int j = t - ctx + 1 + sl;
                for (int d = 0; d < c.D; ++d) {
                    const size_t ch = (size_t)k * tr->Cin + (is_ctx3d(c) ? (size_t)d * ctx + sl : (size_t)sl * c.D + d);
                    bs[ch] = tr->traj;
                    if (j < c.teacher_forcing_steps) {
                        src[ch] = tr->x + (long long)j * tr->frame + d * HW;
                        gdst[ch] = nullptr;
                    } else {
                        src[ch] = tr->out + (long long)(j - 1) * tr->frame + d * HW;
                        gdst[ch] = tr->g_out + (long long)(j - 1) * tr->frame + d * HW;
                    }
                }
            }
        }
    } else {
        // dlwpbench fno.py:49-62,75-103: channels = [constants | prescribed frames t-ctx..t-1 | prognostic frames
        // t-ctx..t-1]; prognostic frame j is the observation while j < ctx, the prediction out[j-ctx] afterwards;
        // out[k] = (last prognostic frame) + net(x_t)
        DLWP_REQUIRE(c.constant_channels == 0 || tr->constants, DLWP_E_INVALID, "fno_trainer: constants not bound");
        DLWP_REQUIRE(c.prescribed_channels == 0 || tr->prescribed, DLWP_E_INVALID, "fno_trainer: prescribed not bound");
        const int Cc = c.constant_channels, Cp = c.prescribed_channels, Cg = c.D;
        for (int k = 0; k < tr->ncalls; ++k) {
            const int t = ctx + k;
            size_t ch = (size_t)k * tr->Cin;
            tr->calls[k].out_slot = k;
            for (int q = 0; q < Cc; ++q, ++ch) { src[ch] = tr->constants + q * HW; gdst[ch] = nullptr; bs[ch] = (long long)Cc * HW; }
            for (int sl = 0; sl < ctx; ++sl)
                for (int q = 0; q < Cp; ++q, ++ch) {
                    src[ch] = tr->prescribed + ((long long)(t - ctx + sl) * Cp + q) * HW;
                    gdst[ch] = nullptr;
                    bs[ch] = (long long)c.T * Cp * HW;
                }
            for (int sl = 0; sl < ctx; ++sl) {
                const int j = t - ctx + sl;
                for (int q = 0; q < Cg; ++q, ++ch) {
                    if (j < ctx) {
                        src[ch] = tr->x + ((long long)j * Cg + q) * HW; gdst[ch] = nullptr; bs[ch] = tr->traj;
                    } else {
                        src[ch] = tr->out + ((long long)(j - ctx) * Cg + q) * HW;
                        gdst[ch] = tr->g_out + ((long long)(j - ctx) * Cg + q) * HW;
                        bs[ch] = tr->traj_out;
                    }
                }
            }
            const int jl = t - 1;  // residual source = last prognostic frame of the window
            if (jl < ctx) { tr->calls[k].res = tr->x + (long long)jl * tr->frame; tr->calls[k].res_bs = tr->traj; }
            else {
                tr->calls[k].res = tr->out + (long long)(jl - ctx) * tr->frame; tr->calls[k].res_bs = tr->traj_out;
                tr->calls[k].gres = tr->g_out + (long long)(jl - ctx) * tr->frame; tr->calls[k].gres_bs = tr->traj_out;
            }
        }
    }
    tr->h_src = src; tr->h_gdst = gdst; tr->h_bs = bs;
    DLWP_HIP(hipMemcpy(tr->src_tab, src.data(), src.size() * sizeof(float*), hipMemcpyHostToDevice));
    DLWP_HIP(hipMemcpy(tr->gdst_tab, gdst.data(), gdst.size() * sizeof(float*), hipMemcpyHostToDevice));
    DLWP_HIP(hipMemcpy(tr->bstride_tab, bs.data(), bs.size() * sizeof(long long), hipMemcpyHostToDevice));
    return DLWP_OK;
}

extern "C" int dlwp_fno_trainer_bind_aux(dlwp_fno_trainer* tr, const float* constants, const float* prescribed) {
    DLWP_REQUIRE(tr, DLWP_E_INVALID, "fno_trainer_bind_aux: NULL trainer");
    tr->constants = constants;
    tr->prescribed = prescribed;
    return DLWP_OK;
}

extern "C" void dlwp_fno_trainer_destroy(dlwp_fno_trainer* tr) {
    if (!tr) return;
    if (tr->graph_exec) (void)hipGraphExecDestroy(tr->graph_exec);
    if (tr->graph) (void)hipGraphDestroy(tr->graph);
    if (tr->cap_stream) (void)hipStreamDestroy(tr->cap_stream);
    dlwp_fno_plan_destroy(tr->plan);
    void* bufs[] = {tr->slab_skip, tr->slab_lift, tr->slab_proj, tr->g_out, tr->h0, tr->pre, tr->xhat, tr->x1, tr->spec,
                    tr->gA, tr->gB, (void*)tr->src_tab, (void*)tr->gdst_tab, tr->bstride_tab, tr->xin, tr->zl, tr->al, tr->zp,
                    tr->ap, tr->gz, tr->gyb, tr->gxin};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    delete tr;
}


extern "C" int dlwp_fno_trainer_bind(dlwp_fno_trainer* tr, float* params, float* grads) {
    DLWP_REQUIRE(tr && params, DLWP_E_INVALID, "fno_trainer_bind: NULL argument");
    if (tr->graph_exec && (params != tr->params || grads != tr->grads)) {
        (void)hipGraphExecDestroy(tr->graph_exec);
        (void)hipGraphDestroy(tr->graph);
        tr->graph_exec = nullptr;
        tr->graph = nullptr;
    }
    tr->params = params;
    tr->grads = grads;
    return DLWP_OK;
}

extern "C" int dlwp_fno_trainer_forward(dlwp_fno_trainer* tr, int keep_activations, void* stream) {
    DLWP_REQUIRE(tr && tr->params && tr->x && tr->out, DLWP_E_INVALID, "fno_trainer_forward: parameters / io not bound");
    return enqueue_forward(tr, keep_activations != 0, (hipStream_t)stream);
}

extern "C" int dlwp_fno_trainer_backward(dlwp_fno_trainer* tr, const float* grad_out, void* stream) {
    DLWP_REQUIRE(tr && tr->params && tr->grads && tr->x && tr->out, DLWP_E_INVALID,
                 "fno_trainer_backward: parameters/grads/io not bound");
    DLWP_REQUIRE(grad_out || (tr->y && tr->loss), DLWP_E_INVALID, "fno_trainer_backward: fused MSE needs y and loss buffers");
    return enqueue_loss_backward(tr, grad_out, (hipStream_t)stream);
}

extern "C" int dlwp_fno_trainer_fwd_bwd(dlwp_fno_trainer* tr, int use_graph, void* stream_) {
    DLWP_REQUIRE(tr && tr->params && tr->grads && tr->x && tr->y && tr->out && tr->loss, DLWP_E_INVALID,
                 "fno_trainer_fwd_bwd: parameters/grads/io not bound");
    hipStream_t s = (hipStream_t)stream_;
    int rc;
    if (!use_graph) {
        if ((rc = enqueue_forward(tr, true, s, true))) return rc;
        return enqueue_loss_backward(tr, nullptr, s, true);
    }
    if (!tr->graph_exec) {
        if (!tr->cap_stream) DLWP_HIP(hipStreamCreateWithFlags(&tr->cap_stream, hipStreamNonBlocking));
        DLWP_HIP(hipStreamBeginCapture(tr->cap_stream, hipStreamCaptureModeThreadLocal));
        rc = enqueue_forward(tr, true, tr->cap_stream, true);
        if (!rc) rc = enqueue_loss_backward(tr, nullptr, tr->cap_stream, true);
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamEndCapture(tr->cap_stream, &graph);
        if (rc) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        if (e != hipSuccess) {
            dlwp_set_error("fno_trainer: hipStreamEndCapture -> %s", hipGetErrorString(e));
            return DLWP_E_HIP;
        }
        tr->graph = graph;
        DLWP_HIP(hipGraphInstantiate(&tr->graph_exec, tr->graph, nullptr, nullptr, 0));
    }
    DLWP_HIP(hipGraphLaunch(tr->graph_exec, s));
    return DLWP_OK;
}

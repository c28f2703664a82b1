/* libdlwpmi — MI355X (gfx950) native kernels for the dlwp-benchmark autoregressive rollout
 * training path.  C ABI: plain pointers and sizes, no torch types.
 *
 * The reference (amazon-science/dlwp-benchmark) has no FFI of its own: its boundary is the
 * Python nn.Module surface.  Each entry point below therefore names the
 * reference call it replaces (file:line under /root/reference/src); the Python host side
 * in dlwp_benchmark_amd/ binds them with ctypes and re-exposes the reference's module
 * constructors / forward signatures.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed from the caller (never freed or retained
 *     beyond the call, except buffers explicitly bound to a trainer);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued there, no hidden
 *     synchronisation (graph-capture safe) unless a function says otherwise;
 *   - return 0 on success, negative DLWP_E_* on failure; dlwp_last_error() gives the
 *     thread-local message;
 *   - arithmetic: IEEE fp32 on the exact-f32 MFMA by default (the reference's fp32 path: the FNO rollout kernels, the GEMMs
 *     under dlwp_set_gemm_precision(0), the FFT and SHT kernels without a _bf16 suffix); bf16 OPERANDS with fp32 accumulation,
 *     epilogues and statistics where the reference trains under bf16 autocast (BASELINE configs C3-C5): dlwp_set_gemm_precision(1),
 *     the *_mixed GEMM entries (bf16 arrays in HBM), and the entry points that say so (dlwp_sfno_tail_*, dlwp_sfno_encode_* /
 *     _decode_*, dlwp_sht_*_bf16, dlwp_wgrad_segments);
 *   - scratch: entry points that need a workspace take it from the caller (dlwp_*_workspace_bytes + a pointer).  Two older ones
 *     keep a library-owned slab per device instead (the sliced weight-gradient GEMM, dlwp_sumsq's partials): grow-only,
 *     allocated outside stream captures only, never freed or re-used for anything else while the process lives.
 */
#ifndef DLWPMI_H
#define DLWPMI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DLWPMI_VERSION 100

#define DLWP_OK 0
#define DLWP_E_INVALID (-1)
#define DLWP_E_HIP (-2)
#define DLWP_E_UNSUPPORTED (-3)
#define DLWP_E_NOMEM (-4)

int dlwp_version(void);
const char* dlwp_last_error(void);

/* Measurement knobs (csrc/tuning.hip): every override of a dispatch heuristic -- kernel family, tile size, workgroup count -- is a   */
/* named entry of ONE registry.  dlwp_set_tuning(name, value) sets an override (name with or without the "DLWP_" prefix of the     */
/* environment variable of the same meaning), dlwp_clear_tuning(name) removes it (NULL: all); while no override is set the            */
/* environment variable DLWP_<NAME> is consulted at the time of use.  Unset = the library's own choice.  No knob changes results     */
/* beyond summation order.  dlwp_get_tuning returns 1 and the value when an override (either kind) is in force, else 0;             */
/* dlwp_tuning_list(i, &name, &doc) enumerates the registry (returns 0 past the end).                                               */
int dlwp_set_tuning(const char* name, int value);
int dlwp_clear_tuning(const char* name);
int dlwp_get_tuning(const char* name, int* value);
int dlwp_tuning_list(int index, const char** name, const char** doc);

/* Live per-kernel accounting (csrc/prof.hip; measurement only, no reference counterpart: the reference times whole epochs with    */
/* time.time(), src/nsbench/scripts/train.py:108,164).  Between dlwp_prof_enable(1) and dlwp_prof_collect() every instrumented      */
/* launch (the GEMM families, window attention, LayerNorm, segmented weight gradients, FFT passes) is bracketed by two HIP events    */
/* on its own stream and recorded with its kernel name, algorithmic flops and algorithmic HBM bytes (from the launch arguments).    */
/* Not recorded under stream capture.  dlwp_prof_collect() blocks until the recorded events have completed, folds the records by    */
/* kernel name, sorts by total time and returns the number of rows; dlwp_prof_get(i, ...) reads row i (name truncated to name_len). */
/* dlwp_prof_enable(2): the GEMM rows also carry the product's shape and epilogue in their name (one row per distinct product).    */
int dlwp_prof_enable(int on);
int dlwp_prof_collect(void);
int dlwp_prof_get(int index, char* name, int name_len, long long* calls, double* ms, double* flops, double* bytes);

/* ------------------------------------------------------------------------------------ */
/* Strided / gathered channel views.  A rollout step reads its input channels straight   */
/* out of the trajectory buffers (observations or earlier predictions) instead of        */
/* re-stacking them (reference: th.stack/th.cat per step, nsbench/models/fno/fno.py:     */
/* 228-237).  Either (base,bstride,cstride) or a per-channel pointer table is used.      */
typedef struct dlwp_chan_src {
    const float* base;
    long long bstride, cstride;        /* elements */
    const float* const* tab;           /* device array [C] of sample-0 plane pointers    */
    const long long* tab_bstride;      /* device array [C] of batch strides (elements)   */
} dlwp_chan_src;
typedef struct dlwp_chan_dst {
    float* base;
    long long bstride, cstride;
    float* const* tab;                 /* entries may be NULL: channel skipped           */
    const long long* tab_bstride;
} dlwp_chan_dst;

/* ------------------------------------------------------------------------------------ */
/* Pointwise 2-layer channel MLP: y = W2 gelu(W1 x + b1) + b2 on [B,C,P] fields.         */
/* Replaces neuralop FNO.lifting / FNO.projection (constructed at nsbench/models/fno/    */
/* fno.py:19-27,205-215; dlwpbench/models/fno/fno.py:38-47).  w1:[Ch,Cin] b1:[Ch]        */
/* w2:[Cout,Ch] b2:[Cout] (Conv2d 1x1 weight layout).                                    */
int dlwp_pwmlp_fwd(const float* x, const float* w1, const float* b1, const float* w2,
                   const float* b2, float* y, int B, int Cin, int Ch, int Cout, int P,
                   void* stream);
/* gx may be NULL. gw1,gb1,gw2,gb2 are ACCUMULATED into (caller zeroes them).            */
int dlwp_pwmlp_bwd(const float* x, const float* w1, const float* b1, const float* w2,
                   const float* gy, float* gx, float* gw1, float* gb1, float* gw2, float* gb2,
                   int B, int Cin, int Ch, int Cout, int P, void* stream);

/* Variant used inside the rollout: parameter gradients are written to a per-workgroup     */
/* partial slab of dlwp_pwmlp_slab_floats() floats (plain stores when slab_accumulate == 0, */
/* read-modify-write by the owning workgroup otherwise: deterministic, not bound by the     */
/* chip-wide float-atomic rate) and folded into the gradients by dlwp_pwmlp_slab_fold.      */
long long dlwp_pwmlp_slab_floats(int B, int Cin, int Ch, int Cout, int P);
int dlwp_pwmlp_bwd_slab(const float* x, const float* w1, const float* b1, const float* w2,
                        const float* gy, float* gx, float* slab, int slab_accumulate, int B,
                        int Cin, int Ch, int Cout, int P, void* stream);
int dlwp_pwmlp_slab_fold(const float* slab, float* gw1, float* gb1, float* gw2, float* gb2, int B,
                         int Cin, int Ch, int Cout, int P, void* stream);

/* ------------------------------------------------------------------------------------ */
/* FNO block: pre = irfft2(W . trunc(rfft2(act(x)))) + Wskip act(x) + bias               */
/* (neuralop FNOBlocks: SpectralConv + linear skip; call sites        */
/* `self.fno(x_t)` nsbench/models/fno/fno.py:38,96,246, dlwpbench/models/fno/fno.py:103). */
/* The mode-truncated transform is a pruned DFT (two small GEMMs per axis) fused with     */
/* the skip GEMM; no full spectrum is ever formed.                                        */
/*   x, pre : [B,C,H,W];  act_in!=0 applies GELU to x on load (x is then the previous     */
/*            block's pre-activation);                                                    */
/*   wspec  : [m1][m2c][Cin][Cout][2] (mode-major complex; m2c = n_modes[1]/2+1);         */
/*   wskip  : [Cout][Cin]; bias : [C];                                                    */
/*   xhat   : [B][m1][m2c][C][2] truncated spectrum of act(x), saved for backward.        */
typedef struct dlwp_fno_plan dlwp_fno_plan;
int dlwp_fno_plan_create(int C, int H, int W, int m1, int m2c, dlwp_fno_plan** out);
/* 3-D (time, y, x) blocks on [B,C,T,H,W] volumes (nsbench FNOContextModule / neuralop FNO with three n_modes): the plan  */
/* treats the volume as a (T*H) x W image with m0*m1 kept row frequencies (separable product twiddles); x, pre:            */
/* [B,C,T*H,W]; wspec [m0*m1][m2c][Cin][Cout][2]; xhat [B][m0*m1][m2c][C][2].                                             */
int dlwp_fno_plan_create3d(int C, int T, int H, int W, int m0, int m1, int m2c, dlwp_fno_plan** out);
void dlwp_fno_plan_destroy(dlwp_fno_plan* plan);
/* bytes of scratch the block calls need for batch B (caller allocates, 256-B aligned)   */
size_t dlwp_fno_block_workspace_bytes(const dlwp_fno_plan* plan, int B);
int dlwp_fno_block_fwd(const dlwp_fno_plan* plan, const float* x, int act_in, const float* wspec,
                       const float* wskip, const float* bias, float* pre, float* xhat, int B,
                       void* workspace, void* stream);
/* g_x = d loss / d x (includes the GELU derivative when act_in). g_wspec/g_wskip/g_bias */
/* are ACCUMULATED into.                                                                  */
int dlwp_fno_block_bwd(const dlwp_fno_plan* plan, const float* x, int act_in, const float* wspec,
                       const float* wskip, const float* g_pre, const float* xhat, float* g_x,
                       float* g_wspec, float* g_wskip, float* g_bias, int B, void* workspace,
                       void* stream);

/* ------------------------------------------------------------------------------------ */
/* Training-step pieces (nsbench/scripts/train.py:113-131: MSELoss, Adam).               */
/* loss_out (device, 1 float) += sum((a-b)^2) * scale                                    */
int dlwp_sqerr_sum(const float* a, const float* b, long long n, float scale, float* loss_out,
                   void* stream);
/* nn.MSELoss forward + backward in one pass (train.py:113,119-122):                      */
/* loss_out (device, 1 float) += mean((pred-target)^2); grad = 2 (pred-target) / n        */
int dlwp_mse_fwd_bwd(const float* pred, const float* target, long long n, float* loss_out,
                     float* grad, void* stream);
/* Evaluation metrics (nsbench/scripts/evaluate.py:232-257: RMSE / accumulated error per time */
/* step; dlwpbench/scripts/evaluate.py:494-546: latitude-weighted RMSE and ACC per (time,      */
/* variable)).  out/target/climatology are [B][G][H][W]; row_weights [H] or NULL (= 1);         */
/* climatology may be NULL.  moments [5][G] (device) is ACCUMULATED into:                       */
/*   sum w (o-t)^2 | sum w |o-t| | sum w (o-c)(t-c) | sum w (o-c)^2 | sum w (t-c)^2             */
int dlwp_error_moments(const float* out, const float* target, const float* climatology,
                       const float* row_weights, int B, int G, int H, int W, float* moments,
                       void* stream);
/* torch.optim.Adam (no weight decay / amsgrad) on a flat buffer; `step` is a device      */
/* int32 counter incremented by the kernel; grads are multiplied by grad_scale first      */
/* (1/world_size after a sum all-reduce) and zeroed afterwards when zero_grad != 0.       */
int dlwp_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int* step,
                   long long n, float lr, float beta1, float beta2, float eps, float grad_scale,
                   int zero_grad, void* stream);
/* The same step with torch.nn.utils.clip_grad_norm_ folded in (dlwpbench scripts/train.py:133-135 clips at  */
/* max_norm = learning rate before optimizer.step()): sumsq (device float, sum(g^2) of the UNSCALED gradient */
/* as dlwp_sumsq leaves it; NULL = no clipping) gives the coefficient min(1, max_norm / (sqrt(*sumsq) *       */
/* |grad_scale| + 1e-6)) that multiplies the gradient as Adam reads it -- the gradient buffer is not         */
/* rewritten (no dlwp_clip_scale pass): use it when the clipped gradient itself is not needed afterwards.    */
int dlwp_adam_step_clipped(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int* step,
                           long long n, float lr, float beta1, float beta2, float eps, float grad_scale,
                           int zero_grad, const float* sumsq, float max_norm, void* stream);
/* out (device float) = sum(g^2)   (for clip_grad_norm_, train.py:123-125)                */
int dlwp_sumsq(const float* g, long long n, float* out, void* stream);
/* g *= min(1, max_norm / (sqrt(*sumsq * grad_scale^2) + 1e-6))  (torch clip_grad_norm_)  */
int dlwp_clip_scale(float* g, long long n, const float* sumsq, float grad_scale, float max_norm,
                    void* stream);

/* ------------------------------------------------------------------------------------ */
/* Fused FNO rollout trainer: the whole autoregressive rollout forward, MSE, and BPTT     */
/* backward of nsbench TFNO2DModule / FNOModule (fno.py:29-41,217-250) and of the         */
/* residual dlwpbench form (dlwpbench/models/fno/fno.py:64-106), captured into one        */
/* hipGraph.  Parameters/gradients live in ONE flat fp32 buffer owned by the caller       */
/* (layout: dlwp_fno_param_offset); the gradient buffer is what data-parallel ranks        */
/* all-reduce (one flat RCCL bucket).                                                     */
typedef struct dlwp_fno_cfg {
    int B, T, D;            /* batch, sequence length, channels per frame (ns: D=1)       */
    int H, W;
    int context_size;       /* ctx frames flattened into channels (TFNO2DModule); 0/1:    */
                            /* FNOModule single-frame form                                 */
    int teacher_forcing_steps;
    int hidden, lifting, projection, n_layers;
    int m1, m2c;            /* kept modes: n_modes[0] rows, n_modes[1]/2+1 columns        */
    int out_channels;       /* = D                                                         */
    int form;               /* DLWP_FNO_FORM_NS: nsbench fno.py:217-250 (x[B,T,D,H,W] -> [B,T,D,H,W]);    */
                            /* DLWP_FNO_FORM_DLWP: dlwpbench fno.py:64-106 (D = prognostic channels,      */
                            /* out[k] = last frame + net([constants|prescribed|prognostic window]),       */
                            /* k = 0..T-ctx-1, teacher_forcing_steps unused)                              */
    int constant_channels;  /* dlwp form: channels of constants [B,1,Cc,H,W]                              */
    int prescribed_channels;/* dlwp form: channels of prescribed [B,T,Cp,H,W]                             */
    int m0;                 /* DLWP_FNO_FORM_NS_CONTEXT3D: kept TEMPORAL modes n_modes[0] (m1 = n_modes[1] rows,  */
                            /* m2c = n_modes[2]/2+1 columns); 0 for the 2-D forms                                 */
} dlwp_fno_cfg;
#define DLWP_FNO_FORM_NS 0
#define DLWP_FNO_FORM_DLWP 1
/* nsbench FNOContextModule (fno.py:44-100): the context window [B, ctx, D, H, W] is a [B, D, ctx, H, W] VOLUME for a     */
/* 3-D (time, y, x) FNO and the last time slice of the result is the prediction; context_size = n_modes[0] (:54).         */
/* The (time, y) transform pair is one dense separable DFT: the block kernels see a (ctx*H) x W image with m0*m1 kept     */
/* "row" frequencies whose twiddles are products e^{-2 pi i (kt t/ctx + ky h/H)}; spectral weights [m0*m1][m2c][C][C][2].  */
#define DLWP_FNO_FORM_NS_CONTEXT3D 2

enum {
    DLWP_FNO_P_LIFT_W1 = 0, DLWP_FNO_P_LIFT_B1, DLWP_FNO_P_LIFT_W2, DLWP_FNO_P_LIFT_B2,
    DLWP_FNO_P_PROJ_W1, DLWP_FNO_P_PROJ_B1, DLWP_FNO_P_PROJ_W2, DLWP_FNO_P_PROJ_B2,
    DLWP_FNO_P_SPEC_W,   /* + layer: [m1][m2c][C][C][2]  */
    DLWP_FNO_P_SKIP_W,   /* + layer: [C][C]              */
    DLWP_FNO_P_SPEC_B    /* + layer: [C]                 */
};
/* element offset and size of a parameter tensor inside the flat buffer; total = offset   */
/* returned for kind = -1.                                                                */
long long dlwp_fno_param_offset(const dlwp_fno_cfg* cfg, int kind, int layer, long long* size);

typedef struct dlwp_fno_trainer dlwp_fno_trainer;
int dlwp_fno_trainer_create(const dlwp_fno_cfg* cfg, dlwp_fno_trainer** out);
void dlwp_fno_trainer_destroy(dlwp_fno_trainer* tr);
/* bind the caller's trajectory buffers (borrowed): x [B,T,D,H,W] observations, y targets  */
/* (may be NULL when only forward / backward(grad_out) are used), out predictions, loss     */
/* (1 device float: mean squared error, may be NULL likewise).  Synchronous (uploads the    */
/* channel gather tables); invalidates a captured graph.                                    */
int dlwp_fno_trainer_bind_io(dlwp_fno_trainer* tr, const float* x, const float* y, float* out,
                             float* loss);
/* dlwp form only, BEFORE bind_io: constants [B,1,Cc,H,W] and prescribed [B,T,Cp,H,W]     */
/* (NULL when the corresponding channel count is 0).  In the dlwp form x is the prognostic */
/* trajectory [B,T,Cg,H,W] and y/out are [B,T-ctx,Cg,H,W].                                 */
int dlwp_fno_trainer_bind_aux(dlwp_fno_trainer* tr, const float* constants, const float* prescribed);
/* bind the caller's flat parameter / gradient buffers (borrowed until destroy/rebind)    */
int dlwp_fno_trainer_bind(dlwp_fno_trainer* tr, float* params, float* grads);
/* rollout forward into the `out` buffer; keep_activations!=0 stores every net call's      */
/* activations for a following dlwp_fno_trainer_backward (0: evaluation)                   */
int dlwp_fno_trainer_forward(dlwp_fno_trainer* tr, int keep_activations, void* stream);
/* BPTT backward of the last kept forward.  grad_out = d loss / d out [B,T,D,H,W], or NULL */
/* for the fused nn.MSELoss(mean) against the `y` buffer (loss value -> `loss` buffer).    */
/* Parameter gradients are ACCUMULATED into the bound gradient buffer.                     */
int dlwp_fno_trainer_backward(dlwp_fno_trainer* tr, const float* grad_out, void* stream);
/* forward(keep) + backward(NULL) in one call.                                             */
/* use_graph!=0 captures the sequence into a hipGraph on first use and replays it.        */
int dlwp_fno_trainer_fwd_bwd(dlwp_fno_trainer* tr, int use_graph, void* stream);

/* ------------------------------------------------------------------------------------ */
/* AFNO2D spectral token mixer (FourCastNet): y = x + irfft2(softshrink(MLP_blockdiag(    */
/* rfft2(x)))) on channels-last x [B,H,W,C].  Replaces AFNO2D.forward                     */
/* nsbench/models/fourcastnet/fourcastnet.py:77-126 (dlwpbench twin :78-127), including   */
/* its kept-mode window computed from H only (:92-93).  Parameter layouts are the         */
/* reference's: w1,w2 [2,nb,bs,bs] (re/im), b1,b2 [2,nb,bs]; hidden_size_factor = 1.      */
/* xsave: caller buffer of dlwp_afno2d_save_elems() floats holding the kept spectrum for   */
/* the backward pass.  Round-1 limits: block size <= 16, H*kept_cols*bs <= 10240 (16x16,   */
/* 32x64 grids); larger grids return DLWP_E_UNSUPPORTED.                                   */
long long dlwp_afno2d_save_elems(int B, int H, int W, int C, int nb, float hard_thresholding_fraction);
int dlwp_afno2d_fwd(const float* x, const float* w1, const float* b1, const float* w2,
                    const float* b2, float* y, float* xsave, int B, int H, int W, int C, int nb,
                    float sparsity_threshold, float hard_thresholding_fraction, void* stream);
/* The same with the block's outer skip folded in (Block.forward, fourcastnet.py:156-160:   */
/* `x = filter(norm1(x)); x = x + residual`): y = AFNO2D(x) + residual, residual [B,H,W,C]   */
/* nullable.  Its gradient is the upstream gradient itself; dlwp_afno2d_bwd is unchanged.    */
int dlwp_afno2d_fwd_res(const float* x, const float* residual, const float* w1, const float* b1,
                        const float* w2, const float* b2, float* y, float* xsave, int B, int H,
                        int W, int C, int nb, float sparsity_threshold,
                        float hard_thresholding_fraction, void* stream);
/* gx = d loss/d x (includes the residual path); gw1,gb1,gw2,gb2 are ACCUMULATED into.     */
int dlwp_afno2d_bwd(const float* gy, const float* xsave, const float* w1, const float* b1,
                    const float* w2, const float* b2, float* gx, float* gw1, float* gb1,
                    float* gw2, float* gb2, int B, int H, int W, int C, int nb,
                    float sparsity_threshold, float hard_thresholding_fraction, void* stream);

/* ------------------------------------------------------------------------------------ */
/* Fused window attention core (Swin W-MSA / SW-MSA and Pangu EarthAttention3D): out =     */
/* softmax(scale q k^T + bias + mask) v without materialising the [windows,heads,N,N]      */
/* scores.  Replaces the middle of WindowAttention.forward nsbench/models/swintransformer/  */
/* swin_transformer.py:134-152 (dlwpbench :133-151) and of EarthAttention3D.forward         */
/* dlwpbench/models/panguweather/panguweather.py:186-208; qkv / proj Linear stay outside.   */
/*   qkv    : [B_, N, 3, heads, d]  (the reference's reshape of the qkv Linear output)      */
/*   bias_table : [TB, ntypes, heads]; the bias of (query q, key k) is                       */
/*            table[ia[q] + ib[k]][window % ntypes][head]  (ia, ib: int32 [N]).  Swin:        */
/*            ntypes = 1, ia = y(2Ww-1)+x, ib = (Wh-1-y)(2Ww-1)+(Ww-1-x); Pangu: see          */
/*            utils/earth_position_index.py:4-45 (index additive in query and key).            */
/*   labels : [nW, N] int32 region labels of the shifted-window mask or NULL; the mask       */
/*            value is -100 where labels differ (swin_transformer.py:377-395)                */
/*   out    : [B_, N, heads*d];  lse : [B_, heads, N] (saved for backward)                   */
/* B_ = batch*nW with the window index fastest.  head_dim d <= 64.                           */
int dlwp_window_attn_fwd(const float* qkv, const float* bias_table, const int* ia, const int* ib,
                         const int* labels, float* out, float* lse, int B_, int nW, int N, int TB,
                         int ntypes, int heads, int d, float scale, void* stream);
/* gqkv is written; gbias_table is ACCUMULATED into; dsum: scratch [B_, heads, N].           */
/* slab: optional scratch of dlwp_window_attn_bwd_slab_floats() floats -- the workgroups then */
/* store their bias-gradient partials there and a fold kernel sums them (one atomic per       */
/* FOLD chunk instead of TB atomics per workgroup); NULL: direct atomics.                     */
/* Earth-specific tables (Pangu: [TB][window types][heads], a head's slice strided by types*heads floats) make   */
/* every workgroup gather its slice one cache line per entry.  pack_table writes the transposed copy               */
/* packed[type][head][TB] once per call; the *_packed entries read contiguous slices from it (packed_table NULL:    */
/* the strided gather, as in the plain entries).  The gradient still goes to the reference-layout gbias_table.      */
int dlwp_window_attn_pack_table(const float* bias_table, float* packed, int TB, int ntypes, int heads,
                                void* stream);
int dlwp_window_attn_fwd_packed(const float* qkv, const float* bias_table, const float* packed_table,
                                const int* ia, const int* ib, const int* labels, float* out, float* lse,
                                int B_, int nW, int N, int TB, int ntypes, int heads, int d, float scale,
                                void* stream);
int dlwp_window_attn_bwd_packed(const float* qkv, const float* bias_table, const float* packed_table,
                                const int* ia, const int* ib, const int* labels, const float* out,
                                const float* lse, const float* gout, float* gqkv, float* gbias_table,
                                float* dsum, float* slab, int B_, int nW, int N, int TB, int ntypes,
                                int heads, int d, float scale, void* stream);
/* Windows whose tokens outside [q_lo, q_hi) are padding that the caller crops (Pangu's pressure-level pad,           */
/* src/dlwpbench/models/panguweather/panguweather.py:283-317: one plane of every (2,7,7) window): the padded tokens  */
/* stay keys / values, their own rows are skipped where the kernel family can (out / lse rows outside the range are  */
/* then unwritten; gout rows outside it are taken as zero; their gqkv query part is written as zero).                */
int dlwp_window_attn_fwd_qrange(const float* qkv, const float* bias_table, const float* packed_table,
                                const int* ia, const int* ib, const int* labels, float* out, float* lse,
                                int B_, int nW, int N, int TB, int ntypes, int heads, int d, float scale,
                                int q_lo, int q_hi, void* stream);
int dlwp_window_attn_bwd_qrange(const float* qkv, const float* bias_table, const float* packed_table,
                                const int* ia, const int* ib, const int* labels, const float* out,
                                const float* lse, const float* gout, float* gqkv, float* gbias_table,
                                float* dsum, float* slab, int B_, int nW, int N, int TB, int ntypes,
                                int heads, int d, float scale, int q_lo, int q_hi, void* stream);

/* bf16 tensors in the window layout (round 6; reference: WindowAttention.forward, swin_transformer.py:123-155, under torch.autocast      */
/* the qkv projection's output, the attention output and their gradients are bf16 tensors there too): qkv [B_, N, 3, heads, d], out and */
/* gout [B_, N, heads, d] and gqkv are bf16 arrays; lse, the bias table and its gradient stay fp32.  Whole query range only.  Supported */
/* where dlwp_window_attn_io_bf16_supported(N, d, TB, B_ * heads) returns 1 (windows of at most 64 tokens, head_dim % 4 == 0, bf16      */
/* matrix mode); otherwise DLWP_E_UNSUPPORTED.                                                                                           */
int dlwp_window_attn_io_bf16_supported(int N, int d, int TB, long long pairs);
int dlwp_window_attn_fwd_bf16(const void* qkv, const float* bias_table, const float* packed_table, const int* ia, const int* ib,
                              const int* labels, void* out, float* lse, int B_, int nW, int N, int TB, int ntypes, int heads, int d,
                              float scale, void* stream);
int dlwp_window_attn_bwd_bf16(const void* qkv, const float* bias_table, const float* packed_table, const int* ia, const int* ib,
                              const int* labels, const void* out, const float* lse, const void* gout, void* gqkv, float* gbias_table,
                              int B_, int nW, int N, int TB, int ntypes, int heads, int d, float scale, void* stream);
long long dlwp_window_attn_bwd_slab_floats(int B_, int N, int heads, int TB);
/* Backward of  reverse(crop) . attention . partition(pad with `fill`)  in ONE launch, for a block whose qkv projection ran */
/* on the real tokens (EarthSpecificBlock.forward, src/dlwpbench/models/panguweather/panguweather.py:283-317: ZeroPad3d ->   */
/* roll -> window_partition -> attention -> window_reverse -> roll -> crop3d; SwinTransformerBlock likewise).  The upstream   */
/* gradient and the qkv gradient stay in the UNPARTITIONED token layout:                                                     */
/*   gout_tokens [B][Ltok][heads*d]   window position n of window w (of nW per sample) reads token dst_map[w*N + n]          */
/*                                     (-1: the position is cropped, zero upstream gradient);                                */
/*   gqkv_tokens [B][Ltok][3*heads*d] written at token src_map[w*N + n]; positions with src_map < 0 are padding that held    */
/*                                     `fill` (the qkv bias): their gradient is summed into gfill [3*heads*d] (ACCUMULATED).  */
/* Both maps are int32 [nW][N], the same for every sample; src_map must hit every token exactly once (constant padding,      */
/* no circular copies).  fill == NULL: qkv / out are the window-layout tensors the forward produced                          */
/* (dlwp_window_attn_fwd_qrange on a gathered qkv).  fill != NULL ([3*heads*d], the qkv bias): the OPERANDS are in the token   */
/* layout too -- qkv [B][Ltok][3*heads*d] read through src_map (padded positions hold fill), out [B][Ltok][heads*d] through   */
/* dst_map, as dlwp_window_attn_fwd_tokens left them; lse stays [B_][heads][N].                                              */
/* Replaces the gather of gout, dlwp_window_attn_bwd_qrange, dlwp_window_scatter and dlwp_window_pad_colsum of that chain.   */
/* Needs the bf16 matrix mode, N <= 128, d <= 32, d % 4 == 0 (dlwp_window_attn_bwd_tokens_supported); otherwise              */
/* DLWP_E_UNSUPPORTED and nothing is launched.                                                                               */
int dlwp_window_attn_bwd_tokens_supported(int N, int d, int TB);
int dlwp_window_attn_bwd_tokens(const float* qkv, const float* fill, const float* bias_table, const float* packed_table,
                                const int* ia, const int* ib, const int* labels, const float* out, const float* lse,
                                const float* gout_tokens, const int* dst_map, const int* src_map, float* gqkv_tokens,
                                float* gfill, float* gbias_table, int B_, int nW, int N, int Ltok, int TB, int ntypes,
                                int heads, int d, float scale, int q_lo, int q_hi, int io_bf16, void* stream);
/* io_bf16 != 0 (both entries; needs fill != NULL): the four token-layout tensors -- qkv, out, gout_tokens, gqkv_tokens -- are  */
/* bf16 arrays behind the float pointers (the qkv projection wrote bf16, the output projection reads bf16, and their backward   */
/* GEMMs take the gradients as bf16 operands without a cast pass); fill / gfill / tables / lse stay fp32.                       */
/* The forward of the same chain in one launch: attention over the windows of a token-layout qkv tensor (no gathered copy),   */
/* output rows written to the tokens dst_map names (positions it drops are computed only as keys / values).  Same shape       */
/* family as the wave-per-window kernels: N <= 128, d <= 32, d % 4 == 0, at least 2048 (window, head) pairs, bf16 matrix mode  */
/* (dlwp_window_attn_fwd_tokens_supported); otherwise DLWP_E_UNSUPPORTED.  lse rows outside the computed query chunks are     */
/* written as zero (the backward entry stages every row's statistic); out rows are written exactly once per token.            */
int dlwp_window_attn_fwd_tokens_supported(int N, int d, long long pairs);
int dlwp_window_attn_fwd_tokens(const float* qkv_tokens, const float* fill, const float* bias_table, const float* packed_table,
                                const int* ia, const int* ib, const int* labels, const int* src_map, const int* dst_map,
                                float* out_tokens, float* lse, int B_, int nW, int N, int Ltok, int TB, int ntypes, int heads,
                                int d, float scale, int q_lo, int q_hi, int io_bf16, void* stream);
int dlwp_window_attn_bwd(const float* qkv, const float* bias_table, const int* ia, const int* ib,
                         const int* labels, const float* out, const float* lse, const float* gout,
                         float* gqkv, float* gbias_table, float* dsum, float* slab, int B_, int nW,
                         int N, int TB, int ntypes, int heads, int d, float scale, void* stream);

/* Head dimensions above 64 (deep stages of the nsbench Swin U-Net: the shipped swintransformer.yaml reaches    */
/* head_dim 192 on its 4 x 4 token map): the host runs q k^T and p v as dlwp_gemm_batched around these two row      */
/* kernels, with the scores s / p [B_, heads, N, N] in HBM (N <= 1024).                                              */
/*   fwd (in place): p = softmax(scale s + table[ia[q] + ib[k]][window % ntypes][head] + mask)                      */
/*   bwd (in place): ds = p (dp - rowsum(p dp)); gbias_table += ds; dp <- scale ds                                  */
int dlwp_window_softmax_fwd(float* s, const float* bias_table, const int* ia, const int* ib, const int* labels,
                            int B_, int nW, int N, int ntypes, int heads, float scale, void* stream);
int dlwp_window_softmax_bwd(const float* p, float* dp, float* gbias_table, const int* ia, const int* ib, int B_,
                            int nW, int N, int ntypes, int heads, float scale, void* stream);

/* ------------------------------------------------------------------------------------ */
/* rFFT2 / irFFT2 of real fields of ANY size, butterflies staged in LDS (mixed-radix Stockham */
/* per axis: radices 4, 2, 3, 5, 7 and whatever prime is left, e.g. 721 = 7 x 103).  Replaces  */
/* torch.fft.rfft2 / irfft2(x, dim=(1, 2), norm="ortho") on channels-last [B, H, W, C] tensors  */
/* (AFNO2D.forward, nsbench/models/fourcastnet/fourcastnet.py:84,123; dlwpbench twin :85,124)   */
/* and rfftn / irfftn(norm="forward") over the last dims of [B, C, H, W] (neuralop SpectralConv, */
/* App. A-1).  layout 0: x [B][H][W][C] <-> X [B][H][W/2+1][C][2] (C even);                      */
/* layout 1: x [B][C][H][W] <-> X [B][C][H][W/2+1][2].  norm: 0 "backward", 1 "ortho",           */
/* 2 "forward" (torch.fft's meaning).  adjoint != 0 selects the transposed transform of the       */
/* OTHER direction with the same norm, i.e. the backward pass:                  */
/*   dlwp_irfft2(adjoint=1): gX (spectrum) -> gx = rfft2^H gX   (interior bins weigh one half)     */
/*   dlwp_rfft2 (adjoint=1): gx (field)    -> gX = irfft2^H gx  (interior bins weigh two)          */
/* DC / Nyquist imaginary parts are ignored by irfft2, as in torch.  work: scratch of X's size.    */
typedef struct dlwp_fft_plan dlwp_fft_plan;
int dlwp_fft_plan_create(int H, int W, dlwp_fft_plan** out);
void dlwp_fft_plan_destroy(dlwp_fft_plan* plan);
int dlwp_rfft2(const dlwp_fft_plan* plan, const float* x, float* X, int B, int C, int layout, int norm,
               int adjoint, void* stream);
int dlwp_irfft2(const dlwp_fft_plan* plan, const float* X, float* x, float* work, int B, int C, int layout,
                int norm, int adjoint, void* stream);
/* The channels-last pair with a WINDOW of the half spectrum as two planes,                          */
/* X [2 (re | im)][B][r1 - r0][c1][C] = rows r0 <= kh < r1, columns kw < c1 of rfft2's [B][H][W/2+1][C]: */
/* the kept modes AFNO2D's block-diagonal complex MLP works on (fourcastnet.py:85-124: rfft2, the slice */
/* x[:, total_modes-kept_modes:total_modes+kept_modes, :kept_modes], irfft2 of the zero-initialised     */
/* result), in the planar layout of the real batched GEMMs that replace the complex einsums -- no copy  */
/* or zero fill between transform and products.  irfft2_planar reads zeros outside the window; adjoint  */
/* as above.  work: scratch of the FULL half spectrum's size (2 B H (W/2+1) C floats).                  */
/* bs > 0 (a divisor of C): block-planar instead, X [B][r1 - r0][c1][C / bs][2 (re | im)][bs] -- with   */
/* dlwp_afno_wq_expand_bp's weights the complex block MLP of a layer is ONE real batched GEMM.          */
int dlwp_rfft2_planar(const dlwp_fft_plan* plan, const float* x, float* X, float* work, int B, int C, int r0, int r1,
                      int c1, int bs, int norm, int adjoint, void* stream);
/* residual (field-shaped [B][H][W][C], or NULL) is added to irfft2_planar's output: AFNO2D's `x + bias` skip    */
/* (fourcastnet.py:126) in the forward pass, the gradient arriving along that skip in the backward pass.      */
/* dlwp_rfft2_planar whose store folds in the derivative of a soft-shrink: a stored component is zeroed where the same   */
/* element of `mask` (X's layout: the saved pre-activation of AFNO2D's second block layer, fourcastnet.py:117-121          */
/* F.softshrink) has magnitude <= lam -- the backward transform of the filter then delivers the gradient of that           */
/* pre-activation directly (no dlwp_act_bwd pass).  mask NULL: dlwp_rfft2_planar.                                          */
int dlwp_rfft2_planar_masked(const dlwp_fft_plan* plan, const float* x, float* X, float* work, const float* mask, float lam,
                             int B, int C, int r0, int r1, int c1, int bs, int norm, int adjoint, void* stream);
/* The planar pair with the spectrum window as a bf16 array (flags = 1; bf16 storage of the AFNO block MLP's operands:  */
/* the reference's einsums run under bf16 autocast, fourcastnet.py:100-121): dlwp_rfft2_planar_ex writes X (and reads    */
/* the mask) as bf16, dlwp_irfft2_planar_ex reads X as bf16; fields, work and residuals stay fp32.  flags = 0: the      */
/* entries above.                                                                                                        */
int dlwp_rfft2_planar_ex(const dlwp_fft_plan* plan, const float* x, void* X, float* work, const void* mask, float lam, int B,
                         int C, int r0, int r1, int c1, int bs, int norm, int adjoint, int flags, void* stream);
int dlwp_irfft2_planar_ex(const dlwp_fft_plan* plan, const void* X, float* x, float* work, const float* residual,
                          const float* residual2, int B, int C, int r0, int r1, int c1, int bs, int norm, int adjoint,
                          int flags, void* stream);
int dlwp_irfft2_planar(const dlwp_fft_plan* plan, const float* X, float* x, float* work, const float* residual, int B,
                       int C, int r0, int r1, int c1, int bs, int norm, int adjoint, void* stream);
/* ... with a second field added in the same store (the AFNO block's outer skip around the filter,                 */
/* fourcastnet.py:156-165 `x = self.filter(x); if self.double_skip: x = x + residual`); residual2 needs residual.   */
int dlwp_irfft2_planar2(const dlwp_fft_plan* plan, const float* X, float* x, float* work, const float* residual,
                        const float* residual2, int B, int C, int r0, int r1, int c1, int bs, int norm, int adjoint,
                        void* stream);

/* General-grid AFNO2D (grids whose block spectrum does not fit LDS): the transforms run as     */
/* dlwp_gemm_batched against DFT tables and the per-mode block MLP as batched GEMMs over the     */
/* channel blocks, on the real image of the complex block weights w [2][nb][bs_in][bs_out]       */
/* (fourcastnet.py:70-75): wq[ri][ro][blk][i][o] = {Wr, Wi; -Wi, Wr}.  fold ACCUMULATES the     */
/* complex gradient from a wq-shaped gradient.                                                   */
int dlwp_afno_wq_expand(const float* w, float* wq, int nb, int bs_in, int bs_out, void* stream);
int dlwp_afno_wq_fold(const float* gq, float* gw, int nb, int bs_in, int bs_out, void* stream);
/* Block-planar form (tokens [T][blk][re | im][bs], dlwp_rfft2_planar with bs > 0): wq [blk][ri][i][ro][o], one  */
/* real 2 bs_in x 2 bs_out matrix per channel block, and the bias b [2][nb][bs_out] reordered to bq [blk][ro][o]; */
/* fold ACCUMULATES both gradients (gw, gb in the reference's parameter layouts, fourcastnet.py:70-75).           */
int dlwp_afno_wq_expand_bp(const float* w, const float* b, float* wq, float* bq, int nb, int bs_in, int bs_out, void* stream);
int dlwp_afno_wq_fold_bp(const float* gq, const float* gbq, float* gw, float* gb, int nb, int bs_in, int bs_out,
                         void* stream);

/* Window partition / reverse of shifted-window attention as one gather each (the reference  */
/* runs pad, roll, partition / reverse, roll, crop as separate full-tensor copies:             */
/* nsbench swin_transformer.py:213-250, dlwpbench panguweather.py:283-317).                    */
/* tokens x [B][D0][D1][D2][C] <-> windows [B*nW][w0*w1*w2][C]; all arrays have 3 entries       */
/* (use 1 for unused leading axes): dims D, padded sizes P (multiples of the window), front     */
/* pads, roll shifts s (rolled[i] = padded[(i + s) mod P], i.e. torch.roll by -s), window sizes, */
/* strides of the window-grid index inside a sample's nW windows, circular (1) or zero (0)       */
/* padding per axis.  scatter with sum_copies != 0 sums every padded copy of a token (adjoint    */
/* of a circularly padded gather).  C must be a multiple of 4.                                   */
int dlwp_window_gather(const float* x, float* windows, int B, int C, const int* dims,
                       const int* padded, const int* front, const int* shift, const int* window,
                       const long long* wstride, const int* circular, void* stream);
/* dlwp_window_gather on a tensor that a token-wise Linear layer has already produced: the padded positions hold  */
/* `fill` [C] (that layer's bias; NULL: zero) -- "pad, then Linear on every window token" == "Linear on the real    */
/* tokens, then pad with the bias" (reference: the qkv projection of EarthAttention3D on the zero-padded windows,   */
/* panguweather.py:283-292,176), so the GEMM runs on the real tokens only.  dlwp_window_pad_colsum is the fill's    */
/* adjoint: gfill[c] += sum over the padded positions of g_windows[..][c] for c >= c_lo (a multiple of 4; lets the   */
/* caller skip channels whose padded rows are known to be zero).                                                   */
int dlwp_window_gather_fill(const float* x, const float* fill, float* windows, int B, int C, const int* dims,
                            const int* padded, const int* front, const int* shift, const int* window,
                            const long long* wstride, const int* circular, void* stream);
int dlwp_window_pad_colsum(const float* g_windows, float* gfill, int B, int C, const int* dims,
                           const int* padded, const int* front, const int* shift, const int* window,
                           const long long* wstride, const int* circular, int c_lo, void* stream);
int dlwp_window_scatter(const float* windows, float* x, int B, int C, const int* dims,
                        const int* padded, const int* front, const int* shift, const int* window,
                        const long long* wstride, const int* circular, int sum_copies,
                        void* stream);
/* dlwp_window_scatter with the block's skip connection folded in: x = residual + reverse(windows)  */
/* (`x = shortcut + drop_path(x)` after window_reverse / roll / crop, swin_transformer.py:250-255 and */
/* panguweather.py:317-319); residual [B][D0*D1*D2][C] nullable.                                      */
int dlwp_window_scatter_add(const float* windows, const float* residual, float* x, int B, int C,
                            const int* dims, const int* padded, const int* front, const int* shift,
                            const int* window, const long long* wstride, const int* circular,
                            int sum_copies, void* stream);
/* Round 5: the gather / scatter pair on bf16 arrays and with the branch's stochastic depth folded in.            */
/* dlwp_window_gather_ex: dlwp_window_gather_fill (fill nullable) with flags bit 0 = x is a bf16 array, bit 1 =     */
/* windows is one, and scale [B] (nullable): the gathered values of sample b are multiplied by scale[b].            */
/* dlwp_window_scatter_ex: x[b] = residual[b] + scale[b] * reverse(windows)[b] (`x = shortcut + drop_path(x)` after */
/* window_reverse, swin_transformer.py:250-256): flags bit 0 = windows is a bf16 array, bit 1 = x is one; residual  */
/* (fp32, nullable) and scale [B] (fp32, nullable).  Each is the other's adjoint with the same scale.               */
int dlwp_window_gather_ex(const void* x, const float* fill, const float* scale, void* windows, int B, int C,
                          const int* dims, const int* padded, const int* front, const int* shift,
                          const int* window, const long long* wstride, const int* circular, int flags,
                          void* stream);
int dlwp_window_scatter_ex(const void* windows, const float* residual, const float* scale, void* x, int B, int C,
                           const int* dims, const int* padded, const int* front, const int* shift,
                           const int* window, const long long* wstride, const int* circular,
                           int sum_copies, int flags, void* stream);
/* Epilogue of a transposed convolution with kernel == stride on channels-last tokens (the Swin U-decoder's          */
/* nn.ConvTranspose2d(k, stride k) + GELU, src/nsbench/models/swintransformer/swin_transformer.py:580-588, dlwpbench  */
/* twin): y [B*H*W][O*kh*kw] is the GEMM x . W[Cin][O*kh*kw] in the weight's own layout; forward: dst[b][h kh+i][w kw+j] */
/* [coff+o] = act(y + bias[o]) with pixel pitch ctot (dst may be the concat buffer of the next level), act 0 none / 1     */
/* GELU; backward != 0: dst = gy [B*H*W][O*kh*kw] = gout * act'(y + bias), gbias[o] += its sums (nullable).               */
int dlwp_upconv_shuffle(const float* y, const float* bias, const float* gout, float* dst, float* gbias,
                        int B, int H, int W, int O, int kh, int kw, int ctot, int coff, int act,
                        int backward, void* stream);
/* Patch merging gather (PatchMerging.forward, src/nsbench/models/swintransformer/swin_transformer.py:291-312 */
/* and the dlwpbench twin): tokens x [B][H][W][C] -> [B][ceil(H/2)][ceil(W/2)][4C] with channel blocks in the    */
/* reference's concat order (0,0), (1,0), (0,1), (1,1) and zeros beyond an odd H / W (its constant pad).        */
/* backward != 0 runs the adjoint: src is the gradient [B][H2][W2][4C], dst the gradient [B][H][W][C].          */
int dlwp_patch_merge(const float* src, float* dst, int B, int H, int W, int C, int backward, void* stream);
/* One lead time of an autoregressive rollout on its sliding input window (csrc/rollout_ops.hip; the reference  */
/* rebuilds the window with stack + cat + add on every step: src/nsbench/models/fourcastnet/fourcastnet.py:     */
/* 262-300, swintransformer/swin_transformer.py:597-640, src/dlwpbench/models/unet/unet.py:64-111):              */
/*   next[b][j] = win[b][j+1] (j < ctx-1);  next[b][ctx-1] = out[b] = win[b][ctx-1] + delta[b].                  */
/* win: [B][ctx][frame] with a batch stride in floats (a slice of the data tensor qualifies); next (nullable:    */
/* last step) [B][ctx][frame]; out [B][frame].  delta_layout 0: delta is [B][frame]; 1: delta holds the patch    */
/* tokens of a linear head, [B][H/ph][W/pw][ph][pw][D] with frame = D*H*W (AFNONet.head output, :296-298).        */
int dlwp_window_advance_fwd(const float* win, long long win_batch_stride, const float* delta, float* next,
                            float* out, int B, int ctx, long long frame, int delta_layout, int D, int H,
                            int W, int ph, int pw, void* stream);
/* Its adjoint, which also sums the gradients that reach the window from its (up to) three readers: g_next from  */
/* the following advance [B][ctx][frame], g_net from the network that read the window (batch stride given: a     */
/* channel slice of a wider input gradient qualifies), g_out from the loss ([B][frame], batch stride given).      */
/* Any of the three may be NULL.  g_win (nullable: the window was data) [B][ctx][frame]; g_delta in delta's layout. */
int dlwp_window_advance_bwd(const float* g_next, const float* g_net, long long net_batch_stride,
                            const float* g_out, long long out_batch_stride, float* g_win, float* g_delta,
                            int B, int ctx, long long frame, int delta_layout, int D, int H, int W, int ph,
                            int pw, void* stream);

/* ------------------------------------------------------------------------------------ */
/* Token-level building blocks of the AFNO / Swin / Pangu blocks (nn.Linear, nn.LayerNorm, */
/* nn.GELU call sites: nsbench/models/fourcastnet/fourcastnet.py:44-46,213,233;             */
/* nsbench/models/swintransformer/swin_transformer.py:35-39,131,153,187,194,274).           */
/* C[M,N] (+)= epilogue(op(A)[M,K] . op(B)[K,N]); row-major with leading dimensions;         */
/* epilogue: + bias[n], optional store of the pre-activation, act (0 none, 1 GELU),          */
/* + residual[m,n] (same layout as C); accumulate != 0 adds into C.  rowsum (optional, [M]): */
/* rowsum[m] += sum_k op(A)[m,k] -- the bias gradient of a Linear layer comes out of the      */
/* weight-gradient product gW = gy^T x for free.  Reductions with few output tiles are split  */
/* along K across workgroups and combined with float atomics (summation order not fixed).     */
int dlwp_gemm(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb,
              int ldc, int transA, int transB, const float* bias, int act, float* preact,
              const float* residual, int accumulate, float* rowsum, void* stream);
/* Operand precision of dlwp_gemm / dlwp_gemm_batched, process-wide: 0 (default) = exact fp32 */
/* MFMA; 1 = operands rounded to bf16 in LDS, v_mfma_f32_16x16x32_bf16, fp32 accumulation --   */
/* the arithmetic of the reference's bf16-autocast runs (BASELINE C3-C5).  HBM tensors stay     */
/* fp32 either way.                                                                             */
int dlwp_set_gemm_precision(int mode);
int dlwp_get_gemm_precision(void);
/* 256 x 256 x 64 tiles in two wave groups half a phase apart (bf16 arrays, y = x W^T): -1 never, 0 by shape (K >= 4096 and at    */
/* least 512 tiles: one workgroup per CU leaves its prologue / epilogue uncovered), 1 wherever the kernel applies (measurement).   */
int dlwp_set_gemm_tile256(int mode);
/* Weight gradients of a token MLP's layers in ONE launch: gw_i [N][K] (+)= g_i^T x_i with g_i [T][N] (the gradient of  */
/* the layer's output), x_i [T][K] (its input), gb_i [N] += column sums of g_i (NULL: none) -- what                        */
/* dlwp_gemm_mixed(transA = 1, rowsum = gb) computes per layer (torch.nn.Linear backward in Mlp / SFNO block tails,      */
/* fourcastnet.py:50-56).  Operands fp32 or bf16 arrays (g_bf16 / x_bf16).  Up to three products are grouped;             */
/* otherwise (or when an operand is not 16-byte aligned) they run one launch each.                                      */
typedef struct dlwp_wgrad_desc {
    const void* g;
    const void* x;
    float* gw;
    float* gb;
    int T, N, K;
    int g_bf16, x_bf16, accumulate;
} dlwp_wgrad_desc;
int dlwp_weight_grad_group(const dlwp_wgrad_desc* products, int n, void* stream);
/* The same gradients for a layer applied SEVERAL times with the same weights (the lead times of a rollout,                 */
/* src/dlwpbench/models/fno/fno.py:217-259): gW += sum_s g_s^T x_s and gb += sum_s sum_t g_s[t] as ONE product over the            */
/* concatenated token axis, up to DLWP_WGRAD_MAX_PRODUCTS layers per call (csrc/wgrad_multi.hip).  Every segment is T tokens;       */
/* g_s [T][N], x_s [T][K] bf16 arrays (16-byte aligned, N and K multiples of 8), gw [N][K] and gb [N] fp32, accumulated into.       */
/* The K slices go to the CALLER's workspace (dlwp_wgrad_segments_workspace_bytes) and are added in slice order by a second        */
/* launch: no allocation, no synchronisation, bit-reproducible weight gradients.                                                   */
#define DLWP_WGRAD_MAX_PRODUCTS 6
#define DLWP_WGRAD_MAX_SEGMENTS 8
typedef struct dlwp_wgrad_seg_product {
    const void* g[DLWP_WGRAD_MAX_SEGMENTS];
    const void* x[DLWP_WGRAD_MAX_SEGMENTS];
    float* gw;
    float* gb;                              /* nullable */
    int N, K;
    int overwrite;                          /* != 0: gw = the sum (no read of its old contents); gb is always accumulated into */
} dlwp_wgrad_seg_product;
size_t dlwp_wgrad_segments_workspace_bytes(const dlwp_wgrad_seg_product* products, int nprod, int nseg, int T);
int dlwp_wgrad_segments(const dlwp_wgrad_seg_product* products, int nprod, int nseg, int T, void* workspace,
                        size_t workspace_bytes, void* stream);
/* Any INDEPENDENT small products: between _begin and _end, the dlwp_gemm* entries park products that fit the generic 64 x 64 kernel   */
/* and are small (<= 16384 deep, latency-bound by themselves); _end launches up to three of them as one grid.  Everything else      */
/* launches at once.  Used around the two gradient products of a Linear layer's backward pass (gx = g W and gW = g^T x read the     */
/* same g and do not depend on each other).  One group per HOST THREAD (thread-local queue); not reentrant.  Contract: between       */
/* _begin and _end the caller issues no kernel that depends on a parked product's output, and keeps every operand of a parked        */
/* product alive and unmodified until _end returns (the launches happen there).                                                      */
int dlwp_gemm_group_begin(void);
int dlwp_gemm_group_end(void* stream);

/* Strided-batched form: batch z = z1*nb2 + z2 (z1 < nb1, z2 < nb2) works on A + z1*sA1 +      */
/* z2*sA2, B + z1*sB1 + z2*sB2, C (and residual) likewise; strides in floats, 0 = shared.     */
/* res_before_act != 0 adds the residual before the activation: C = act(A.B + bias + res);    */
/* preact (optional, C's layout and strides) receives the value the activation is applied to.  */
/* act: 0 none, 1 GELU, 2 ReLU, 3 soft-shrink(act_param); bias has its own batch strides.       */
/* act 4 is the backward form C = (A.B) * GELU'(z) with the saved pre-activation z passed as    */
/* `residual` (preact must be NULL): gx = g W of a Linear layer arrives already multiplied by  */
/* the derivative of the GELU that produced its input (token-MLP backward, no gelu_bwd pass).  */
/* act 5 / 6: the same with ReLU' ([z > 0]) and soft-shrink' ([|z| > act_param]): the AFNO block */
/* MLP's activations (fourcastnet.py:100-117, F.relu and F.softshrink).                        */
/* act 7 (also in dlwp_gemm / dlwp_gemm_mixed): GELU whose `preact` output (required) receives  */
/* the DERIVATIVE GELU'(A.B + bias (+ res)) instead of the pre-activation; act 8: the backward   */
/* form C = (A.B) * f with the stored factor f passed as `residual` -- the token MLP's backward  */
/* product then multiplies by what the forward call stored and evaluates no exponential.        */
/* Without an epilogue, long-K products with few output tiles are split along K (atomics).      */
/* Used for the spherical transforms (per-order Legendre matrices) and the per-degree SFNO     */
/* spectral weights, where one launch covers every (sample, order) or degree.                  */
int dlwp_gemm_batched(const float* A, const float* B, float* C, int M, int N, int K, int lda,
                      int ldb, int ldc, int transA, int transB, int nb1, int nb2, long long sA1,
                      long long sA2, long long sB1, long long sB2, long long sC1, long long sC2,
                      const float* bias, long long sBi1, long long sBi2, int act, float act_param,
                      float* preact, const float* residual, long long sR1, long long sR2,
                      int res_before_act, int accumulate, void* stream);
/* bf16 STORAGE for the token GEMMs (BASELINE configs C3-C5 train under bf16 autocast: activations and the   */
/* weights the matrix units read are bf16 in the reference, src/dlwpbench/scripts/train.py:120-135 autocast).   */
/* dlwp_gemm / dlwp_gemm_batched with a storage mask `dtypes`: 1 = A, 2 = B, 4 = C and preact, 8 = residual   */
/* point at bf16 (2-byte) arrays; leading dimensions and batch strides stay in ELEMENTS.  Products accumulate  */
/* in fp32 and the epilogue (bias, activation, residual) runs in fp32 before the output is rounded.  A bf16    */
/* output cannot be accumulated into and is never split along K.  Unset bits are fp32 as before.               */
int dlwp_gemm_mixed(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb,
                    int ldc, int transA, int transB, const float* bias, int act, void* preact,
                    const void* residual, int accumulate, float* rowsum, int dtypes, void* stream);
int dlwp_gemm_batched_mixed(const void* A, const void* B, void* C, int M, int N, int K, int lda,
                            int ldb, int ldc, int transA, int transB, int nb1, int nb2,
                            long long sA1, long long sA2, long long sB1, long long sB2,
                            long long sC1, long long sC2, const float* bias, long long sBi1,
                            long long sBi2, int act, float act_param, void* preact,
                            const void* residual, long long sR1, long long sR2, int res_before_act,
                            int accumulate, int dtypes, void* stream);
/* Stochastic depth inside the last product of a residual branch (timm DropPath as the reference's Swin / Pangu */
/* blocks apply it: x = shortcut + drop_path(branch(x)), nsbench swin_transformer.py:255-256, dlwpbench         */
/* panguweather.py:317-323): C = (op(A) op(B) + bias) * row_scale[m / rows_per_scale] + residual, row_scale one   */
/* fp32 number per sample (0 or 1 / keep), rows_per_scale = tokens per sample.  Storage mask as dlwp_gemm_mixed.  */
/* No activation, pre-activation output, accumulation or row sums in this form.                                   */
int dlwp_gemm_rowscale(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb,
                       int ldc, int transA, int transB, const float* bias, const void* residual,
                       const float* row_scale, int rows_per_scale, int dtypes, void* stream);
/* dst[i] = bf16(src[i]) (round to nearest even): the per-step bf16 copy of the flat fp32 master weights.     */
int dlwp_cast_bf16(const float* src, void* dst, long long n, void* stream);
/* Transposed bf16 copies of a list of fp32 matrices inside two flat buffers, one launch (round 6): matrix m is src_base[descs[4m]      */
/* ...] viewed [rows = descs[4m+2]][cols = descs[4m+3]], its copy dst_base[descs[4m+1] ...] viewed [cols][rows] (bf16).  descs is a    */
/* DEVICE array of n x 4 64-bit integers; max_tiles = the largest ceil(rows/64) * ceil(cols/64).  The per-step [in][out] copies of    */
/* the Linear weights the input-gradient products read (reference: autograd's gx = g W for torch.nn.Linear, token layers of           */
/* swin_transformer.py / panguweather.py / fourcastnet.py).                                                                             */
int dlwp_transpose_cast_bf16_many(const float* src_base, void* dst_base, const long long* descs, int n, int max_tiles, void* stream);
/* dst[s][i] = bf16(src[s][i] * scale[s]), i < per_sample (a multiple of 4): the backward of dlwp_gemm_rowscale's */
/* scale on the way into the branch's bf16 products.                                                              */
int dlwp_cast_bf16_scaled(const float* src, const float* scale, void* dst, int nsamples, long long per_sample,
                          void* stream);
/* Fused real spherical harmonic transforms on channels-last fields (torch_harmonics.RealSHT /  */
/* InverseRealSHT, constructed at src/dlwpbench/models/fno/fno.py:183-200 and                   */
/* models/fourcastnet/fourcastnet.py:411-428): longitude DFT and Legendre   */
/* transform in one launch, the (latitude x order) plane stays in LDS.                          */
/*   analysis : x [B][nlat][nlon][C] -> X [lmax][B][mmax][2][C]                                 */
/*              T[k][q][c] = sum_n A1[q][n] x[b][k][n][c]   (A1 [2 mmax][nlon], q = 2m + re|im)  */
/*              X[l][b][m][ri][c] = sum_k A2[m][l][k] T[k][2m+ri][c]   (A2 [mmax][lmax][nlat])   */
/*   synthesis: X -> x, T[k][2m+ri][c] = sum_l S1[m][l][k] X[l][b][m][ri][c],                   */
/*              x[b][k][n][c] = sum_q S2[q][n] T[k][q][c]; tables given TRANSPOSED:             */
/*              S1t [mmax][nlat][lmax], S2t [nlon][2 mmax].                                     */
/* The adjoint of analysis(A1, A2) is synthesis(S1 = A2, S2 = A1) and vice versa (backward      */
/* passes).  dlwp_sht_fused_supported tells whether a shape fits (C % 16 == 0, mmax % 8 == 0,   */
/* nlon <= 128, nlat <= 64, lmax <= 64, all multiples of 4; tables 16-byte aligned); other      */
/* shapes use dlwp_gemm_batched.                                                                */
int dlwp_sht_fused_supported(int nlat, int nlon, int C, int mmax, int lmax);
int dlwp_sht_analysis(const float* x, const float* A1, const float* A2, float* X, int B, int nlat,
                      int nlon, int C, int mmax, int lmax, void* stream);
int dlwp_sht_synthesis(const float* X, const float* S1t, const float* S2t, float* x, int B, int nlat,
                       int nlon, int C, int mmax, int lmax, void* stream);
/* The same two transforms on the bf16 matrix cores for the bf16-storage chain (csrc/sht_bf16.hip): x fp32, X bf16, tables    */
/* bf16 with the contraction index contiguous: analysis A1 [2 mmax][nlon], A2 [mmax][lmax][nlat]; synthesis S1t                */
/* [mmax][nlat][lmax] (note: the TRANSPOSED Legendre table) and S2 [nlon][2 mmax] (NOT transposed); `residual` (nullable,     */
/* x's layout) is added to the synthesised field (the skip gradient in the backward pass of a forked analysis).               */
/* dlwp_sht_bf16_supported: C % 16 == 0, nlat / nlon / lmax / 2 mmax multiples of 8, nlat <= 64, nlon <= 128, lmax <= 64 and    */
/* the spectrum image within the LDS; other shapes use dlwp_gemm_batched_mixed.                                               */
int dlwp_sht_bf16_supported(int nlat, int nlon, int C, int mmax, int lmax);
int dlwp_sht_analysis_bf16(const float* x, const void* A1, const void* A2, void* X, int B, int nlat, int nlon, int C, int mmax,
                           int lmax, void* stream);
int dlwp_sht_synthesis_bf16(const void* X, const void* S1t, const void* S2, const float* residual, float* x, int B, int nlat,
                            int nlon, int C, int mmax, int lmax, void* stream);
/* The same with flags.  DLWP_SHT_TRIANGULAR: the caller guarantees X[l][.][m] = 0 for every order m > l (a spectrum that RealSHT  */
/* produced, its image under the per-degree weights of dlwp_dhconv_apply, or the gradient of either): those entries are not     */
/* read (48 % of the image at lmax = mmax = 32); with other spectra the flag gives wrong results.                               */
/* DLWP_SHT_FIELD_BF16: the FIELD tensor (the synthesis output x, which then takes no residual; the analysis input x) is a bf16  */
/* array in x's layout instead of fp32 -- for fields whose only other user is a bf16-operand kernel (the output y of an SFNO      */
/* block's spectral filter, read once by dlwp_sfno_tail_fwd; the gradient the tail's backward call leaves in gt_lp).             */
#define DLWP_SHT_TRIANGULAR 1
#define DLWP_SHT_FIELD_BF16 2
int dlwp_sht_synthesis_bf16_ex(const void* X, const void* S1t, const void* S2, const float* residual, void* x, int B, int nlat,
                               int nlon, int C, int mmax, int lmax, int flags, void* stream);
int dlwp_sht_analysis_bf16_ex(const void* x, const void* A1, const void* A2, void* X, int B, int nlat, int nlon, int C, int mmax,
                              int lmax, int flags, void* stream);
/* SFNO "driscoll-healy" spectral weights (torch_harmonics, constructed at                    */
/* src/dlwpbench/models/fno/fno.py:183-200): w [Cin][Cout][L][2] complex, one matrix per       */
/* degree l.  expand: wexp[l] = [[Wr, Wi], [-Wi, Wr]] as a real [2Cin][2Cout] matrix, so that   */
/* the complex contraction "bixy,iox->boxy" is one dlwp_gemm_batched over l on rows            */
/* [Xr | Xi]; fold: gw += the complex gradient read back from the [L][2Cin][2Cout] GEMM result. */
int dlwp_cweight_expand(const float* w, float* wexp, int Cin, int Cout, int L, void* stream);
int dlwp_cweight_fold(const float* gexp, float* gw, int Cin, int Cout, int L, void* stream);
/* The same contraction as dedicated bf16 kernels on spectra X [L][rows][C] whose rows are (sample, order, re | im) -- the        */
/* (re, im) parts of a complex row are two consecutive rows -- without the expanded image (csrc/dhconv.hip):                    */
/*   dlwp_dhconv_pack   w [Cin][Cout][L][2] fp32 -> the forward and the backward fragment-order bf16 images                      */
/*                      (dlwp_dhconv_image_elems(Cin, Cout, L) elements each)                                                   */
/*   dlwp_dhconv_apply  Y[l] = X[l] W[:, :, l] (transposed = 0, forward image, K = Cin, N = Cout) or                            */
/*                      gX[l] = gY[l] conj(W[:, :, l])^T (transposed = 1, backward image, K = Cout, N = Cin); `rows` = B * mmax * 2; */
/*                      mmax > 0: a sample's rows are its orders 0 .. mmax - 1 and orders m > l are KNOWN to be zero (spectra  */
/*                      of the SHT): those row tiles are skipped and written as zeros; mmax = 0: dense rows                     */
/*   dlwp_dhconv_wgrad  G [L][Cin][2 Cout] fp32 = sum over nseg (X, gY) pairs (the lead times of a rollout) of conj(X)^T gY     */
/*                      per degree, real parts in columns [0, Cout), imaginary parts in [Cout, 2 Cout) (overwritten)           */
/*   dlwp_dhconv_fold   gw [Cin][Cout][L][2] += G                                                                               */
/* dlwp_dhconv_supported: Cin, Cout in {128, 256}, L <= 64; other widths use dlwp_cweight_expand + dlwp_gemm_batched_mixed.       */
int dlwp_dhconv_supported(int Cin, int Cout, int L);
long long dlwp_dhconv_image_elems(int Cin, int Cout, int L);
int dlwp_dhconv_pack(const float* w, void* fwd_img, void* bwd_img, int Cin, int Cout, int L, void* stream);
/* the same for up to 16 weights of one shape (the layers of a network) in one launch; host arrays of n device pointers */
int dlwp_dhconv_pack_many(const float* const* w, void* const* fwd_img, void* const* bwd_img, int n, int Cin, int Cout, int L,
                          void* stream);
int dlwp_dhconv_apply(const void* X, const void* image, void* Y, int L, int rows, int K, int N, int mmax, int transposed,
                      void* stream);
int dlwp_dhconv_wgrad(const void* const* X, const void* const* gY, int nseg, float* G, int L, int rows, int Cin, int Cout,
                      int mmax, void* stream);
int dlwp_dhconv_fold(const float* G, float* gw, int Cin, int Cout, int L, void* stream);
/* The tail of an SFNO block as ONE launch per direction (csrc/mlp_chain.hip; the block is      */
/* torch_harmonics' SphericalFourierNeuralOperatorBlock, constructed at                         */
/* src/dlwpbench/models/fno/fno.py:183-200 and models/fourcastnet/fourcastnet.py:411-428,        */
/* SURVEY.md App. A-2): with y = the spectral filter's output and x = the block input,           */
/*     z0 = y + x Ws^T + bs,  t = GELU(z0),  z1 = t W1^T + b1,  h = GELU(z1),                     */
/*     out = h W2^T + b2 (+ x when outer)                                                       */
/* and, given g = d loss / d out,                                                               */
/*     gh = (g W2) * GELU'(z1),  gt = (gh W1) * GELU'(z0) (= d / d y),  gx = gt Ws (+ g).        */
/* Tokens are rows: x, y, out, g, gt, gx [T][C] fp32; z0, t [T][C] and z1, h, gh [T][hidden]    */
/* bf16 arrays; x_lp / g_lp (nullable) and gt_lp receive bf16 copies of x / g / gt for the       */
/* weight-gradient products.  Arithmetic: bf16 operands, fp32 accumulation and epilogues.        */
/* Round 5: the arrays the argument structs call z0 / z1 hold the DERIVATIVES GELU'(z0) and      */
/* GELU'(z1), evaluated by the forward call from the fp32 pre-activations together with the      */
/* activations; the backward call multiplies by them (no exponential / reciprocal per element    */
/* there).  They are opaque to the caller: forward output, backward input.                        */
/* The six weight matrices are read from fragment-order bf16 images built by                    */
/* dlwp_mlp_chain_pack: image of W' [rows][cols] with W' = W (transpose 0, W row-major           */
/* [rows][cols]) or W^T (transpose 1, W row-major [cols][rows]); rows % 16 == 0, cols % 32 == 0, */
/* rows * cols bf16 elements.  Forward images: Ws [C][C], W1 [hidden][C], W2 [C][hidden] as they  */
/* are; backward images: the transposes W2^T [hidden][C], W1^T [C][hidden], Ws^T [C][C].          */
/* dlwp_mlp_chain_supported: (C, hidden) pairs with a compiled kernel (256/512, 128/256,         */
/* 64/128); other widths use three dlwp_gemm_mixed calls per direction.                          */
typedef struct dlwp_sfno_tail_fwd_args {
    const float *x, *y;                     /* [T][C] */
    const void *ws_img, *w1_img, *w2_img;   /* dlwp_mlp_chain_pack images */
    const float *bs, *b1, *b2;              /* [C], [hidden], [C]; nullable */
    void *x_lp;                             /* [T][C] bf16, nullable */
    void *z0, *t, *z1, *h;                  /* bf16 outputs: GELU'(z0) [T][C], t [T][C], GELU'(z1) [T][hidden], h [T][hidden] */
    float *out;                             /* [T][C] */
    int T, C, hidden, outer;
    int y_bf16;                             /* != 0: y is a bf16 array [T][C] (dlwp_sht_synthesis_bf16_ex with DLWP_SHT_FIELD_BF16) */
} dlwp_sfno_tail_fwd_args;
typedef struct dlwp_sfno_tail_bwd_args {
    const float *g;                         /* [T][C] */
    const void *w2t_img, *w1t_img, *wst_img;
    const void *z1, *z0;                    /* the forward call's z1 / z0 arrays (activation derivatives, bf16) */
    void *g_lp;                             /* [T][C] bf16, nullable */
    void *gh;                               /* [T][hidden] bf16 */
    float *gt;                              /* [T][C]; nullable: only the bf16 copy gt_lp is written */
    void *gt_lp;                            /* [T][C] bf16 */
    float *gx;                              /* [T][C] */
    int T, C, hidden, outer;
} dlwp_sfno_tail_bwd_args;
int dlwp_mlp_chain_supported(int C, int hidden);
int dlwp_mlp_chain_pack(const float* W, int rows, int cols, int transpose, void* image, void* stream);
/* all six images of a tail in one launch: images [6][C * hidden] bf16 = forward Ws, W1, W2, backward W2^T, W1^T, Ws^T  */
/* (the two C x C images use the front of their slots); ws [C][C], w1 [hidden][C], w2 [C][hidden] fp32.               */
int dlwp_sfno_tail_pack(const float* ws, const float* w1, const float* w2, int C, int hidden, void* images, void* stream);
/* the same for up to 8 block tails of equal widths in ONE launch (images[b]: the 6-image slot of block b) */
int dlwp_sfno_tail_pack_many(const float* const* ws, const float* const* w1, const float* const* w2, int n, int C, int hidden,
                             void* const* images, void* stream);
int dlwp_sfno_tail_fwd(const dlwp_sfno_tail_fwd_args* args, void* stream);
int dlwp_sfno_tail_bwd(const dlwp_sfno_tail_bwd_args* args, void* stream);
/* Token MLP y = fc2(GELU(fc1 x)) (+ residual) of the AFNO / Swin / Pangu blocks (nsbench/models/fourcastnet/fourcastnet.py:40-56, */
/* swintransformer/swin_transformer.py:25-48) and its input-gradient chain as ONE launch per direction for wide hidden layers      */
/* (csrc/mlp_stream.hip): a workgroup owns 64 tokens and walks the hidden layer in chunks of 128 with the output accumulators in    */
/* registers; z, h ([T][hidden] bf16) are written for the other pass and the weight gradients but never read back as operands.     */
/*   pack: images [4][E * hidden] bf16 = forward W1 [hidden][E], W2 [E][hidden], backward W2^T, W1^T (dlwp_mlp_chain_pack order)   */
/*   fwd : x [T][E] (bf16 array when x_bf16, else fp32 with an optional bf16 copy x_lp) -> z, h, y [T][E] fp32 (+ residual)        */
/*   bwd : g [T][E] fp32 (-> bf16 copy g_lp, nullable), z -> gh [T][hidden] bf16, gx [T][E] (bf16 array when gx_bf16)              */
/* dlwp_mlp_stream_supported: E = 768, hidden a multiple of 128; other widths use two dlwp_gemm_mixed calls per direction.          */
int dlwp_mlp_stream_supported(int E, int hidden);
int dlwp_mlp_stream_pack(const float* w1, const float* w2, int E, int hidden, void* images, void* stream);
int dlwp_mlp_stream_fwd(const void* x, int x_bf16, void* x_lp, const void* w1_img, const float* b1, const void* w2_img, const float* b2,
                        const float* residual, void* z, void* h, float* y, int T, int E, int hidden, void* stream);
int dlwp_mlp_stream_bwd(const float* g, void* g_lp, const void* w2t_img, const void* w1t_img, const void* z, void* gh, void* gx,
                        int gx_bf16, int T, int E, int hidden, void* stream);
/* SFNO encoder / decoder with the rollout's frame assembly, ONE launch per direction each (csrc/sfno_io.hip).  Reference:   */
/* SphericalFourierNeuralOperatorNet's encoder / decoder 1x1-convolution MLPs, position embedding and big skip (constructed at */
/* src/dlwpbench/models/fno/fno.py:183-200) inside SFNO2DModule.forward's loop (fno.py:217-259; clean form unet.py:64-111):    */
/*   x_t = cat(constants[:, 0], prescribed[:, t-1], frame);  t0 = W2e GELU(W1e x_t + b1e) + pos;                               */
/*   y = W2d GELU(Wd [t ; x_t] + bd);  out = frame + y.                                                                        */
/* Token tensors are rows (tokens [T][E] fp32, T = B * HW); frames and gradients of frames are NCHW planes given by the        */
/* pointer of sample 0 and a batch stride in floats; tok_lp [T][DLWP_SFNO_IO_KP] holds the gathered input channels of a token  */
/* as bf16 (zero beyond the real channels).  z, h ([T][E] bf16): the hidden layer's pre-activation and activation (forward:   */
/* written; backward: z read, h receives the hidden gradient gh).  One argument struct serves the four calls:                 */
/*   encode_fwd  src* (up to three plane groups) -> tok_lp, z, h, tokens = t0 (+ pos [HW][E], nullable)                        */
/*   encode_bwd  tokens = g (d loss / d t0) -> tokens_lp (bf16 copy of g), h = gh; the token gradient gh W1e (+ tok_grad, the   */
/*               decoder's share, nullable) has its channels frame_c0 .. frame_c0 + frame_c - 1 written to `frame`             */
/*               (+ frame_add, nullable: the gradient that reached the same frame through the decoder's residual)             */
/*   decode_fwd  tokens = t, tok_lp -> tokens_lp (bf16 copy of t), z, h, frame[b][c] = frame_add[b][c] + y (c < frame_c)       */
/*   decode_bwd  src[0] = g_out (src_c[0] planes) -> tok_lp (bf16 rows of g_out), h = gh, tokens = g_t, tok_grad = g_tok        */
/* Weight images: dlwp_sfno_io_pack builds the eight zero-padded fragment-order images (slot i at images + i *                */
/* dlwp_sfno_io_image_elems(E) bf16 elements: encode fwd 1 / 2, encode bwd 1 / 2, decode fwd 1 / 2, decode bwd 1 / 2) from     */
/* enc_w1 [E][in], enc_w2 [E][E], dec_w1 [E][E (+ in with big_skip)], dec_w2 [out][E].  bf16 operands, fp32 accumulation.     */
#define DLWP_SFNO_IO_KP 32
typedef struct dlwp_sfno_io_args {
    const float* src[3];
    long long src_bs[3];
    int src_c[3];
    int HW, T, E;
    float* tokens;
    void* tokens_lp;
    void* tok_lp;
    const void *w1_img, *w2_img;
    const float *bias, *pos;
    void *z, *h;
    float* frame;
    long long frame_bs;
    int frame_c, frame_c0;
    const float* frame_add;
    long long frame_add_bs;
    float* tok_grad;
} dlwp_sfno_io_args;
int dlwp_sfno_io_supported(int E, int in_chans, int out_chans);
long long dlwp_sfno_io_image_elems(int E);
int dlwp_sfno_io_pack(const float* enc_w1, const float* enc_w2, const float* dec_w1, const float* dec_w2, int E, int in_chans,
                      int out_chans, int big_skip, void* images, void* stream);
int dlwp_sfno_encode_fwd(const dlwp_sfno_io_args* args, void* stream);
int dlwp_sfno_encode_bwd(const dlwp_sfno_io_args* args, void* stream);
int dlwp_sfno_decode_fwd(const dlwp_sfno_io_args* args, void* stream);
int dlwp_sfno_decode_bwd(const dlwp_sfno_io_args* args, void* stream);
/* LayerNorm over the last dimension of x [T,C]; mean/rstd [T] are saved for backward.       */
int dlwp_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                       float* mean, float* rstd, int T, int C, float eps, void* stream);
/* The same with the output stored as bf16 (y_bf16 != 0: y is a 2-byte array) for LayerNorms whose only   */
/* reader is a GEMM under bf16 storage (dlwp_gemm_mixed); statistics and the backward pass stay fp32.       */
int dlwp_layernorm_fwd_ex(const float* x, const float* gamma, const float* beta, void* y, float* mean,
                          float* rstd, int T, int C, float eps, int y_bf16, void* stream);
/* Backward of a LayerNorm whose bf16 output fed a GEMM: the upstream gradient arrives as a bf16 array      */
/* (gy_bf16 != 0) straight from that GEMM's input-gradient product; gadd as in dlwp_layernorm_bwd_res.      */
int dlwp_layernorm_bwd_ex(const float* x, const float* gamma, const float* mean, const float* rstd,
                          const void* gy, int gy_bf16, const float* gadd, float* gx, float* ggamma,
                          float* gbeta, int T, int C, void* stream);

/* As dlwp_layernorm_bwd_ex, with a second output (round 6): gx_bf16[t][c] = bf16(gx[t][c] * row_scale[t / rows_per_sample]) -- the      */
/* gradient the residual branch that ENDS in this stream wants for its backward products (per-sample stochastic-depth scale, bf16       */
/* operands; reference: DropPath in Block.forward, swin_transformer.py:255-256, whose backward scales the branch gradient), written    */
/* from the registers that hold gx instead of by a separate pass over it.  row_scale NULL: a plain bf16 copy.                           */
int dlwp_layernorm_bwd_lowp(const float* x, const float* gamma, const float* mean, const float* rstd, const void* gy, int gy_bf16,
                            const float* gadd, float* gx, float* ggamma, float* gbeta, int T, int C, void* gx_bf16,
                            const float* row_scale, int rows_per_sample, void* stream);
/* gx written; ggamma/gbeta ACCUMULATED into.  C <= 2048.                                    */
int dlwp_layernorm_bwd(const float* x, const float* gamma, const float* mean, const float* rstd,
                       const float* gy, float* gx, float* ggamma, float* gbeta, int T, int C,
                       void* stream);
/* The same with gx = (LayerNorm backward) + gadd [T,C] (nullable): in the pre-norm residual   */
/* blocks (`x + f(norm(x))`: nsbench fourcastnet.py:156-165, swin_transformer.py:229-256;       */
/* dlwpbench panguweather.py:143-159) the gradient of the block input is the sum of the residual */
/* branch's and the norm's -- formed here instead of by a separate elementwise add.              */
int dlwp_layernorm_bwd_res(const float* x, const float* gamma, const float* mean, const float* rstd,
                           const float* gy, const float* gadd, float* gx, float* ggamma,
                           float* gbeta, int T, int C, void* stream);
/* Instance normalisation of channels-last tokens x [B][P][C]: per (sample, channel) mean / biased       */
/* variance over the P = H*W tokens, y = (x - mean) rstd gamma + beta (+ residual, same layout, optional).  */
/* Replaces nn.InstanceNorm2d(embed_dim, eps, affine=True) inside torch_harmonics' SFNO                      */
/* (normalization_layer="instance_norm": src/dlwpbench/configs/model/fourcastnetv2.yaml:23, constructed at   */
/* src/dlwpbench/models/fourcastnet/fourcastnet.py:411-428).  stats [B][C][2] receives (mean, rstd).         */
int dlwp_instnorm_fwd(const float* x, const float* gamma, const float* beta, const float* residual,
                      float* y, float* stats, int B, int P, int C, float eps, void* stream);
/* gx written; ggamma / gbeta ACCUMULATED into; work: scratch [B][C][2].                                    */
int dlwp_instnorm_bwd(const float* x, const float* gamma, const float* stats, const float* gy, float* gx,
                      float* ggamma, float* gbeta, float* work, int B, int P, int C, void* stream);
/* gz = gy * gelu'(z) (exact erf GELU)                                                       */
int dlwp_gelu_bwd(const float* z, const float* gy, float* gz, long long n, void* stream);
/* gz = gy * act'(z) for the epilogue activations of dlwp_gemm_batched (1 GELU, 2 ReLU, 3 soft-shrink) */
int dlwp_act_bwd(const float* z, const float* gy, float* gz, long long n, int act, float act_param,
                 void* stream);
/* Stochastic depth (timm DropPath: src/nsbench/models/swintransformer/swin_transformer.py:193,   */
/* 255-256; dlwpbench twin :192,261-262; panguweather.py:262-323): the per-sample keep mask, already    */
/* divided by the keep probability, scales a residual branch: out[b][i] = x[b][i] + scale[b] t[b][i]    */
/* (x may be NULL: plain scaling, which is also the backward of the branch: gt = scale[b] g).           */
/* t, x, out: [B][n]; scale: [B] device floats, or NULL for unit scales (a plain fused add).              */
int dlwp_scale_rows_add(const float* t, const float* scale, const float* x, float* out, int B,
                        long long n, void* stream);
/* out[b][i] = t[b][i] + p[i]: a learned position embedding [n] added to every sample's tokens   */
/* (`x = patch_embed(x) + pos_embed`, src/nsbench/models/fourcastnet/fourcastnet.py:253; dlwpbench */
/* twin).  Its parameter gradient is dlwp_colsum over the batch.                                   */
int dlwp_add_bcast(const float* t, const float* p, float* out, int B, long long n, void* stream);
/* out[n] += sum_t g[t][n]   (bias gradients)                                                */
int dlwp_colsum(const float* g, float* out, int T, int N, void* stream);
/* dst[r][c] += src[r * src_rs + c * src_cs] (dst row pitch dst_ld) for up to DLWP_ADD2D_MAX small matrices in ONE launch: */
/* padded, concatenated or transposed temporary gradients handed to the parameters' gradient buffers (the encoder's  */
/* / decoder's 1x1 convolution weights and the position embedding of SFNO2DModule, dlwpbench fno.py:217-259).        */
#define DLWP_ADD2D_MAX 8
typedef struct dlwp_add2d_desc {
    float* dst;
    const float* src;
    long long dst_ld, src_rs, src_cs;
    int rows, cols;
} dlwp_add2d_desc;
int dlwp_add2d_many(const dlwp_add2d_desc* descs, int n, void* stream);
/* out[n] += sum_t g[t][n] for a bf16 array g (tall form, T > 16)                                                      */
int dlwp_colsum_bf16(const void* g, float* out, int T, int N, void* stream);
/* the same with overwrite != 0: out[n] = sum_t g[t][n] (the first of several accumulating calls needs no zero fill) */
int dlwp_colsum_ex(const float* g, float* out, int T, int N, int overwrite, void* stream);

/* ------------------------------------------------------------------------------------ */
/* Complex mode-n product for Tucker-factorised spectral weights (TFNO, dlwpbench/models/   */
/* fno/fno.py:136-146): out[O,N,I] = sum_r in[O,r,I] * U[N,r]; complex = interleaved re/im. */
/* The factorised weight core x1 U_i x2 U_o x3 U_x x4 U_y is expanded into the dense layout   */
/* of dlwp_fno_block_* once per optimizer step.  _bwd writes gin and gU.                      */
int dlwp_cmode_product(const float* in, const float* U, float* out, int O, int R, int N, int I,
                       void* stream);
int dlwp_cmode_product_bwd(const float* in, const float* U, const float* gout, float* gin,
                           float* gU, int O, int R, int N, int I, void* stream);

/* ------------------------------------------------------------------------------------ */
/* Data-parallel exchange over RCCL (xGMI).  The reference trains single-process            */
/* (nsbench/scripts/train.py:36,66); sharding trajectory samples over the GPUs of a node is a  */
/* build addition whose only collective is ONE sum all-reduce of the flat fp32  */
/* gradient buffer per optimizer step (then Adam with grad_scale = 1 / world).  One communicator */
/* per process / GPU: rank 0 calls dlwp_comm_unique_id and hands the 128 bytes to every rank     */
/* (any side channel), every rank calls dlwp_comm_create (blocks until all have joined).         */
/* Collectives are enqueued on the caller's stream.  librccl is loaded at the first call.        */
typedef struct dlwp_comm dlwp_comm;
int dlwp_comm_unique_id(void* out128);
int dlwp_comm_create(const void* unique_id128, int rank, int world, dlwp_comm** out);
void dlwp_comm_destroy(dlwp_comm* comm);
/* buf[0:n] <- sum over ranks (in place)                                                   */
int dlwp_comm_allreduce(dlwp_comm* comm, float* buf, long long n, void* stream);
/* buf[0:n] <- root's buf (initial parameters, Adam state on resume)                       */
int dlwp_comm_broadcast(dlwp_comm* comm, float* buf, long long n, int root, void* stream);

/* bench probe: ONE forward `spatial` launch of an inner FNO block as the rollout issues it     */
/* (x = previous pre-activation, GELU on load; spec = [B][m1][m2c][C][2] mixed modes; fused      */
/* W-axis DFT of gelu(pre) into x1_out [B][H][m2c][C][2]).                                       */
int dlwp_fno_spatial_fwd_probe(const dlwp_fno_plan* plan, const float* x, const float* spec,
                               const float* wskip, const float* bias, float* pre, float* x1_out,
                               int B, void* stream);
/* bench probe: ONE forward per-mode launch (H-axis step of x1 [B][H][m2c][C][2] + the mode-truncated complex channel      */
/* contraction with wspec on the matrix cores) -> xhat, y [B][m1][m2c][C][2].                                               */
int dlwp_fno_mix_fwd_probe(const dlwp_fno_plan* plan, const float* x1, const float* wspec, float* xhat,
                           float* y, int B, void* stream);
/* debug: enqueue n dependent empty kernels of `blocks` workgroups (per-kernel floor probe)   */
int dlwp_debug_null_kernels(int n, int blocks, void* stream);
/* same with a body: every wave spins for `cycles` shader cycles; grid, block size and dynamic LDS chosen by the caller */
int dlwp_debug_spin_kernels(int n, int blocks, int threads, int cycles, int lds_bytes, void* stream);
/* debug: out[0] = shader-clock ticks, out[1] = 100 MHz ticks spent in a dependent-FMA loop   */
int dlwp_debug_clock_probe(unsigned long long* out, int iters, int blocks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DLWPMI_H */

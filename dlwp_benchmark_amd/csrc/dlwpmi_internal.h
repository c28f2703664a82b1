// Internal (non-exported) interfaces shared between the translation units of libdlwpmi.
#pragma once
#include <climits>
#include <hip/hip_runtime.h>
#include "../../include/dlwpmi.h"

// tuning.hip -- the registry of measurement knobs: the override of knob `name` (dlwp_set_tuning, else the environment variable
// DLWP_<name>, read at the time of the call) or DLWP_TUNE_UNSET = "the library's own choice"
constexpr int DLWP_TUNE_UNSET = INT_MIN;
int dlwp_tune(const char* name);
inline bool dlwp_tune_on(const char* name) { const int v = dlwp_tune(name); return v != DLWP_TUNE_UNSET && v != 0; }
inline int dlwp_tune_or(const char* name, int dflt) { const int v = dlwp_tune(name); return v == DLWP_TUNE_UNSET ? dflt : v; }

// prof.hip -- live per-kernel accounting (dlwp_prof_enable / _collect / _get): a scope object around a launch records an event pair
// on the launch stream plus the kernel's name and its algorithmic flops / HBM bytes; free when disabled or under stream capture
bool dlwp_prof_on();
bool dlwp_prof_detail();     // dlwp_prof_enable(2): row names carry shapes (one row per distinct product / launch geometry)
struct dlwp_prof_scope {
    int idx;
    hipStream_t stream;
    dlwp_prof_scope(hipStream_t s, double flops, double bytes, const char* fmt, ...) __attribute__((format(printf, 5, 6)));
    ~dlwp_prof_scope();
    dlwp_prof_scope(const dlwp_prof_scope&) = delete;
    dlwp_prof_scope& operator=(const dlwp_prof_scope&) = delete;
};

// pwmlp.hip — strided/gathered channel views, optional residual and fused MSE gradient
int dlwp_pwmlp_fwd_ex(const dlwp_chan_src* x, const float* w1, const float* b1, const float* w2,
                      const float* b2, const dlwp_chan_dst* y, const dlwp_chan_src* res, int B,
                      int Cin, int Ch, int Cout, int P, hipStream_t stream);
// as dlwp_pwmlp_fwd_ex, plus (x1_out != nullptr) the W-axis pruned DFT of every finished output row written to x1_out
// in dlwp_fno_rows_dft's layout -- the lifting MLP then feeds the first spectral block without a rows launch
struct dlwp_fno_plan;
bool dlwp_pwmlp_rows_fusable(const dlwp_fno_plan* plan, int Cout, int P);
int dlwp_pwmlp_fwd_rows_ex(const dlwp_chan_src* x, const float* w1, const float* b1, const float* w2,
                           const float* b2, const dlwp_chan_dst* y, const dlwp_chan_src* res, int B,
                           int Cin, int Ch, int Cout, int P, const dlwp_fno_plan* rows_plan, float2* x1_out,
                           hipStream_t stream);
// projection MLP of one net call chained with the lifting MLP (+ rows DFT) of the next in one launch; next_ch[o] = input channel
// of the second MLP fed by output channel o of the first (-1: none).  DLWP_E_UNSUPPORTED (error string untouched) when the
// shapes do not allow it: the caller issues the two launches instead.
int dlwp_pwmlp_fwd_chain_ex(const dlwp_chan_src* x1v, const float* w11, const float* b11, const float* w12, const float* b12,
                            const dlwp_chan_dst* y1, const dlwp_chan_src* res1, int Cin1, int Ch1, int Cout1,
                            const dlwp_chan_src* x2v, const float* w21, const float* b21, const float* w22, const float* b22,
                            const dlwp_chan_dst* y2, int Cin2, int Ch2, int Cout2, const signed char* next_ch, int B, int P,
                            const dlwp_fno_plan* rows_plan, float2* x1_out, hipStream_t stream);
int dlwp_pwmlp_bwd_ex(const dlwp_chan_src* x, const float* w1, const float* b1, const float* w2,
                      const dlwp_chan_src* gy, const dlwp_chan_src* pred, const dlwp_chan_src* target,
                      float mse_scale, const dlwp_chan_dst* gx, int gx_accumulate, const dlwp_chan_dst* gres,
                      float* gw1, float* gb1, float* gw2, float* gb2, float* slab, int slab_accumulate, int B,
                      int Cin, int Ch, int Cout, int P, hipStream_t stream);
// as dlwp_pwmlp_bwd_ex, plus (x1_out != nullptr) the adjoint W-axis DFT of every finished gx row written to x1_out
// (dlwp_fno_rows_dft(plan, gx, 0, 1, ...)'s result): the projection backward feeds the last block's backward directly
int dlwp_pwmlp_bwd_rows_ex(const dlwp_chan_src* x, const float* w1, const float* b1, const float* w2,
                           const dlwp_chan_src* gy, const dlwp_chan_src* pred, const dlwp_chan_src* target,
                           float mse_scale, const dlwp_chan_dst* gx, int gx_accumulate, const dlwp_chan_dst* gres,
                           float* gw1, float* gb1, float* gw2, float* gb2, float* slab, int slab_accumulate, int B,
                           int Cin, int Ch, int Cout, int P, const dlwp_fno_plan* rows_plan, float2* x1_out,
                           hipStream_t stream);
// per-workgroup parameter-gradient slabs of pwmlp_bwd: [slab_count][slab_stride] floats
long long dlwp_pwmlp_slab_stride(int Cin, int Ch, int Cout);
int dlwp_pwmlp_slab_count(int B, int P);
int dlwp_pwmlp_slab_reduce(const float* slab, int nslab, int Cin, int Ch, int Cout, float* gw1, float* gb1,
                           float* gw2, float* gb2, hipStream_t stream);

// fno_block.hip — the three stages of a block, exposed so the trainer can interleave them
struct dlwp_fno_plan {
    int C, H, W, m1, m2c;
    int C_pad, NP;            // channels padded to 16; 2*m2c padded to 16
    int force_wide;           // use the channel-blocked kernels whatever the width (3-D plans: hundreds of row frequencies)
    // device tables (built in double on the host)
    float2* twH;              // [m1][H]    e^{-2 pi i k_j h / H}, k_j = signed kept row frequency
    float* FT_fwd;            // [NP][W]    forward W-axis DFT (scale 1/(H W)), n = 2 kx (+1: imag)
    float* FT_adj;            // [NP][W]    adjoint of the inverse W-axis step (scale c_kx)
    float* G_inv;             // [NP][W]    inverse W-axis step (c_kx cos, -c_kx sin)
    float* G_adj;             // [NP][W]    adjoint of the forward W-axis step (scale 1/(H W))
};

// rows: x1[b][h][kx][c] (complex) = sum_w act(x[b][c][h][w]) * FT[2kx(+1)][w]
int dlwp_fno_rows_dft(const dlwp_fno_plan* p, const float* x, int act_in, int adjoint, float2* x1,
                      int B, hipStream_t stream);
// per-mode stage, forward: xhat[b][j][kx][i] = sum_h x1[b][h][kx][i] twH[j][h];
//                          y[b][j][kx][o] = sum_i xhat * wspec[j][kx][i][o]
int dlwp_fno_mix_fwd(const dlwp_fno_plan* p, const float2* x1, const float2* wspec, float2* xhat,
                     float2* y, int B, hipStream_t stream);
// backward: ghat = H-step of g1; gx[b][j][kx][i] = sum_o ghat conj(w); gw += conj(xhat) ghat
int dlwp_fno_mix_bwd(const dlwp_fno_plan* p, const float2* g1, const float2* wspec,
                     const float2* xhat, float2* gxhat, float2* g_wspec, int B, hipStream_t stream);
// spatial stage (one workgroup per image row, all channels).
struct dlwp_fno_spatial_args {
    const float* tin;        // [B,C,H,W] GEMM input (fwd: x, bwd: g_pre)
    int act_tin;             // apply GELU to tin on load (fwd only)
    const float2* spec;      // [B][m1][m2c][C] complex: fwd y, bwd gxhat
    const float* wskip;      // [C][C] (fwd uses W, bwd uses W^T)
    int transpose_w;
    const float* bias;       // fwd only (nullable)
    const float* pprev;      // bwd: the block input x (pre-activation if act_prev)
    int act_prev;            // bwd: x was passed through GELU => multiply by gelu'(pprev)
    float* out;              // [B,C,H,W]: fwd pre, bwd g_x
    float2* x1_out;          // optional fused row DFT of the result (fwd: of gelu(out) when
    int x1_act;              //   x1_act; bwd: adjoint table), layout [b][h][kx][c]
    int x1_adjoint;
    float* g_wskip;          // bwd: += g_pre . act(x)^T   (nullable)
    float* g_bias;           // bwd: += sum g_pre          (nullable)
    float* gslab;            // bwd: per-workgroup {g_wskip,g_bias} partial slab [B*H][dlwp_fno_gslab_stride]
    int gslab_accumulate;    //      instead of same-address float atomics (0: overwrite, 1: read-modify-write)
    int inverse_adjoint;     // which G table: 0 = inverse step (fwd), 1 = adjoint of fwd step
    int B;
};
int dlwp_fno_spatial(const dlwp_fno_plan* p, const dlwp_fno_spatial_args* a, hipStream_t stream);
long long dlwp_fno_gslab_stride(int C);
// every partial slab of a training step folded by ONE launch (slab-parallel, coalesced):
//   plain job  (pwmlp == 0): d1[i] += sum_s slab[s*stride + i] (i < n1), d2[i] += sum_s slab[s*stride + n1 + i] (i < n2)
//   pwmlp job  (pwmlp != 0): the accumulator-tile slabs of dlwp_pwmlp_bwd_ex -> d1 = gw1, d2 = gb1, d3 = gw2, d4 = gb2
struct dlwp_fold_job {
    const float* slab;
    int nslab, pwmlp;
    long long stride, n1, n2;      // plain jobs
    int Cin, Ch, Cout;             // pwmlp jobs
    float *d1, *d2, *d3, *d4;
};
int dlwp_fold_slabs(const dlwp_fold_job* jobs, int njobs, hipStream_t stream);
// train_ops.hip -- start of a fused rollout step in one launch: *loss = 0, g_out[0:n] = 0, out[b][0:row] = x[b][0:row]
int dlwp_rollout_prep(float* loss, float* g_out, long long n, float* out, const float* x, long long out_bs, long long x_bs,
                      long long row, int B, hipStream_t stream);

// token_ops.hip -- the MFMA GEMM with the channels-first extensions of the wide FNO path (fno_wide.hip).
// C_z[M][N] (+)= epilogue(op(A_z)[M][K] . op(B_z)[K][N]), z < nb, operand z at base + z * s? (0 = shared).  Batches that share
// the output (sC == 0, nb > 1) and long-K products are combined with float atomics (C zeroed first unless accumulate).
struct dlwp_gemm_args {
    const float *A, *B;
    float* C;
    int M, N, K, lda, ldb, ldc, transA, transB;
    int nb;
    long long sA, sB, sC, sR;
    const float* bias;       // [N], or [M] when bias_row
    int bias_row;
    int act;                 // 0 none, 1 GELU, 4: C = (A.B) * GELU'(residual)
    float* preact;           // optional store of the value the activation is applied to (C's layout)
    const float* residual;   // C's layout, stride sR; added before (res_before_act) or after the activation
    int res_before_act, accumulate;
    int act_b;               // GELU applied to the B operand on load
    float* rowsum;           // optional [M]: += sum_k op(A)[m][k] over every batch (bias gradients)
};
int dlwp_gemm_run(const dlwp_gemm_args& g, hipStream_t stream);

// fno_wide.hip -- hidden_channels > 64 (the fused kernels keep all channels of a row in LDS and stop at 64)
inline bool dlwp_fno_is_wide(const dlwp_fno_plan* p) { return p->C_pad > 64 || p->force_wide; }
int dlwp_fno_rows_dft_wide(const dlwp_fno_plan* p, const float* x, int act_in, int adjoint, float2* x1, int B,
                           hipStream_t stream);
int dlwp_fno_spatial_wide(const dlwp_fno_plan* p, const dlwp_fno_spatial_args* a, hipStream_t stream);
int dlwp_fno_skip_wgrad(const dlwp_fno_plan* p, const float* g_pre, const float* x, int act_x, float* g_wskip, float* g_bias,
                        int B, hipStream_t stream);
int dlwp_gather_channels(const float* const* src_tab, const long long* bstride_tab, float* dense, int B, int C, long long P,
                         hipStream_t stream);
int dlwp_scatter_add_channels(float* const* dst_tab, const long long* bstride_tab, const float* dense, int B, int C, long long P,
                              hipStream_t stream);
int dlwp_proj_gy(const float* g_out, const float* pred, const float* target, float mse_scale, float* gy, float* gres,
                 long long bs_out, long long bs_res, long long CP, int B, hipStream_t stream);
int dlwp_cfmlp_fwd(const float* x, long long x_bs, const float* w1, const float* b1, const float* w2, const float* b2, float* y,
                   long long y_bs, const float* res, long long res_bs, float* zpre, float* act, int B, int Cin, int Ch, int Cout,
                   int P, hipStream_t stream, long long x_cs = 0);      // x_cs: channel stride of x (0 = P)
int dlwp_cfmlp_bwd(const float* x, long long x_bs, const float* w1, const float* w2, const float* gy, long long gy_bs,
                   const float* zpre, const float* act, float* gx, long long gx_bs, float* gz, float* gw1, float* gb1, float* gw2,
                   float* gb2, int B, int Cin, int Ch, int Cout, int P, hipStream_t stream, long long x_cs = 0, long long gx_cs = 0);

// csrc/winattn_small.hip: the wave-per-(window, head) attention kernels for many short windows (<= 128 tokens, head_dim <= 32);
// dlwp_window_attn_fwd / bwd route to them when dlwp_winattn_small_applies().
bool dlwp_winattn_small_applies(int N, int d, long long pairs);
int dlwp_winattn_small_fwd(const float* qkv, const float* table, const float* packed, const int* ia, const int* ib,
                           const int* labels, float* out, float* lse, int B_, int nW, int N, int TB, int ntypes, int heads, int d,
                           float scale, int q_lo, int q_hi, void* stream, int io_bf16 = 0);
// io_bf16: qkv, out (and gout, gqkv) are bf16 arrays in the window layout (dlwp_winattn_io_bf16_applies must hold)
bool dlwp_winattn_io_bf16_applies(int N, int d, int TB, long long pairs);
int dlwp_winattn_small_bwd(const float* qkv, const float* table, const float* packed, const int* ia, const int* ib,
                           const int* labels, const float* out, const float* lse, const float* gout, float* gqkv, float* gtable,
                           int B_, int nW, int N, int TB, int ntypes, int heads, int d, float scale, int q_lo, int q_hi, void* stream,
                           int io_bf16 = 0);

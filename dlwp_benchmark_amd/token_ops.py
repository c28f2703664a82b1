"""nn.Linear / nn.LayerNorm / token-MLP counterparts whose arithmetic runs in libdlwpmi
(dlwp_gemm, dlwp_layernorm_*, dlwp_gelu_bwd, dlwp_colsum).  They subclass the torch modules only to
keep parameter names (`weight`, `bias`) and initialisation identical to the reference's layers
(nsbench/models/fourcastnet/fourcastnet.py:40-56,213; swintransformer/swin_transformer.py:25-40,131,153).
There is no torch fallback: CPU tensors are refused by lib.ptr().
"""
import ctypes as C

import torch
import torch.nn as nn

from . import lib as L


_BF = torch.bfloat16


def _dtypes(A, B, C, preact, residual):
    """Storage mask of dlwp_gemm_mixed: 1 A, 2 B, 4 C (+ preact), 8 residual are bf16 arrays."""
    if preact is not None and preact.dtype != C.dtype:
        raise L.DlwpError("gemm: preact must have the output's storage type")
    return (1 if A.dtype == _BF else 0) | (2 if B.dtype == _BF else 0) | (4 if C.dtype == _BF else 0) | \
        (8 if residual is not None and residual.dtype == _BF else 0)


def _gemm(A, B, C, M, N, K, lda, ldb, ldc, tA, tB, bias=None, act=0, preact=None, residual=None, accumulate=0,
          rowsum=None):
    dt = _dtypes(A, B, C, preact, residual)
    if dt:
        L.check(L.load().dlwp_gemm_mixed(L.ptr(A), L.ptr(B), L.ptr(C), M, N, K, lda, ldb, ldc, tA, tB, L.ptr(bias), act,
                                         L.ptr(preact), L.ptr(residual), accumulate, L.ptr(rowsum), dt, L.stream()))
        return
    L.check(L.load().dlwp_gemm(L.ptr(A), L.ptr(B), L.ptr(C), M, N, K, lda, ldb, ldc, tA, tB, L.ptr(bias), act,
                               L.ptr(preact), L.ptr(residual), accumulate, L.ptr(rowsum), L.stream()))


def _gemm_rowscale(A, B, C, M, N, K, lda, ldb, ldc, tA, tB, bias, residual, row_scale):
    """C = (op(A) op(B) + bias) * row_scale[sample of the row] + residual (dlwp_gemm_rowscale): M rows = row_scale.numel() samples
    of equally many tokens."""
    nb = row_scale.numel()
    if nb == 0 or M % nb or row_scale.dtype != torch.float32 or not row_scale.is_contiguous():
        raise L.DlwpError("gemm_rowscale: one contiguous fp32 scale per sample, the rows a whole multiple of the samples")
    L.check(L.load().dlwp_gemm_rowscale(L.ptr(A), L.ptr(B), L.ptr(C), M, N, K, lda, ldb, ldc, tA, tB, L.ptr(bias), L.ptr(residual),
                                        L.ptr(row_scale), M // nb, _dtypes(A, B, C, None, residual), L.stream()))


# The scaled bf16 gradient a LayerNorm backward has already written beside its fp32 gx (dlwp_layernorm_bwd_lowp) for the branch that ends
# in its input: keyed by gx's address, valid inside one backward pass (cleared by an engine callback when the pass ends).
# env DLWP_LN_BWD_LOWP=0: A/B runs against the separate cast launch
LN_BWD_LOWP = __import__("os").environ.get("DLWP_LN_BWD_LOWP", "1") != "0"
_GRAD_LOWP = {"map": {}, "armed": False, "hits": 0, "misses": 0}


def _grad_lowp_clear():
    _GRAD_LOWP["map"].clear()
    _GRAD_LOWP["armed"] = False


def _grad_lowp_put(gx, lowp, scale):
    # the entry holds gx (and the scale) itself: while it is alive no other tensor can live at gx's address, so a hit by address IS gx
    # (or a view of all of it) -- an unconsumed entry whose gx had been freed could otherwise match a later tensor at the recycled address
    _GRAD_LOWP["map"][gx.data_ptr()] = (lowp, 0 if scale is None else scale.data_ptr(), gx.numel(), gx, scale)
    if not _GRAD_LOWP["armed"]:
        _GRAD_LOWP["armed"] = True
        torch.autograd.Variable._execution_engine.queue_callback(_grad_lowp_clear)


def _scaled_grad(g2, row_scale, lowp):
    """row_scale[sample] * g2 for the backward of a product that applied the scale (g2 fp32 [T][N]): as bf16 when the backward
    products read bf16 operands (cast and scale in one pass), else fp32 (dlwp_scale_rows_add without a residual)."""
    if lowp:
        hit = _GRAD_LOWP["map"].pop(g2.data_ptr(), None)
        if hit is not None and hit[1] == row_scale.data_ptr() and hit[2] == g2.numel():
            _GRAD_LOWP["hits"] += 1
            return hit[0].reshape(g2.shape)          # the LayerNorm backward that produced g2 wrote it already
        _GRAD_LOWP["misses"] += 1
    nb = row_scale.numel()
    out = torch.empty_like(g2, dtype=_BF if lowp else torch.float32)
    if lowp and (g2.numel() // nb) % 4 == 0:
        L.check(L.load().dlwp_cast_bf16_scaled(L.ptr(g2), L.ptr(row_scale), L.ptr(out), nb, g2.numel() // nb, L.stream()))
        return out
    out = torch.empty_like(g2)
    L.check(L.load().dlwp_scale_rows_add(L.ptr(g2), L.ptr(row_scale), None, L.ptr(out), nb, g2.numel() // nb, L.stream()))
    return out


def _wmat(w, rows):
    """(matrix the forward / input-gradient GEMMs read, fp32 matrix): the bf16 shadow of `w` when the engine keeps one."""
    w32 = w.contiguous().reshape(rows, -1)
    sh = L.shadow(w)
    return (sh.reshape(rows, -1) if sh is not None else w32), w32


def _act_dtype():
    return _BF if (L.storage_bf16() and L.SHADOW_ACTIVE) else torch.float32


# env: A/B runs of the stored-derivative GELU (act 7 forward / act 8 backward, round 5) against act 1 / act 4 (the backward
# product evaluates GELU' from the stored pre-activation)
_ACT_F, _ACT_B = ((7, 8) if __import__("os").environ.get("DLWP_MLP_STORE_D", "1") != "0" else (1, 4))
LOWP_MIN_DEPTH = int(__import__("os").environ.get("DLWP_LOWP_MIN_DEPTH", "768"))      # env: A/B runs of the break-even below


def _lowp(t, depth):
    return _lowp_impl(t, depth)


def _lowp_impl(t, depth):
    """Under bf16 storage: a bf16 copy of an fp32 GEMM operand that is read by two products (one cast launch, then both
    products take the 16-byte bf16 path).  The matrix units round fp32 operands to bf16 anyway, so the results are unchanged.
    `depth` = the other dimension of those products: the cast is one more pass over t, the saving grows with depth -- measured
    break-even between 512 (SFNO MLP: 12.5 -> 13.2 ms with the casts) and 768 (Pangu C4: 22.9 -> 22.6; AFNO 3072: 30.3 -> 28.4)."""
    if t is None or t.dtype == _BF or _act_dtype() != _BF or depth < LOWP_MIN_DEPTH:
        return t
    if _GRAD_LOWP["map"]:
        hit = _GRAD_LOWP["map"].pop(t.data_ptr(), None)
        if hit is not None and hit[1] == 0 and hit[2] == t.numel():
            _GRAD_LOWP["hits"] += 1
            return hit[0].reshape(t.shape)           # the LayerNorm backward that produced t wrote the copy already
    out = torch.empty_like(t, dtype=_BF)
    L.check(L.load().dlwp_cast_bf16(L.ptr(t), L.ptr(out), t.numel(), L.stream()))
    return out


def _grad_slot(p):
    """The preallocated gradient buffer of a leaf parameter (train_engine.flatten_parameters), or None.
    When present, backward kernels accumulate straight into it (no temporary, no autograd add)."""
    if isinstance(p, torch.nn.Parameter) and p.grad is not None and p.grad.is_contiguous():
        return p.grad
    return None


def _dptr(t, offset=0):
    """Device pointer of `t` advanced by `offset` elements (views into bigger buffers are passed as base + offset)."""
    return None if t is None else L.ptr(t) + t.element_size() * offset


def _gemm_batched(A, B, C, M, N, K, lda, ldb, ldc, tA, tB, nb1=1, nb2=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), bias=None, act=0,
                  preact=None, residual=None, sR=(0, 0), res_pre=0, accumulate=0, sBi=(0, 0), act_param=0.0, oA=0, oB=0, oC=0,
                  oR=0, oBi=0):
    """dlwp_gemm_batched; o* are element offsets into the (contiguous) buffers (preact shares C's offset)."""
    dt = _dtypes(A, B, C, preact, residual)
    if dt:
        L.check(L.load().dlwp_gemm_batched_mixed(_dptr(A, oA), _dptr(B, oB), _dptr(C, oC), M, N, K, lda, ldb, ldc, tA, tB, nb1,
                                                 nb2, sA[0], sA[1], sB[0], sB[1], sC[0], sC[1], _dptr(bias, oBi), sBi[0],
                                                 sBi[1], act, act_param, _dptr(preact, oC), _dptr(residual, oR), sR[0],
                                                 sR[1], res_pre, accumulate, dt, L.stream()))
        return
    L.check(L.load().dlwp_gemm_batched(_dptr(A, oA), _dptr(B, oB), _dptr(C, oC), M, N, K, lda, ldb, ldc, tA, tB, nb1, nb2,
                                       sA[0], sA[1], sB[0], sB[1], sC[0], sC[1], _dptr(bias, oBi), sBi[0], sBi[1], act,
                                       act_param, _dptr(preact, oC), _dptr(residual, oR), sR[0], sR[1], res_pre, accumulate,
                                       L.stream()))


class _LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) + residual, or with res_pre y = act(x W^T + b + residual); act = 0 none / 1 GELU.
    weight is [N, K] or a 1x1 convolution weight [N, K, 1, 1] (same memory)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, residual, res_pre=False, out_lowp=False, row_scale=None, wbatch=None):
        """row_scale (fp32 [samples], no activation): y = (x W^T + b) * row_scale[sample] + residual -- stochastic depth of the branch
        this product ends, inside its epilogue (DropPath.branch).
        out_lowp (bf16 storage only, no activation / residual): y is written as bf16 for a consumer that reads bf16 rows (the
        window-attention kernels in their token-layout bf16 mode); the upstream gradient then arrives as bf16 too and feeds both
        backward products without a cast."""
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).contiguous()
        if x2.dtype != _BF:               # a bf16 input comes from a LayerNorm that wrote it for this GEMM (bf16 storage)
            x2 = x2.float()
        T, K = x2.shape
        N = weight.shape[0]
        w, _ = _wmat(weight, N)             # the bf16 copy under bf16 storage (forward and input-gradient products)
        wbatch = wbatch if (wbatch is not None and WGRAD_BATCH and w.dtype == _BF and ctx.needs_input_grad[1] and not act) else None
        if w.dtype == _BF:
            x2 = _lowp(x2, N if (wbatch is None or not WGRAD_FORCE_CAST) else LOWP_MIN_DEPTH)               # read by the forward product and by gW = g^T x
        lowp_out = bool(out_lowp) and w.dtype == _BF and x2.dtype == _BF and not act and residual is None
        y = torch.empty(T, N, device=x.device, dtype=_BF if lowp_out else torch.float32)
        z = torch.empty(T, N, device=x.device) if act else None
        r2 = residual.reshape(-1, N).contiguous() if residual is not None else None
        if row_scale is not None:
            if act or res_pre:
                raise L.DlwpError("linear: row_scale goes with no activation")
            _gemm_rowscale(x2, w, y, T, N, K, K, K, N, 0, 1, bias, r2, row_scale)
        elif res_pre:
            _gemm_batched(x2, w, y, T, N, K, K, K, N, 0, 1, bias=bias, act=act, preact=z, residual=r2, res_pre=1)
        else:
            _gemm(x2, w, y, T, N, K, K, K, N, 0, 1, bias, act, z, r2)
        ctx.row_scale, ctx.wbatch = row_scale, wbatch
        if wbatch is not None:
            wbatch.enrol()
        # the [in][out] bf16 copy (train_engine): the input-gradient product then runs in the k-contiguous form (see _input_grad)
        ctx.wt = L.shadow_t(weight) if (w.dtype == _BF and weight.dim() == 2) else None
        ctx.save_for_backward(x2, w, z)
        ctx.has_bias, ctx.has_res, ctx.act, ctx.shape = bias is not None, residual is not None, act, shape
        ctx.res_pre, ctx.wshape = bool(res_pre) and act, weight.shape
        ctx.in_dtype = x.dtype if x.dtype == _BF else torch.float32
        ctx.wslot = _grad_slot(weight)
        ctx.bslot = _grad_slot(bias) if bias is not None else None
        return y.reshape(*shape[:-1], N)

    @staticmethod
    def backward(ctx, gy):
        lib = L.load()
        x2, w, z = ctx.saved_tensors
        T, K = x2.shape
        N = w.shape[0]
        g2 = gy.reshape(-1, N).contiguous()
        if g2.dtype != _BF or w.dtype != _BF or ctx.act or ctx.row_scale is not None:      # (a bf16 gradient of a bf16 output feeds the bf16 products as it is)
            g2 = g2.float()
        gres = gy if ctx.has_res else None
        if ctx.row_scale is not None:            # the branch's gradient is the scaled one; the residual's (gres) is not
            g2 = _scaled_grad(g2, ctx.row_scale, w.dtype == _BF)
        if ctx.act:
            gz = torch.empty_like(g2)
            L.check(lib.dlwp_gelu_bwd(L.ptr(z), L.ptr(g2), L.ptr(gz), g2.numel(), L.stream()))
            g2 = gz
            if ctx.res_pre and ctx.has_res:      # the residual sits inside the activation
                gres = gz.reshape(gy.shape)
        wb = ctx.wbatch
        if w.dtype == _BF:
            g2 = _lowp(g2, K if (wb is None or not WGRAD_FORCE_CAST) else LOWP_MIN_DEPTH)                    # read by both products below
        gx = torch.empty(T, K, device=g2.device, dtype=ctx.in_dtype)   # a bf16 input (LayerNorm output) takes a bf16 gradient
        if wb is not None:
            if wb.takes(g2, x2, ctx.wslot, ctx.bslot, ctx.has_bias):       # the weight gradient joins the block's one launch
                _input_grad(g2, w, ctx.wt, gx, T, K, N)
                wb.add(g2, x2, ctx.wslot, ctx.bslot, ctx.has_bias, ctx.wshape)
                return gx.reshape(ctx.shape), None, None, None, gres, None, None, None, None
        gw, gb = None, None
        if ctx.has_bias:
            gb = ctx.bslot if ctx.bslot is not None else torch.zeros(N, device=g2.device)
        if ctx.wslot is None:
            gw = torch.empty(N, K, device=g2.device)
        # the two products read the same g and do not depend on each other: one launch while they are small (lib.gemm_group)
        with L.gemm_group():
            _input_grad(g2, w, ctx.wt, gx, T, K, N)                  # gx = g W
            # gW = g^T x, with the bias gradient (column sums of g) produced by the same kernel; both go straight
            # into the parameters' gradient buffers when those exist (fused gradient accumulation)
            if ctx.wslot is not None:
                _gemm(g2, x2, ctx.wslot, N, K, T, N, K, K, 1, 0, accumulate=1, rowsum=gb)
            else:
                _gemm(g2, x2, gw, N, K, T, N, K, K, 1, 0, rowsum=gb)
        if gw is not None:
            gw = gw.reshape(ctx.wshape)
        if ctx.bslot is not None:
            gb = None
        if wb is not None:
            wb.done()
        return gx.reshape(ctx.shape), gw, gb, None, gres, None, None, None, None


def _input_grad(g2, w, wt, gx, T, K, N, **epi):
    """gx [T][K] = g2 [T][N] W, W = w [N][K]: as y = g (W^T)^T on the transposed bf16 copy wt [K][N] when there is one and g2 is a bf16
    array (both operands k-contiguous: the faster form of the LDS-DMA kernels), else on w in the [k][n] form.  epi: _gemm_batched
    epilogue arguments (the MLP's hidden gradient multiplies by the stored GELU derivative)."""
    if wt is not None and g2.dtype == _BF and wt.dtype == _BF and not INPUT_GRAD_NN:
        if epi:
            _gemm_batched(g2, wt, gx, T, K, N, N, N, K, 0, 1, **epi)
        else:
            _gemm(g2, wt, gx, T, K, N, N, N, K, 0, 1)
    elif epi:
        _gemm_batched(g2, w, gx, T, K, N, N, K, K, 0, 0, **epi)
    else:
        _gemm(g2, w, gx, T, K, N, N, K, K, 0, 0)


# env: 1 = the input-gradient products stay in the [k][n] form even where a transposed weight copy exists (A/B runs)
INPUT_GRAD_NN = __import__("os").environ.get("DLWP_INPUT_GRAD_NN", "0") == "1"


def _weight_grad(g2, x2, weight_slot, bias_slot, has_bias, wshape):
    """gW = g^T x with the bias gradient (column sums of g) as a by-product of the same kernel; both land straight in
    the parameters' gradient buffers when those exist.  Returns (gw, gb) for autograd (None where a slot was used)."""
    T, N = g2.shape
    K = x2.shape[1]
    gb = None
    if has_bias:
        gb = bias_slot if bias_slot is not None else torch.zeros(N, device=g2.device)
    if weight_slot is not None:
        _gemm(g2, x2, weight_slot, N, K, T, N, K, K, 1, 0, accumulate=1, rowsum=gb)
        gw = None
    else:
        gw = torch.empty(N, K, device=g2.device)
        _gemm(g2, x2, gw, N, K, T, N, K, K, 1, 0, rowsum=gb)
        gw = gw.reshape(wshape)
    return gw, (None if bias_slot is not None else gb)


class _WgradDesc(C.Structure):
    """dlwp_wgrad_desc (include/dlwpmi.h)"""
    _fields_ = [("g", C.c_void_p), ("x", C.c_void_p), ("gw", C.c_void_p), ("gb", C.c_void_p), ("T", C.c_int), ("N", C.c_int),
                ("K", C.c_int), ("g_bf16", C.c_int), ("x_bf16", C.c_int), ("accumulate", C.c_int)]


def _weight_grad_group(items):
    """[_weight_grad(*item) for item in items] with the products of the whole list in ONE launch (dlwp_weight_grad_group): the
    weight gradients of a block's layers are independent, small, latency-bound split-K products."""
    n = len(items)
    descs = (_WgradDesc * n)()
    outs, keep = [], []
    for i, (g2, x2, wslot, bslot, has_bias, wshape) in enumerate(items):
        T, N = g2.shape
        K = x2.shape[1]
        gb = None
        if has_bias:
            gb = bslot if bslot is not None else torch.zeros(N, device=g2.device)
        gw = wslot if wslot is not None else torch.empty(N, K, device=g2.device)
        keep.append((g2, x2, gw, gb))
        descs[i] = _WgradDesc(L.ptr(g2), L.ptr(x2), L.ptr(gw), L.ptr(gb), T, N, K, int(g2.dtype == _BF), int(x2.dtype == _BF),
                              int(wslot is not None))
        outs.append((None if wslot is not None else gw.reshape(wshape), None if (bslot is not None or gb is None) else gb))
    L.check(L.load().dlwp_weight_grad_group(C.cast(descs, C.c_void_p), n, L.stream()))
    return outs


class _WgradSegProduct(C.Structure):
    """dlwp_wgrad_seg_product (include/dlwpmi.h)"""
    _fields_ = [("g", C.c_void_p * 8), ("x", C.c_void_p * 8), ("gw", C.c_void_p), ("gb", C.c_void_p), ("N", C.c_int), ("K", C.c_int),
                ("overwrite", C.c_int)]


_WGRAD_MAX_SEGMENTS = 8
_workspaces = {}      # device -> [tensors]: the last one is current; earlier (smaller) ones stay alive for the graphs that captured them


def _workspace(nbytes, device):
    """Caller-side scratch of the library calls that ask for one (dlwp_*_workspace_bytes): grow-only per device, never freed while
    the process lives -- a hipGraph that captured a launch keeps reading / writing the buffer it was captured with."""
    held = _workspaces.setdefault(device, [])
    if not held or held[-1].numel() < nbytes:
        held.append(torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device))
    return held[-1]


def _weight_grad_segments(layers):
    """gW_i += sum_s g_is^T x_is, gb_i += sum_s colsum(g_is) for up to four layers in one product launch + one reduction launch
    (dlwp_wgrad_segments).  layers: [(g segments, x segments, weight slot or None, bias slot or None, has_bias, weight shape)],
    every segment a bf16 [T][cols] array, the same number of segments and tokens for every layer.
    Returns [(gw, gb)] for autograd (None where the slot was used)."""
    lib = L.load()
    n = len(layers)
    nseg, T = len(layers[0][0]), layers[0][0][0].shape[0]
    descs = (_WgradSegProduct * n)()
    outs, keep = [], []
    for i, (gs, xs, wslot, bslot, has_bias, wshape) in enumerate(layers):
        N, K = gs[0].shape[1], xs[0].shape[1]
        assert len(gs) == nseg and len(xs) == nseg and all(t.shape[0] == T and t.dtype == _BF for t in (*gs, *xs))
        dev = gs[0].device
        fresh = wslot is None or getattr(wslot, "_dlwp_overwrite", False)      # nothing to add to: the product overwrites (no zero fill)
        gw = wslot if wslot is not None else torch.empty(N, K, device=dev)
        gb = None
        if has_bias:
            gb = bslot if bslot is not None else torch.zeros(N, device=dev)
        d = descs[i]
        for s_ in range(nseg):
            d.g[s_], d.x[s_] = L.ptr(gs[s_]), L.ptr(xs[s_])
        d.gw, d.gb, d.N, d.K, d.overwrite = L.ptr(gw), L.ptr(gb), N, K, int(fresh)
        keep.append((gw, gb))
        outs.append((None if wslot is not None else gw.reshape(wshape), None if (bslot is not None or gb is None) else gb))
    need = lib.dlwp_wgrad_segments_workspace_bytes(C.cast(descs, C.c_void_p), n, nseg, T)
    if need == 0:
        L.check(-1)
    ws = _workspace(need, layers[0][0][0].device)
    L.check(lib.dlwp_wgrad_segments(C.cast(descs, C.c_void_p), n, nseg, T, L.ptr(ws), ws.numel(), L.stream()))
    return outs


# env: A/B runs of the per-block weight-gradient launch (round 5) against one product launch per layer
WGRAD_BATCH = __import__("os").environ.get("DLWP_WGRAD_BATCH", "1") != "0"
# env: 0 = a layer whose operands are not bf16 arrays already computes its weight gradient on its own instead of casting them for the
# block's launch (A/B runs)
WGRAD_FORCE_CAST = __import__("os").environ.get("DLWP_WGRAD_FORCE_CAST", "1") != "0"


class WgradBatch:
    """The weight gradients of ONE application of a transformer block (qkv, proj, fc1, fc2: four products over the same tokens) in one
    dlwp_wgrad_segments launch (+ its reduction) instead of a split-K launch and a slab reduction per layer -- each of those is a
    5 - 10 GFLOP product that cannot fill the chip by itself.  The block's forward makes a fresh object and hands it to its Linear /
    Mlp calls: a node that will compute weight gradients enrols in its forward, and in its backward either hands over its (g, x) pair
    (bf16 arrays, gradient slots present) or computes the product at once and signs off; the launch happens when the last enrolled
    node has reported.  Nothing global: an application whose backward never runs just drops its object with the graph.
    Gradient writes of the block therefore land when its FIRST layer (qkv) runs backward, i.e. before the block's own full-backward
    hook but after the hooks of its sub-modules: a data-parallel unit must not be finer than the block (the block classes carry
    `wgrad_batch_block = True`; ddp.BucketedGradAllReduce checks)."""

    def __init__(self):
        self.enrolled, self.pending, self.items, self._armed = 0, 0, [], False

    def enrol(self, n=1):
        self.enrolled += n
        self.pending += n

    @staticmethod
    def takes(g2, x2, wslot, bslot, has_bias):
        return (WGRAD_BATCH and g2.dtype == _BF and x2.dtype == _BF and wslot is not None and (bslot is not None or not has_bias)
                and g2.shape[1] % 8 == 0 and x2.shape[1] % 8 == 0 and g2.is_contiguous() and x2.is_contiguous()
                and L.ptr(g2) % 16 == 0 and L.ptr(x2) % 16 == 0 and L.ptr(wslot) % 16 == 0)

    def add(self, g2, x2, wslot, bslot, has_bias, wshape):
        self.items.append(([g2], [x2], wslot, bslot, has_bias, wshape))
        self.done()

    def _arm(self):
        """Safety net, once per backward pass: when the pass ends (autograd's engine callback) whatever was handed over but never
        launched -- an enrolled node whose backward did not run in this pass: torch.autograd.grad on a subset, a detached branch --
        is launched then, and the counter is re-armed for another pass over a retained graph."""
        if not self._armed:
            self._armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)

    def _end_of_backward(self):
        self._armed = False
        if self.items:
            self._launch()
        self.pending = self.enrolled

    def _launch(self):
        items, self.items = self.items, []
        by_tokens = {}
        for it in items:
            by_tokens.setdefault(it[0][0].shape[0], []).append(it)
        for group in by_tokens.values():
            for i in range(0, len(group), 4):          # DLWP_WGRAD_MAX_PRODUCTS
                _weight_grad_segments(group[i:i + 4])

    def done(self):
        self._arm()
        self.pending -= 1
        if self.pending < 0:
            raise L.DlwpError("WgradBatch: more backward reports than enrolled nodes (a node's backward ran twice inside one pass)")
        if self.pending == 0 and self.items:
            self._launch()


class _MlpFn(torch.autograd.Function):
    """y = fc2(GELU(fc1 x)) (+ residual) as ONE autograd node (reference: Mlp.forward, nsbench/models/fourcastnet/
    fourcastnet.py:50-56, swintransformer/swin_transformer.py:42-48).  Forward: two GEMMs (bias + GELU, bias + residual
    epilogues).  Backward: four GEMMs and nothing else -- the product g W2 leaves its kernel already multiplied by
    GELU'(z) (epilogue act 4), so the hidden-width gradient is written once and never re-read by an elementwise pass."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual, row_scale=None, wbatch=None):
        """row_scale (fp32 [samples]): y = fc2(...) * row_scale[sample] + residual (DropPath.branch)."""
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).contiguous()
        if x2.dtype != _BF:               # a bf16 input comes from a LayerNorm that wrote it for this GEMM (bf16 storage)
            x2 = x2.float()
        T, K = x2.shape
        Hd, N = w1.shape[0], w2.shape[0]
        # bf16 storage (lib.set_storage): h, z live in HBM as bf16 and the GEMMs read the engine's bf16 weight copies
        (w1m, _), (w2m, _) = _wmat(w1, Hd), _wmat(w2, N)
        adt = _act_dtype()
        if adt == _BF:
            x2 = _lowp(x2, Hd)              # read by fc1 and by gW1 = gh^T x
        h = torch.empty(T, Hd, device=x.device, dtype=adt)
        z = torch.empty(T, Hd, device=x.device, dtype=adt)
        # act 7: z receives GELU'(pre-activation), evaluated with the activation -- the backward product multiplies by it (act 8)
        _gemm(x2, w1m, h, T, Hd, K, K, K, Hd, 0, 1, b1, _ACT_F, z, None)
        y = torch.empty(T, N, device=x.device)
        r2 = residual.reshape(-1, N).contiguous() if residual is not None else None
        if row_scale is not None:
            _gemm_rowscale(h, w2m, y, T, N, Hd, Hd, Hd, N, 0, 1, b2, r2, row_scale)
        else:
            _gemm(h, w2m, y, T, N, Hd, Hd, Hd, N, 0, 1, b2, 0, None, r2)
        ctx.row_scale = row_scale
        ctx.wbatch = wbatch if (wbatch is not None and WGRAD_BATCH and adt == _BF and ctx.needs_input_grad[1] and ctx.needs_input_grad[3]) else None
        if ctx.wbatch is not None:
            ctx.wbatch.enrol(2)
        ctx.wt = (L.shadow_t(w1), L.shadow_t(w2)) if (adt == _BF and w1.dim() == 2 and w2.dim() == 2) else (None, None)
        ctx.save_for_backward(x2, w1m, w2m, z, h)
        ctx.shape, ctx.has_res = shape, residual is not None
        ctx.in_dtype = x.dtype if x.dtype == _BF else torch.float32
        ctx.w1shape, ctx.w2shape = w1.shape, w2.shape
        ctx.slots = (_grad_slot(w1), _grad_slot(b1) if b1 is not None else None,
                     _grad_slot(w2), _grad_slot(b2) if b2 is not None else None)
        ctx.has_b = (b1 is not None, b2 is not None)
        return y.reshape(*shape[:-1], N)

    @staticmethod
    def backward(ctx, gy):
        x2, w1m, w2m, z, h = ctx.saved_tensors
        T, K = x2.shape
        Hd, N = w1m.shape[0], w2m.shape[0]
        g2 = gy.reshape(-1, N).contiguous().float()
        if ctx.row_scale is not None:
            g2 = _scaled_grad(g2, ctx.row_scale, h.dtype == _BF)
        elif h.dtype == _BF:
            g2 = _lowp(g2, Hd if (ctx.wbatch is None or not WGRAD_FORCE_CAST) else LOWP_MIN_DEPTH)                   # read by gh = g W2 and gW2 = g^T h
        gh = torch.empty(T, Hd, device=g2.device, dtype=h.dtype)
        _input_grad(g2, w2m, ctx.wt[1], gh, T, Hd, N, act=_ACT_B, residual=z)             # (g W2) * GELU'(z), z = the stored derivative
        gx = torch.empty(T, K, device=g2.device, dtype=ctx.in_dtype)   # a bf16 input (LayerNorm output) takes a bf16 gradient
        wb = ctx.wbatch
        if wb is not None:
            pairs = [(g2, h, ctx.slots[2], ctx.slots[3], ctx.has_b[1], ctx.w2shape), (gh, x2, ctx.slots[0], ctx.slots[1], ctx.has_b[0], ctx.w1shape)]
            if all(wb.takes(*q[:5]) for q in pairs):       # the weight gradients join the block's one launch
                _input_grad(gh, w1m, ctx.wt[0], gx, T, K, Hd)
                for q in pairs:
                    wb.add(*q)
                return gx.reshape(ctx.shape), None, None, None, None, (gy if ctx.has_res else None), None, None
        # gx = gh W1 and the two weight gradients do not depend on each other: one launch while they are small (lib.gemm_group parks
        # products of at most 2.2 GFLOP; larger ones launch at once)
        if T <= 4096:          # (nsbench AFNO 64 x 64, 1024 tokens: 754 -> 796 samples/s)
            with L.gemm_group():
                _input_grad(gh, w1m, ctx.wt[0], gx, T, K, Hd)
                gw2, gb2 = _weight_grad(g2, h, ctx.slots[2], ctx.slots[3], ctx.has_b[1], ctx.w2shape)
                gw1, gb1 = _weight_grad(gh, x2, ctx.slots[0], ctx.slots[1], ctx.has_b[0], ctx.w1shape)
        else:                  # more tokens: the input gradient on its own kernel, the weight gradients on the dedicated grouped one
            _input_grad(gh, w1m, ctx.wt[0], gx, T, K, Hd)
            (gw2, gb2), (gw1, gb1) = _weight_grad_group([(g2, h, ctx.slots[2], ctx.slots[3], ctx.has_b[1], ctx.w2shape),
                                                         (gh, x2, ctx.slots[0], ctx.slots[1], ctx.has_b[0], ctx.w1shape)])
        if wb is not None:
            wb.done()
            wb.done()
        return gx.reshape(ctx.shape), gw1, gb1, gw2, gb2, (gy if ctx.has_res else None), None, None


# opt-in (DLWP_MLP_STREAM=1): measured 321 / 267 us forward / backward at T = 16200, E = 768 against 238 us for the forward's two GEMMs in
# the C5 step - the one-launch MLP saves the 200 MB hidden round trip but runs its MFMAs at 476 TFLOP/s
MLP_STREAM = __import__("os").environ.get("DLWP_MLP_STREAM", "0") == "1"


class _MlpStreamFn(torch.autograd.Function):
    """_MlpFn with each direction's two activation products as ONE launch (dlwp_mlp_stream_fwd / _bwd, csrc/mlp_stream.hip) for
    wide hidden layers under bf16 operands + bf16 storage: the stored tensors (z, h, gh as bf16) and the weight-gradient
    products are those of _MlpFn; the weights are read from fragment-order images packed from the fp32 master weights."""

    @staticmethod
    def applies(x, w1, w2):
        E, Hd = w1.shape[1], w1.shape[0]
        return (MLP_STREAM and _act_dtype() == _BF and x.is_cuda and x.shape[-1] == E and w1.dim() == 2 and w2.shape == (E, Hd)
                and L.load().dlwp_mlp_stream_supported(E, Hd) == 1)

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual):
        lib = L.load()
        shape = x.shape
        E, Hd = w1.shape[1], w1.shape[0]
        x2 = x.reshape(-1, E).contiguous()
        if x2.dtype != _BF:
            x2 = x2.float()
        T, dev = x2.shape[0], x.device
        imgs = torch.empty(4, E * Hd, device=dev, dtype=_BF)
        L.check(lib.dlwp_mlp_stream_pack(L.ptr(w1.detach().contiguous()), L.ptr(w2.detach().contiguous()), E, Hd, L.ptr(imgs), L.stream()))
        x_lp = x2 if x2.dtype == _BF else torch.empty(T, E, device=dev, dtype=_BF)
        z = torch.empty(T, Hd, device=dev, dtype=_BF)
        h = torch.empty(T, Hd, device=dev, dtype=_BF)
        y = torch.empty(T, E, device=dev)
        r2 = residual.reshape(-1, E).contiguous().float() if residual is not None else None
        L.check(lib.dlwp_mlp_stream_fwd(L.ptr(x2), int(x2.dtype == _BF), None if x2.dtype == _BF else L.ptr(x_lp), L.ptr(imgs[0]), L.ptr(b1),
                                        L.ptr(imgs[1]), L.ptr(b2), L.ptr(r2), L.ptr(z), L.ptr(h), L.ptr(y), T, E, Hd, L.stream()))
        ctx.save_for_backward(x_lp, z, h, imgs)
        ctx.shape, ctx.has_res = shape, residual is not None
        ctx.in_dtype = x.dtype if x.dtype == _BF else torch.float32
        ctx.w1shape, ctx.w2shape = w1.shape, w2.shape
        ctx.slots = (_grad_slot(w1), _grad_slot(b1) if b1 is not None else None, _grad_slot(w2), _grad_slot(b2) if b2 is not None else None)
        ctx.has_b = (b1 is not None, b2 is not None)
        return y.reshape(*shape[:-1], E)

    @staticmethod
    def backward(ctx, gy):
        lib = L.load()
        x_lp, z, h, imgs = ctx.saved_tensors
        T, E = x_lp.shape
        Hd = h.shape[1]
        dev = x_lp.device
        g32 = gy.reshape(-1, E).contiguous().float()
        g_lp = torch.empty(T, E, device=dev, dtype=_BF)
        gh = torch.empty(T, Hd, device=dev, dtype=_BF)
        gx = torch.empty(T, E, device=dev, dtype=ctx.in_dtype)
        L.check(lib.dlwp_mlp_stream_bwd(L.ptr(g32), L.ptr(g_lp), L.ptr(imgs[2]), L.ptr(imgs[3]), L.ptr(z), L.ptr(gh), L.ptr(gx),
                                        int(ctx.in_dtype == _BF), T, E, Hd, L.stream()))
        (gw2, gb2), (gw1, gb1) = _weight_grad_group([(g_lp, h, ctx.slots[2], ctx.slots[3], ctx.has_b[1], ctx.w2shape),
                                                     (gh, x_lp, ctx.slots[0], ctx.slots[1], ctx.has_b[0], ctx.w1shape)])
        return gx.reshape(ctx.shape), gw1, gb1, gw2, gb2, (gy if ctx.has_res else None)


def mlp(x, w1, b1, w2, b2, residual=None, row_scale=None, wbatch=None):
    if row_scale is None and _MlpStreamFn.applies(x, w1, w2):
        return _MlpStreamFn.apply(x, w1, b1, w2, b2, residual)
    y = _MlpFn.apply(x, w1, b1, w2, b2, residual, row_scale, wbatch)
    # this node's backward casts its upstream gradient to bf16 for its two products (_lowp): a LayerNorm that reads y writes that copy
    # in its own backward instead (_LayerNormFn / dlwp_layernorm_bwd_lowp; with a stochastic-depth scale DropPath.branch tags y itself)
    if (row_scale is None and LN_BWD_LOWP and _act_dtype() == _BF
            and (w1.shape[0] >= LOWP_MIN_DEPTH or (wbatch is not None and WGRAD_BATCH and WGRAD_FORCE_CAST))):
        y._dlwp_branch_scale = "plain"
    return y


class _SkipMlpFn(torch.autograd.Function):
    """out = fc2(GELU(fc1 t)) (+ x),  t = GELU(y + skip(x)): the tail of an SFNO block (inner linear skip, activation,
    MLP, outer identity skip) as ONE autograd node: three GEMMs forward, six GEMMs backward, both GELU
    derivatives applied in GEMM epilogues (act 4) and the outer skip's gradient added by the epilogue of the inner
    skip's input-gradient product -- no elementwise kernel and no autograd accumulation on the token tensors."""

    @staticmethod
    def forward(ctx, y, x, ws, bs, w1, b1, w2, b2, outer):
        shape = x.shape
        C = shape[-1]
        x2 = x.reshape(-1, C).contiguous().float()
        y2 = y.reshape(-1, C).contiguous().float()
        T = x2.shape[0]
        Hd, N = w1.shape[0], w2.shape[0]
        (wsm, _), (w1m, _), (w2m, _) = _wmat(ws, C), _wmat(w1, Hd), _wmat(w2, N)
        adt = _act_dtype()                  # bf16 storage: the hidden-width tensors h, z1 (and gh) are bf16 in HBM
        # t only feeds GEMMs (fc1 and gW1 = gh^T t), which round it to bf16 when they read it: under bf16 storage the epilogue
        # writes that rounding once (same values, half the bytes, and both consumers become bf16 x bf16 products); the stored
        # pre-activation z0 keeps t's type (a GEMM writes its output and pre-activation in one storage type)
        t = torch.empty(T, C, device=x.device, dtype=adt)
        z0 = torch.empty(T, C, device=x.device, dtype=adt)
        _gemm_batched(x2, wsm, t, T, C, C, C, C, C, 0, 1, bias=bs, act=_ACT_F, preact=z0, residual=y2, res_pre=1)     # z0, z1: GELU'(.)
        h = torch.empty(T, Hd, device=x.device, dtype=adt)
        z1 = torch.empty(T, Hd, device=x.device, dtype=adt)
        _gemm(t, w1m, h, T, Hd, C, C, C, Hd, 0, 1, b1, _ACT_F, z1, None)
        out = torch.empty(T, N, device=x.device)
        _gemm(h, w2m, out, T, N, Hd, Hd, Hd, N, 0, 1, b2, 0, None, x2 if outer else None)
        ctx.save_for_backward(x2, wsm, w1m, w2m, z0, t, z1, h)
        ctx.shape, ctx.outer = shape, bool(outer)
        ctx.wshapes = (ws.shape, w1.shape, w2.shape)
        ctx.slots = tuple(_grad_slot(p) if p is not None else None for p in (ws, bs, w1, b1, w2, b2))
        ctx.has_b = (bs is not None, b1 is not None, b2 is not None)
        return out.reshape(*shape[:-1], N)

    @staticmethod
    def backward(ctx, gout):
        x2, wsm, w1m, w2m, z0, t, z1, h = ctx.saved_tensors
        T, C = x2.shape
        Hd, N = w1m.shape[0], w2m.shape[0]
        g32 = gout.reshape(-1, N).contiguous().float()
        g2 = _lowp(g32, Hd) if h.dtype == _BF else g32      # read by gh = g W2 and gW2 = g^T h
        gh = torch.empty(T, Hd, device=g2.device, dtype=h.dtype)
        _gemm_batched(g2, w2m, gh, T, Hd, N, N, Hd, Hd, 0, 0, act=_ACT_B, residual=z1)      # (g W2) * GELU'(z1)
        gt = torch.empty(T, C, device=g2.device)
        _gemm_batched(gh, w1m, gt, T, C, Hd, Hd, C, C, 0, 0, act=_ACT_B, residual=z0)       # (gh W1) * GELU'(z0) = d/d(y + skip)
        gx = torch.empty(T, C, device=g2.device)
        _gemm_batched(gt, wsm, gx, T, C, C, C, C, C, 0, 0, residual=g32 if ctx.outer else None)   # + outer skip
        # the three weight gradients (+ bias gradients) do not depend on each other: one launch
        (gw2, gb2), (gw1, gb1), (gws, gbs) = _weight_grad_group([
            (g2, h, ctx.slots[4], ctx.slots[5], ctx.has_b[2], ctx.wshapes[2]),
            (gh, t, ctx.slots[2], ctx.slots[3], ctx.has_b[1], ctx.wshapes[1]),
            (gt, x2, ctx.slots[0], ctx.slots[1], ctx.has_b[0], ctx.wshapes[0])])
        return gt.reshape(ctx.shape), gx.reshape(ctx.shape), gws, gbs, gw1, gb1, gw2, gb2, None


class _TailFwdArgs(C.Structure):
    """dlwp_sfno_tail_fwd_args (include/dlwpmi.h)"""
    _fields_ = [(n, C.c_void_p) for n in ("x", "y", "ws_img", "w1_img", "w2_img", "bs", "b1", "b2", "x_lp", "z0", "t", "z1", "h",
                                          "out")] + [(n, C.c_int) for n in ("T", "C", "hidden", "outer", "y_bf16")]


class _TailBwdArgs(C.Structure):
    """dlwp_sfno_tail_bwd_args (include/dlwpmi.h)"""
    _fields_ = [(n, C.c_void_p) for n in ("g", "w2t_img", "w1t_img", "wst_img", "z1", "z0", "g_lp", "gh", "gt", "gt_lp", "gx")] + \
               [(n, C.c_int) for n in ("T", "C", "hidden", "outer")]


def _chain_images(ws, w1, w2):
    """The six fragment-order bf16 images of an SFNO block tail's weights (dlwp_mlp_chain_pack: forward Ws, W1, W2; backward
    W2^T, W1^T, Ws^T), built from the fp32 master weights.  Inside a sht.spectral_weight_scope (one rollout pass: the parameters
    cannot change) they are built once and shared by every lead time, together with the list that collects the lead times'
    weight-gradient operands (the last backward pass through these weights runs ONE product over all of them).  -> (images, state)"""
    from . import sht
    scope = sht._wexp_scope
    key = ("chain", id(ws), id(w1), id(w2))
    if scope is not None and key in scope:
        return scope[key][0], scope[key][2]
    lib = L.load()
    C_, Hd = ws.shape[0], w1.shape[0]
    imgs = torch.empty(6, C_ * Hd, device=ws.device, dtype=_BF)       # (the two C x C images use the front of their rows)
    L.check(lib.dlwp_sfno_tail_pack(L.ptr(ws.detach().contiguous()), L.ptr(w1.detach().contiguous()), L.ptr(w2.detach().contiguous()),
                                    C_, Hd, L.ptr(imgs), L.stream()))
    # uses: applications of these weights whose backward pass is still to come; pending: their weight-gradient operands
    state = {"uses": 0, "pending": []}
    if scope is not None:
        scope[key] = (imgs, (ws, w1, w2), state)      # keeps the parameters alive: the ids stay unique inside the scope
    return imgs, state


def prepack_chain_images(tails):
    """Inside a sht.spectral_weight_scope: the images of every block tail in `tails` ([(ws, w1, w2)], equal widths) that has none
    yet, in ONE launch (dlwp_sfno_tail_pack_many) instead of one per block at its first use."""
    from . import sht
    scope = sht._wexp_scope
    if scope is None or not tails or not CHAIN_TAIL or _act_dtype() != _BF:
        return
    todo = [t for t in tails if ("chain", id(t[0]), id(t[1]), id(t[2])) not in scope]
    if not todo or len(todo) > 8 or any(t[0].shape != todo[0][0].shape or t[1].shape != todo[0][1].shape or not t[0].is_cuda for t in todo):
        return
    lib = L.load()
    C_, Hd = todo[0][0].shape[0], todo[0][1].shape[0]
    if lib.dlwp_mlp_chain_supported(C_, Hd) != 1:
        return
    imgs = torch.empty(len(todo), 6, C_ * Hd, device=todo[0][0].device, dtype=_BF)
    srcs = [[w.detach().contiguous() for w in t] for t in todo]
    arr = lambda k: (C.c_void_p * len(todo))(*[L.ptr(s_[k]) for s_ in srcs])
    ip = (C.c_void_p * len(todo))(*[L.ptr(imgs[i]) for i in range(len(todo))])
    L.check(lib.dlwp_sfno_tail_pack_many(arr(0), arr(1), arr(2), len(todo), C_, Hd, ip, L.stream()))
    for i, t in enumerate(todo):
        scope[("chain", id(t[0]), id(t[1]), id(t[2]))] = (imgs[i], t, {"uses": 0, "pending": []})


CHAIN_TAIL = __import__("os").environ.get("DLWP_SFNO_CHAIN", "1") != "0"      # env: A/B runs against the three-GEMM tail


class _SkipMlpChainFn(torch.autograd.Function):
    """_SkipMlpFn with each direction as ONE launch (dlwp_sfno_tail_fwd / _bwd, csrc/mlp_chain.hip): bf16 operands and bf16
    storage only (the arithmetic and the stored tensors are those of _SkipMlpFn under lib.set_storage("bf16")), widths with a
    compiled kernel (dlwp_mlp_chain_supported)."""

    @staticmethod
    def applies(x, ws, w1, w2):
        C_ = x.shape[-1]
        return (CHAIN_TAIL and _act_dtype() == _BF and x.is_cuda and tuple(ws.shape[:2]) == (C_, C_) and w2.shape[0] == C_
                and w1.shape[1] == C_ and w2.shape[1] == w1.shape[0]
                and L.load().dlwp_mlp_chain_supported(C_, w1.shape[0]) == 1)

    @staticmethod
    def forward(ctx, y, x, ws, bs, w1, b1, w2, b2, outer):
        shape = x.shape
        C_ = shape[-1]
        x2 = x.reshape(-1, C_).contiguous().float()
        # y as a bf16 array (the spectral filter's synthesis wrote it that way for this launch: sht.InverseRealSHT(field_bf16=True))
        # is read as it is; its gradient then leaves as the bf16 copy the weight-gradient product needs anyway
        y2 = y.reshape(-1, C_).contiguous()
        if y2.dtype != _BF:
            y2 = y2.float()
        ctx.y_bf16 = y2.dtype == _BF
        T, Hd = x2.shape[0], w1.shape[0]
        imgs, state = _chain_images(ws, w1, w2)
        if any(ctx.needs_input_grad):
            state["uses"] += 1
        ctx.state = state
        dev = x.device
        x_lp = torch.empty(T, C_, device=dev, dtype=_BF)
        z0 = torch.empty(T, C_, device=dev, dtype=_BF)
        t = torch.empty(T, C_, device=dev, dtype=_BF)
        z1 = torch.empty(T, Hd, device=dev, dtype=_BF)
        h = torch.empty(T, Hd, device=dev, dtype=_BF)
        out = torch.empty(T, C_, device=dev)
        a = _TailFwdArgs(L.ptr(x2), L.ptr(y2), L.ptr(imgs[0]), L.ptr(imgs[1]), L.ptr(imgs[2]), L.ptr(bs), L.ptr(b1), L.ptr(b2),
                         L.ptr(x_lp), L.ptr(z0), L.ptr(t), L.ptr(z1), L.ptr(h), L.ptr(out), T, C_, Hd, int(bool(outer)),
                         int(ctx.y_bf16))
        L.check(L.load().dlwp_sfno_tail_fwd(C.byref(a), L.stream()))
        ctx.save_for_backward(x_lp, z0, t, z1, h, imgs)
        ctx.shape, ctx.outer = shape, bool(outer)
        ctx.wshapes = (ws.shape, w1.shape, w2.shape)
        ctx.slots = tuple(_grad_slot(p) if p is not None else None for p in (ws, bs, w1, b1, w2, b2))
        ctx.has_b = (bs is not None, b1 is not None, b2 is not None)
        return out.reshape(*shape[:-1], C_)

    @staticmethod
    def backward(ctx, gout):
        x_lp, z0, t, z1, h, imgs = ctx.saved_tensors
        T, C_ = x_lp.shape
        Hd = h.shape[1]
        dev = x_lp.device
        g32 = gout.reshape(-1, C_).contiguous().float()
        g_lp = torch.empty(T, C_, device=dev, dtype=_BF)
        gh = torch.empty(T, Hd, device=dev, dtype=_BF)
        gt_lp = torch.empty(T, C_, device=dev, dtype=_BF)
        gt = gt_lp if ctx.y_bf16 else torch.empty(T, C_, device=dev)          # a bf16 y takes a bf16 gradient: no fp32 copy is written
        gx = torch.empty(T, C_, device=dev)
        a = _TailBwdArgs(L.ptr(g32), L.ptr(imgs[3]), L.ptr(imgs[4]), L.ptr(imgs[5]), L.ptr(z1), L.ptr(z0), L.ptr(g_lp), L.ptr(gh),
                         None if ctx.y_bf16 else L.ptr(gt), L.ptr(gt_lp), L.ptr(gx), T, C_, Hd, int(ctx.outer))
        L.check(L.load().dlwp_sfno_tail_bwd(C.byref(a), L.stream()))
        # The three weight gradients (+ bias gradients).  A rollout applies these weights once per lead time: the operand pairs
        # wait in the shared list and the LAST backward pass through the weights multiplies all of them in one product over the
        # concatenated token axis (dlwp_wgrad_segments) -- one launch pair per block and step instead of one per lead time.
        st = ctx.state
        st["pending"].append((g_lp, h, gh, t, gt_lp, x_lp))
        st["uses"] -= 1
        if st["uses"] > 0 and len(st["pending"]) < _WGRAD_MAX_SEGMENTS:
            return gt.reshape(ctx.shape), gx.reshape(ctx.shape), None, None, None, None, None, None, None
        segs, st["pending"] = st["pending"], []
        col = lambda i: [s_[i] for s_ in segs]
        (gw2, gb2), (gw1, gb1), (gws, gbs) = _weight_grad_segments([
            (col(0), col(1), ctx.slots[4], ctx.slots[5], ctx.has_b[2], ctx.wshapes[2]),
            (col(2), col(3), ctx.slots[2], ctx.slots[3], ctx.has_b[1], ctx.wshapes[1]),
            (col(4), col(5), ctx.slots[0], ctx.slots[1], ctx.has_b[0], ctx.wshapes[0])])
        return gt.reshape(ctx.shape), gx.reshape(ctx.shape), gws, gbs, gw1, gb1, gw2, gb2, None


def skip_mlp(y, x, ws, bs, w1, b1, w2, b2, outer=True):
    if y.shape == x.shape and _SkipMlpChainFn.applies(x, ws, w1, w2):
        return _SkipMlpChainFn.apply(y, x, ws, bs, w1, b1, w2, b2, outer)
    return _SkipMlpFn.apply(y, x, ws, bs, w1, b1, w2, b2, outer)


def linear(x, weight, bias=None, act=0, residual=None, res_pre=False):
    return _LinearFn.apply(x, weight, bias, act, residual, res_pre)


class Conv1x1(nn.Conv2d):
    """nn.Conv2d(kernel_size=1) parameters (weight [out, in, 1, 1]) applied to channels-last tokens [..., in]:
    one MFMA GEMM with the bias / GELU / residual epilogue."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__(in_channels, out_channels, kernel_size=1, bias=bias)

    def forward(self, tokens, act=0, residual=None, res_pre=False):
        return _LinearFn.apply(tokens, self.weight, self.bias, act, residual, res_pre)


class _LayerNormFn(torch.autograd.Function):
    """LayerNorm over the last dimension.  With fork=True the node returns (x, LN(x)): the pre-norm residual blocks
    (`x + f(norm(x))`) take their skip connection from the first output, so that the gradient of the residual branch and the
    LayerNorm backward meet inside dlwp_layernorm_bwd_res -- one kernel, no elementwise add on the autograd thread."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, fork=False, gemm_input=False):
        lib = L.load()
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).contiguous().float()
        T, C_ = x2.shape
        # gemm_input: the caller promises that only GEMMs read the output -- under bf16 storage it is then written as bf16
        lowp = gemm_input and _act_dtype() == _BF
        y = torch.empty_like(x2, dtype=_BF if lowp else torch.float32)
        mean = torch.empty(T, device=x.device)
        rstd = torch.empty(T, device=x.device)
        L.check(lib.dlwp_layernorm_fwd_ex(L.ptr(x2), L.ptr(gamma.contiguous()), L.ptr(beta.contiguous()), L.ptr(y),
                                          L.ptr(mean), L.ptr(rstd), T, C_, eps, int(lowp), L.stream()))
        ctx.save_for_backward(x2, gamma, mean, rstd)
        ctx.shape, ctx.fork = shape, fork
        ctx.slots = (_grad_slot(gamma), _grad_slot(beta))
        # x is the output of a residual branch that applied a per-sample scale (DropPath.branch tags it): that branch's backward wants
        # bf16(gx * scale) -- this node's backward kernel writes it beside gx
        sc = getattr(x, "_dlwp_branch_scale", None) if LN_BWD_LOWP else None
        ctx.next_plain = isinstance(sc, str) and sc == "plain" and _act_dtype() == _BF and C_ % 4 == 0      # (an unscaled copy, see mlp())
        ctx.next_scale = sc if (isinstance(sc, torch.Tensor) and _act_dtype() == _BF and x.dim() == 3 and sc.numel() == shape[0]
                                and sc.dtype == torch.float32 and C_ % 4 == 0) else None
        if fork:
            return x.view_as(x), y.reshape(shape)
        return y.reshape(shape)

    @staticmethod
    def backward(ctx, *grads):
        lib = L.load()
        gres, gy = grads if ctx.fork else (None, grads[0])
        if gy is None:                      # only the skip output was used
            return gres, None, None, None, None, None
        x2, gamma, mean, rstd = ctx.saved_tensors
        T, C_ = x2.shape
        g2 = gy.reshape(-1, C_).contiguous()          # bf16 when the (bf16) output fed a GEMM: read as it is
        if g2.dtype != _BF:
            g2 = g2.float()
        r2 = gres.reshape(-1, C_).contiguous().float() if gres is not None else None
        gx = torch.empty_like(x2)
        fused = ctx.slots[0] is not None and ctx.slots[1] is not None
        gg, gb = ctx.slots if fused else (torch.zeros_like(gamma), torch.zeros_like(gamma))
        if ctx.next_scale is not None or ctx.next_plain:
            lowp = torch.empty_like(gx, dtype=_BF)
            nb = ctx.next_scale.numel() if ctx.next_scale is not None else 1
            L.check(lib.dlwp_layernorm_bwd_lowp(L.ptr(x2), L.ptr(gamma.contiguous()), L.ptr(mean), L.ptr(rstd), L.ptr(g2), int(g2.dtype == _BF),
                                                L.ptr(r2), L.ptr(gx), L.ptr(gg), L.ptr(gb), T, C_, L.ptr(lowp), L.ptr(ctx.next_scale),
                                                T // nb, L.stream()))
            _grad_lowp_put(gx, lowp, ctx.next_scale)
        else:
            L.check(lib.dlwp_layernorm_bwd_ex(L.ptr(x2), L.ptr(gamma.contiguous()), L.ptr(mean), L.ptr(rstd), L.ptr(g2),
                                              int(g2.dtype == _BF), L.ptr(r2), L.ptr(gx), L.ptr(gg), L.ptr(gb), T, C_, L.stream()))
        if fused:
            return gx.reshape(ctx.shape), None, None, None, None, None
        return gx.reshape(ctx.shape), gg, gb, None, None, None


class _ScaleRowsAddFn(torch.autograd.Function):
    """out[b] = residual[b] + scale[b] * t[b] (residual optional): one streaming kernel (dlwp_scale_rows_add); the
    backward of the branch is the same kernel without the residual."""

    @staticmethod
    def forward(ctx, t, scale, residual):
        t2 = t.contiguous().float()
        r2 = residual.contiguous().float() if residual is not None else None
        B = t2.shape[0]
        out = torch.empty_like(t2)
        L.check(L.load().dlwp_scale_rows_add(L.ptr(t2), L.ptr(scale), L.ptr(r2), L.ptr(out), B, t2.numel() // B, L.stream()))
        ctx.save_for_backward(scale)
        ctx.has_res, ctx.t_dtype = residual is not None, t.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        g2 = g.contiguous().float()
        B = g2.shape[0]
        gt = torch.empty_like(g2)
        L.check(L.load().dlwp_scale_rows_add(L.ptr(g2), L.ptr(scale), None, L.ptr(gt), B, g2.numel() // B, L.stream()))
        return gt.to(ctx.t_dtype), None, (g if ctx.has_res else None)


class _AddFn(torch.autograd.Function):
    """a + b (same shape) as one dlwp_scale_rows_add launch without scales; both gradients are the upstream gradient itself."""

    @staticmethod
    def forward(ctx, a, b):
        a2, b2 = a.contiguous().float(), b.contiguous().float()
        out = torch.empty_like(a2)
        L.check(L.load().dlwp_scale_rows_add(L.ptr(a2), None, L.ptr(b2), L.ptr(out), 1, a2.numel(), L.stream()))
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


def add_tokens(a, b):
    """a + b on equally shaped token tensors through libdlwpmi."""
    assert a.shape == b.shape, (a.shape, b.shape)
    return _AddFn.apply(a, b)


class _AddBcastFn(torch.autograd.Function):
    """tokens [B, ...] + a parameter [1, ...] broadcast over the batch (position embedding).  The parameter's gradient is the
    batch sum of the upstream gradient, accumulated straight into its .grad when that buffer exists."""

    @staticmethod
    def forward(ctx, t, p):
        t2 = t.contiguous().float()
        B = t2.shape[0]
        assert p.numel() * B == t2.numel(), (tuple(t.shape), tuple(p.shape))
        out = torch.empty_like(t2)
        L.check(L.load().dlwp_add_bcast(L.ptr(t2), L.ptr(p.contiguous()), L.ptr(out), B, p.numel(), L.stream()))
        ctx.slot, ctx.pshape = _grad_slot(p), p.shape
        return out

    @staticmethod
    def backward(ctx, g):
        g2 = g.contiguous().float()
        B = g2.shape[0]
        n = g2.numel() // B
        gp = ctx.slot if ctx.slot is not None else torch.zeros(ctx.pshape, device=g.device)
        L.check(L.load().dlwp_colsum(L.ptr(g2), L.ptr(gp), B, n, L.stream()))
        return g, (None if ctx.slot is not None else gp)


def add_pos_embed(tokens, pos):
    """tokens [B, P, E] + pos [1, P, E]."""
    return _AddBcastFn.apply(tokens, pos)


def norm_fork(norm, x, gemm_input=False):
    """(skip, norm(x)) of a pre-norm residual block; LayerNorm.fork where the norm layer offers it."""
    return norm.fork(x, gemm_input) if hasattr(norm, "fork") else (x, norm(x))


# env: A/B runs of the DropPath scale inside the GEMM epilogues (round 5) against the separate dlwp_scale_rows_add pass
DROPPATH_FUSED = __import__("os").environ.get("DLWP_DROPPATH_FUSED", "1") != "0"


class DropPath(nn.Module):
    """Stochastic depth per sample (timm.models.layers.DropPath, used at nsbench/models/swintransformer/
    swin_transformer.py:193,255-256, dlwpbench twin :192,261-262 and panguweather.py:262-323): in training mode the whole
    residual branch of a sample is dropped with probability p and the survivors are scaled by 1 / (1 - p).  The Bernoulli
    draw is torch's (B numbers); applying it is one libdlwpmi kernel fused with the residual add.  `active` tells the
    blocks whether to take this path at all (eval mode / p = 0 keep the residual inside the GEMM epilogues)."""

    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.p = self.drop_prob = float(drop_prob)
        self.scale_by_keep = scale_by_keep
        self._pool, self._pool_index = None, 0

    @property
    def active(self):
        return self.training and self.p > 0.0

    def mask(self, batch, device):
        """This call's per-sample scales (0 or 1 / keep), fp32 [batch]: from the model's pool when it drew for this call."""
        mask = self._pool.take(self._pool_index, batch) if self._pool is not None else None
        if mask is None:
            keep = 1.0 - self.p
            mask = torch.empty(batch, device=device, dtype=torch.float32).bernoulli_(keep)
            if keep > 0.0 and self.scale_by_keep:
                mask = mask / keep
        return mask

    def forward(self, t, residual=None):
        if not self.active:
            return t if residual is None else residual + t
        return _ScaleRowsAddFn.apply(t, self.mask(t.shape[0], t.device), residual)

    def branch(self, fn, t, residual, **kw):
        """residual + drop_path(fn(t)) for a Linear / Mlp module `fn` (residual [batch, tokens, C]): the per-sample scale and the residual
        add run in the epilogue of fn's last product (dlwp_gemm_rowscale) instead of a pass of their own over the tokens."""
        if not self.active:
            return fn(t, residual=residual, **kw)
        if not DROPPATH_FUSED:
            return self(fn(t, **kw), residual=residual)
        mask = self.mask(residual.shape[0], t.device)
        y = fn(t, residual=residual, row_scale=mask, **kw)
        y._dlwp_branch_scale = mask      # a LayerNorm that reads y writes this branch's scaled bf16 gradient in its backward (_LayerNormFn)
        return y

    def extra_repr(self):
        return f"drop_prob={self.p:0.3f}"


class DropPathPool:
    """The keep masks of ALL DropPath modules of a model for one network call, drawn together: one Bernoulli kernel and one
    scaling kernel per call instead of two per block (the masks are independent per block and per sample either way; each
    block keeps its own drop probability).  The model calls draw() at the top of its one-step forward in training mode; a mask
    is handed out once per draw, so a DropPath called outside that protocol falls back to drawing its own."""

    USES = 2      # a block applies its DropPath twice per call (attention branch, MLP branch): independent masks for both

    def __init__(self, model):
        self.mods = [m for m in model.modules() if isinstance(m, DropPath) and m.p > 0.0]
        for i, m in enumerate(self.mods):
            m._pool, m._pool_index = self, i
        self._probs = self._inv = self._masks = None
        self._taken = []

    def draw(self, batch, device):
        if not self.mods or not self.mods[0].training:
            return
        if self._probs is None or self._probs.shape[1] != batch or self._probs.device != torch.device(device):
            keep = torch.tensor([1.0 - m.p for m in self.mods for _ in range(self.USES)], dtype=torch.float32)
            inv = torch.tensor([1.0 / (1.0 - m.p) if (m.p < 1.0 and m.scale_by_keep) else 1.0
                                for m in self.mods for _ in range(self.USES)])
            self._probs = keep[:, None].expand(keep.numel(), batch).contiguous().to(device)
            self._inv = inv[:, None].to(device)
        self._masks = torch.bernoulli(self._probs) * self._inv
        self._taken = [0] * len(self.mods)

    def take(self, i, batch):
        if self._masks is None or self._taken[i] >= self.USES or self._masks.shape[1] != batch:
            return None
        self._taken[i] += 1
        return self._masks[self.USES * i + self._taken[i] - 1]


class _InstanceNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, residual):
        lib = L.load()
        shape = x.shape
        B, C_ = shape[0], shape[-1]
        x3 = x.reshape(B, -1, C_).contiguous().float()
        P = x3.shape[1]
        r3 = residual.reshape(B, -1, C_).contiguous().float() if residual is not None else None
        y = torch.empty_like(x3)
        stats = torch.empty(B, C_, 2, device=x.device)
        L.check(lib.dlwp_instnorm_fwd(L.ptr(x3), L.ptr(gamma.contiguous()), L.ptr(beta.contiguous()), L.ptr(r3), L.ptr(y),
                                      L.ptr(stats), B, P, C_, eps, L.stream()))
        ctx.save_for_backward(x3, gamma, stats)
        ctx.shape, ctx.has_res = shape, residual is not None
        ctx.slots = (_grad_slot(gamma), _grad_slot(beta))
        return y.reshape(shape)

    @staticmethod
    def backward(ctx, gy):
        lib = L.load()
        x3, gamma, stats = ctx.saved_tensors
        B, P, C_ = x3.shape
        g3 = gy.reshape(B, P, C_).contiguous().float()
        gx = torch.empty_like(x3)
        work = torch.empty(B, C_, 2, device=x3.device)
        fused = ctx.slots[0] is not None and ctx.slots[1] is not None
        gg, gb = ctx.slots if fused else (torch.zeros_like(gamma), torch.zeros_like(gamma))
        L.check(lib.dlwp_instnorm_bwd(L.ptr(x3), L.ptr(gamma.contiguous()), L.ptr(stats), L.ptr(g3), L.ptr(gx), L.ptr(gg),
                                      L.ptr(gb), L.ptr(work), B, P, C_, L.stream()))
        gres = gy if ctx.has_res else None
        if fused:
            return gx.reshape(ctx.shape), None, None, None, gres
        return gx.reshape(ctx.shape), gg, gb, None, gres


class InstanceNorm(nn.Module):
    """nn.InstanceNorm2d(num_features, eps, affine=True, track_running_stats=False) on channels-last tokens
    [B, H, W, C] (statistics per sample and channel over the H*W tokens); parameter names `weight` / `bias` as torch's."""

    def __init__(self, num_features, eps=1e-6):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))

    def forward(self, x, residual=None):
        return _InstanceNormFn.apply(x, self.weight, self.bias, float(self.eps), residual)


class Linear(nn.Linear):
    def forward(self, x, act=0, residual=None, out_lowp=False, row_scale=None, wbatch=None):
        return _LinearFn.apply(x, self.weight, self.bias, act, residual, False, out_lowp, row_scale, wbatch)


class LayerNorm(nn.LayerNorm):
    def forward(self, x, gemm_input=False, fork=False):
        return _LayerNormFn.apply(x, self.weight, self.bias, float(self.eps), fork, gemm_input)

    def fork(self, x, gemm_input=False):
        """(x, norm(x)) for a pre-norm residual block: use the first item as the skip connection (see _LayerNormFn).
        gemm_input=True: nothing but GEMMs (Linear / Mlp) reads norm(x); it may then be stored as bf16 (lib.set_storage).
        Goes through __call__ so that module hooks (ddp.BucketedGradAllReduce counts entries and exits) see the call."""
        return self(x, gemm_input, True)


class Mlp(nn.Module):
    """fc1 -> GELU -> fc2 (+ residual): two MFMA GEMMs, GELU and the residual add fused in the epilogues."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        if drop:
            raise NotImplementedError("dropout is not on the MI355X hot path (configs use 0.0)")
        if act_layer is not nn.GELU:
            raise NotImplementedError("only GELU token MLPs are on the hot path")
        self.fc1 = Linear(in_features, hidden_features or in_features)
        self.act = nn.GELU()   # kept for module-tree compatibility; applied inside fc1's epilogue
        self.fc2 = Linear(hidden_features or in_features, out_features or in_features)

    def forward(self, x, residual=None, row_scale=None, wbatch=None):
        return mlp(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, residual, row_scale, wbatch)


class PatchConv2d(nn.Conv2d):
    """Conv2d with kernel_size == stride (patch embedding, 1x1 heads): an unfold (pure data movement) followed by
    the MFMA GEMM.  Parameter names/shapes are nn.Conv2d's (reference: PatchEmbed.proj fourcastnet.py:310,
    swin_transformer.py:430; SwinTransformer.final :591)."""

    def forward(self, x, act=0):
        ph, pw = self.kernel_size
        assert tuple(self.stride) == (ph, pw) and self.padding == (0, 0), "only kernel_size == stride, padding 0"
        B, C_, H, W = x.shape
        h, w = H // ph, W // pw
        cols = x.reshape(B, C_, h, ph, w, pw).permute(0, 2, 4, 1, 3, 5).reshape(B * h * w, C_ * ph * pw)
        # the Parameter itself ([O, C, ph, pw]: the same memory as the [O, C*ph*pw] matrix) so that its gradient lands in .grad
        y = _LinearFn.apply(cols, self.weight, self.bias, act, None, False)
        return y.reshape(B, h, w, self.out_channels).permute(0, 3, 1, 2)

    def forward_tokens(self, x, act=0):
        """1 x 1 form on channels-last tokens [B, H, W, Cin] -> [B, H, W, O]: the GEMM alone (no unfold copy)."""
        assert tuple(self.kernel_size) == (1, 1) and tuple(self.stride) == (1, 1)
        return _LinearFn.apply(x, self.weight, self.bias, act, None, False)


class _UpConvTokensFn(torch.autograd.Function):
    """ConvTranspose2d(kernel == stride) + activation on channels-last tokens [B, H, W, Cin] -> [B, H kh, W kw, O]: one GEMM in
    the weight's own [Cin][O kh kw] layout (no transposed copy), then dlwp_upconv_shuffle (bias + activation + k x k interleave,
    one launch).  Backward: the adjoint shuffle (with the activation's derivative and the bias gradient), then gx = gy W^T and
    gW += x^T gy straight into the weight's gradient buffer."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        B, H, W_, Cin = x.shape
        O, kh, kw = weight.shape[1], weight.shape[2], weight.shape[3]
        N = O * kh * kw
        x2 = x.reshape(-1, Cin).contiguous().float()
        M = x2.shape[0]
        wsh = L.shadow(weight)
        wm = (wsh if wsh is not None else weight.contiguous()).reshape(Cin, N)
        y = torch.empty(M, N, device=x.device)
        _gemm(x2, wm, y, M, N, Cin, Cin, N, N, 0, 0)
        out = torch.empty(B, H * kh, W_ * kw, O, device=x.device)
        L.check(L.load().dlwp_upconv_shuffle(L.ptr(y), L.ptr(bias), None, L.ptr(out), None, B, H, W_, O, kh, kw, O, 0, act, 0,
                                             L.stream()))
        ctx.save_for_backward(x2, wm, y, bias)
        ctx.cfg = (B, H, W_, Cin, O, kh, kw, act)
        ctx.wslot, ctx.bslot, ctx.wshape = _grad_slot(weight), (_grad_slot(bias) if bias is not None else None), weight.shape
        return out

    @staticmethod
    def backward(ctx, gout):
        x2, wm, y, bias = ctx.saved_tensors
        B, H, W_, Cin, O, kh, kw, act = ctx.cfg
        M, N = x2.shape[0], O * kh * kw
        g = gout.contiguous().float()
        gy = torch.empty(M, N, device=g.device)
        gb = None
        if bias is not None:
            gb = ctx.bslot if ctx.bslot is not None else torch.zeros(O, device=g.device)
        L.check(L.load().dlwp_upconv_shuffle(L.ptr(y), L.ptr(bias), L.ptr(g), L.ptr(gy), L.ptr(gb), B, H, W_, O, kh, kw, O, 0, act, 1,
                                             L.stream()))
        gx = torch.empty(M, Cin, device=g.device)
        _gemm(gy, wm, gx, M, Cin, N, N, N, Cin, 0, 1)                       # gx = gy W^T
        if ctx.wslot is not None:
            _gemm(x2, gy, ctx.wslot, Cin, N, M, Cin, N, N, 1, 0, accumulate=1)  # gW += x^T gy, in the parameter's layout
            gw = None
        else:
            gw = torch.empty(Cin, N, device=g.device)
            _gemm(x2, gy, gw, Cin, N, M, Cin, N, N, 1, 0)
            gw = gw.reshape(ctx.wshape)
        return gx.reshape(B, H, W_, Cin), gw, (None if (bias is None or ctx.bslot is not None) else gb), None


class UpConvT2d(nn.ConvTranspose2d):
    """ConvTranspose2d with kernel_size == stride (Swin U-decoder, swin_transformer.py:580-588): GEMM over the
    input pixels followed by a pixel shuffle; an elementwise activation commutes with the shuffle and is fused
    into the GEMM epilogue."""

    def forward(self, x, act=0):
        kh, kw = self.kernel_size
        assert tuple(self.stride) == (kh, kw) and self.padding == (0, 0) and self.output_padding == (0, 0)
        B, C_, H, W = x.shape
        O = self.out_channels
        tokens = x.permute(0, 2, 3, 1).reshape(B * H * W, C_)
        wmat = self.weight.reshape(C_, O * kh * kw).t()                 # [O*kh*kw, Cin] as a Linear weight
        bias = self.bias.repeat_interleave(kh * kw) if self.bias is not None else None
        y = _LinearFn.apply(tokens, wmat, bias, act, None, False)
        return y.reshape(B, H, W, O, kh, kw).permute(0, 3, 1, 4, 2, 5).reshape(B, O, H * kh, W * kw)

    def forward_tokens(self, x, act=0):
        """Channels-last form: x [B, H, W, Cin] -> [B, H kh, W kw, O] without NCHW round trips (see _UpConvTokensFn)."""
        kh, kw = self.kernel_size
        assert tuple(self.stride) == (kh, kw) and self.padding == (0, 0) and self.output_padding == (0, 0)
        return _UpConvTokensFn.apply(x, self.weight, self.bias, act)

"""Train-step engine for the modules whose backward runs through autograd over libdlwpmi ops
(AFNONet, SwinTransformer, PanguWeather): the counterpart of the reference's inner training loop
(nsbench/scripts/train.py:113-131, dlwpbench/scripts/train.py:222-262) with the MI355X-specific parts:

* all parameters live in ONE flat fp32 buffer (and all gradients in another), so the optimizer is one
  dlwp_adam_step launch and data-parallel training needs one RCCL all-reduce (no bucketing logic);
* forward rollout + MSE + backward + optimizer are captured once into a hipGraph (torch.cuda.CUDAGraph is the
  plumbing that owns the capture stream and the private memory pool) and replayed per batch, which removes the
  per-op host dispatch that otherwise dominates these launch-bound models.

torch is used for memory, streams and graph capture; every arithmetic kernel in the captured step is either a
libdlwpmi kernel or a torch data-movement op (permute/roll/pad/cat copies, gradient accumulation adds).
"""
import torch

from . import lib as L
from .fno_engine import FusedAdam


# env: A/B runs of the transposed bf16 weight copies for the input-gradient products (round 6); matrices below the size keep the [k][n] form
_WEIGHT_T = __import__("os").environ.get("DLWP_WEIGHT_T", "1") != "0"
_WEIGHT_T_MIN = int(__import__("os").environ.get("DLWP_WEIGHT_T_MIN", "4096"))
_FLAT_ALIGN = int(__import__("os").environ.get("DLWP_FLAT_ALIGN", "8"))      # env: A/B runs against the 4-element slices of rounds 1-4


def flatten_parameters(module):
    """Re-point every parameter of `module` (and its .grad) at a slice of one flat buffer.
    Returns (flat_params, flat_grads).  Parameter names, shapes and values are unchanged."""
    params = [p for p in module.parameters() if p.requires_grad]
    if not params:
        raise ValueError("module has no trainable parameters")
    dev = params[0].device
    if dev.type != "cuda":
        raise L.DlwpError("flatten_parameters: move the module to the GPU first (no CPU path)")
    # 8-element aligned slices: 32 bytes in the fp32 buffers and 16 bytes in the bf16 copy (same offsets), the alignment the GEMMs'
    # 16-byte operand loads need -- with 4-element slices (rounds 1-4) a weight matrix that landed on an odd multiple of 4 had its
    # bf16 copy 8-byte aligned only and its products fell back to the widening operand path
    offs, n = [], 0
    for p in params:
        offs.append(n)
        n += (p.numel() + _FLAT_ALIGN - 1) // _FLAT_ALIGN * _FLAT_ALIGN
    flat = torch.zeros(n, device=dev)
    grad = torch.zeros(n, device=dev)
    for p, o in zip(params, offs):
        flat[o:o + p.numel()].copy_(p.data.reshape(-1))
        p.data = flat[o:o + p.numel()].view(p.shape)
        p.grad = grad[o:o + p.numel()].view(p.shape)
    skip = getattr(module, "dlwp_skip_weight_shadow", False)
    if L.storage_bf16() and not (skip() if callable(skip) else skip):
        # the bf16 copy the GEMMs read (lib.shadow): same offsets, refreshed from the fp32 master weights by one cast launch
        # at the top of every step (refresh_bf16_weights).  A module whose kernels all read their own per-step weight images
        # (packed from the fp32 master weights) opts out with `dlwp_skip_weight_shadow`.
        flat16 = torch.zeros(n, device=dev, dtype=torch.bfloat16)
        for p, o in zip(params, offs):
            p._dlwp_bf16 = flat16[o:o + p.numel()].view(p.shape)
        module._dlwp_flat16 = (flat, flat16)
        # transposed copies ([in][out]) of the 2-D weights for the input-gradient products (lib.shadow_t, dlwp_transpose_cast_bf16_many):
        # one more flat bf16 buffer, one more launch per step
        if _WEIGHT_T:
            descs, nt, max_tiles = [], 0, 0
            for p, o in zip(params, offs):
                if p.dim() == 2 and p.shape[0] % 8 == 0 and p.shape[1] % 8 == 0 and p.numel() >= _WEIGHT_T_MIN:
                    descs.append((o, nt, p.shape[0], p.shape[1], p))
                    nt += (p.numel() + _FLAT_ALIGN - 1) // _FLAT_ALIGN * _FLAT_ALIGN
                    max_tiles = max(max_tiles, ((p.shape[0] + 63) // 64) * ((p.shape[1] + 63) // 64))
            if descs:
                flat16t = torch.zeros(nt, device=dev, dtype=torch.bfloat16)
                for so, do, r, c, p in descs:
                    p._dlwp_bf16_t = flat16t[do:do + r * c].view(c, r)
                table = torch.tensor([[so, do, r, c] for so, do, r, c, _ in descs], dtype=torch.int64).to(dev)
                module._dlwp_flat16t = (flat, flat16t, table, len(descs), max_tiles)
    return flat, grad


def refresh_bf16_weights(module):
    pair = getattr(module, "_dlwp_flat16", None)
    if pair is not None:
        flat, flat16 = pair
        L.check(L.load().dlwp_cast_bf16(L.ptr(flat), L.ptr(flat16), flat.numel(), L.stream()))
    tr = getattr(module, "_dlwp_flat16t", None)
    if tr is not None:
        flat, flat16t, table, n, max_tiles = tr
        L.check(L.load().dlwp_transpose_cast_bf16_many(L.ptr(flat), L.ptr(flat16t), L.ptr(table), n, max_tiles, L.stream()))


class _SqErr(torch.autograd.Function):
    """mean((a - b)^2) with the gradient produced by the same kernel (dlwp_mse_fwd_bwd)."""

    @staticmethod
    def forward(ctx, pred, target):
        if pred.dtype != torch.float32 or target.dtype != torch.float32 or pred.shape != target.shape:
            raise L.DlwpError(f"mse_loss: fp32 tensors of one shape needed (pred {pred.dtype} {tuple(pred.shape)}, target {target.dtype} "
                              f"{tuple(target.shape)})")
        pred = pred.contiguous()
        target = target.contiguous()
        loss = torch.zeros(1, device=pred.device)
        g = torch.empty_like(pred)
        L.check(L.load().dlwp_mse_fwd_bwd(L.ptr(pred), L.ptr(target), pred.numel(), L.ptr(loss), L.ptr(g), L.stream()))
        ctx.save_for_backward(g)
        return loss[0]

    @staticmethod
    def backward(ctx, gl):
        (g,) = ctx.saved_tensors
        return g * gl, None


def mse_loss(pred, target):
    return _SqErr.apply(pred, target)


def mse_loss_and_grad(pred, target, loss_out=None):
    """(mean((pred - target)^2), its gradient with respect to pred) from the one dlwp_mse_fwd_bwd launch, for a caller that seeds
    the backward pass itself (torch.autograd.backward(pred, grad)): no ones_like seed, no seed x gradient product.  loss_out: a
    one-element fp32 tensor that receives the loss (zeroed here; the kernel accumulates into it)."""
    # the kernel reads raw float pointers: what loss.backward() used to validate is checked here
    if pred.dtype != torch.float32 or target.dtype != torch.float32:
        raise L.DlwpError(f"mse_loss_and_grad: fp32 tensors needed (pred {pred.dtype}, target {target.dtype})")
    if pred.shape != target.shape or pred.device != target.device:
        raise L.DlwpError(f"mse_loss_and_grad: pred {tuple(pred.shape)} on {pred.device} vs target {tuple(target.shape)} on {target.device}")
    p2 = pred.detach().contiguous()
    loss = torch.zeros(1, device=pred.device) if loss_out is None else loss_out.zero_()
    g = torch.empty_like(p2)
    L.check(L.load().dlwp_mse_fwd_bwd(L.ptr(p2), L.ptr(target.contiguous()), p2.numel(), L.ptr(loss), L.ptr(g), L.stream()))
    return loss.reshape(()), g


class GraphedTrainStep:
    """Captures `loss = mse(model(**inputs), target); loss.backward(); [all-reduce]; adam` and replays it.

    inputs/target given at construction fix the shapes; `__call__(inputs, target)` copies a new batch into the
    static buffers and replays.  With `allreduce` (ddp.FlatGradAllReduce) the capture is split into
    forward+backward | all-reduce (eager, RCCL) | optimizer so that the collective stays outside the graph.
    A reducer with `in_graph = True` (ddp.RcclComm: the C ABI's communicator enqueues ncclAllReduce on the caller's stream,
    which RCCL allows under stream capture) is captured INSIDE the one graph, between backward and the optimizer: a replay
    is then the whole data-parallel step.
    """

    def __init__(self, model, inputs, target, lr=1e-3, clip_max_norm=None, allreduce=None, grad_scale=1.0,
                 use_graph=True, call=None, graph_optimizer=True):
        self.model = model
        self.flat, self.grad = flatten_parameters(model)
        self.opt = FusedAdam(self.flat, self.grad, lr=lr)
        self.inputs = {k: v.clone() for k, v in inputs.items()}
        self.target = target.clone()
        if use_graph and hasattr(allreduce, "finish"):
            raise L.DlwpError("a bucketed reducer (ddp.BucketedGradAllReduce) launches collectives from backward hooks: build "
                              "the step with use_graph=False (the overlap replaces the capture for the large 1-step models)")
        self.clip, self.allreduce, self.grad_scale = clip_max_norm, allreduce, grad_scale
        self.call = call or (lambda m, kw: m(**kw))
        self.loss = torch.zeros((), device=self.flat.device)
        self.use_graph = use_graph
        self.collective_in_graph = False
        # graph_optimizer=False keeps clip + Adam outside the capture (two eager launches): the learning rate and the clip
        # threshold are launch arguments, so a schedule that changes them per epoch needs them re-read at every step
        self.graph_optimizer = graph_optimizer
        self.g_fb = self.g_opt = None
        if use_graph:
            self._capture()

    def _fwd_bwd(self):
        refresh_bf16_weights(self.model)          # no-op unless lib.set_storage("bf16") was active at construction
        prev, L.SHADOW_ACTIVE = L.SHADOW_ACTIVE, True
        try:
            out = self.call(self.model, self.inputs)
            _, g = mse_loss_and_grad(out, self.target, self.loss)
            torch.autograd.backward(out, g)           # = mse_loss(out, target).backward() without the seed launches
        finally:
            L.SHADOW_ACTIVE = prev

    def _optimize(self):
        # clip_grad_norm_ (dlwpbench train.py:133-135) is folded into the update: the norm pass, then Adam applies the coefficient
        self.opt.step(grad_scale=self.grad_scale, clip_max_norm=self.clip)

    def _capture(self):
        # warm-up on a side stream (allocator + lazy initialisation), then restore the untouched state
        flat0 = self.flat.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._fwd_bwd()
                self.grad.zero_()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.flat.copy_(flat0)
        self.g_fb = torch.cuda.CUDAGraph()
        self.collective_in_graph = bool(getattr(self.allreduce, "in_graph", False)) and self.graph_optimizer
        with torch.cuda.graph(self.g_fb):
            self._fwd_bwd()
            if self.collective_in_graph:
                self.allreduce(self.grad)
            if (self.allreduce is None or self.collective_in_graph) and self.graph_optimizer:
                self._optimize()
        if self.allreduce is not None and self.graph_optimizer and not self.collective_in_graph:
            self.g_opt = torch.cuda.CUDAGraph()
            pool = self.g_fb.pool()
            with torch.cuda.graph(self.g_opt, pool=pool):
                self._optimize()
        # the capture itself does not run the kernels; moments/step are still zero, gradients too
        self.grad.zero_()

    # ---- micro-batch form (dlwpbench/scripts/train.py:214-233: gradients of the micro-batches are SUMMED un-scaled, the
    # optional clip acts on the running sum after every micro-backward, one optimizer step per batch)
    def accumulate(self, inputs, target):
        """forward + loss + backward of one micro-batch; gradients add to what earlier micro-batches left."""
        for k, v in inputs.items():
            self.inputs[k].copy_(v)
        self.target.copy_(target)
        if hasattr(self.allreduce, "hold"):           # bucketed reducer: gradients are final only after the LAST micro-batch
            self.allreduce.hold = True                # (reset by its finish(), i.e. by apply())
        if self.use_graph:
            if self.graph_optimizer and self.allreduce is None:
                raise L.DlwpError("accumulate(): build the step with graph_optimizer=False (the optimizer must not be in the capture)")
            self.g_fb.replay()
        else:
            self._fwd_bwd()
        if self.clip is not None:
            self.opt.clip_grad_norm_(self.clip, grad_scale=1.0)
        return self.loss

    def apply(self):
        """all-reduce (if any) + Adam on the accumulated gradients."""
        if self.allreduce is not None:
            self.allreduce(self.grad)
        self.opt.step(grad_scale=self.grad_scale)

    def __call__(self, inputs=None, target=None):
        if inputs is not None:
            for k, v in inputs.items():
                self.inputs[k].copy_(v)
        if target is not None:
            self.target.copy_(target)
        if not self.use_graph:
            self._fwd_bwd()
            if self.allreduce is not None:
                self.allreduce(self.grad)
            self._optimize()
            return self.loss
        self.g_fb.replay()
        if getattr(self, "collective_in_graph", False):
            return self.loss
        if self.allreduce is not None:
            self.allreduce(self.grad)
        if self.g_opt is not None:
            self.g_opt.replay()
        elif self.allreduce is not None or not self.graph_optimizer:
            self._optimize()
        return self.loss

"""Drop-in counterparts of the reference's Navier-Stokes FNO rollout modules.

Reference (file:line under /root/reference/src):
  nsbench/models/fno/fno.py:10-41     FNOModule     (single-frame input, no context)
  nsbench/models/fno/fno.py:193-250   TFNO2DModule  (context frames flattened into channels;
                                      instantiates the dense neuralop FNO, :205)
Same constructor kwargs (extra kwargs such as `type`/`name` are swallowed), same
`forward(x[B,T,D,H,W], teacher_forcing_steps) -> [B,T,D,H,W]`, state_dict keys under `fno.`.

The rollout, its BPTT backward and the fused train step run entirely in libdlwpmi
(dlwp_fno_trainer_*); there is no PyTorch implementation of the arithmetic in this package.
"""
import torch
import torch.nn as nn

from ..fno_engine import FnoParamLayout, FnoRolloutTrainer, FusedAdam, make_cfg


class _RolloutFn(torch.autograd.Function):
    """autograd bridge: forward = trainer rollout (activations kept), backward = BPTT kernels.

    The shape-keyed trainer holds ONE set of BPTT activations; every forward stamps a generation number on it and backward
    refuses to run against another forward's activations (two outstanding forwards of one shape -- e.g.
    crit(m(x1)) + crit(m(x2)), or a validation forward between forward and backward -- would otherwise differentiate the
    wrong rollout without any error)."""

    @staticmethod
    def forward(ctx, flat, module, trainer, *derived):
        """derived: tensors the spectral weights inside `flat` were computed from (TFNO: the dense mode-major weights
        rebuilt from the Tucker factors); backward hands them the gradient of the corresponding flat-buffer entries."""
        ctx.module, ctx.trainer, ctx.n_derived = module, trainer, len(derived)
        trainer.generation += 1
        ctx.generation = trainer.generation
        trainer.forward(keep_activations=True)
        return trainer.out.clone()

    @staticmethod
    def backward(ctx, grad_out):
        module, trainer = ctx.module, ctx.trainer
        if ctx.generation != trainer.generation:
            raise RuntimeError(
                "FNO rollout: another forward of the same shape ran before this backward; its BPTT activations were "
                "overwritten (one outstanding forward per (B, T, H, W, teacher_forcing) shape is supported -- call "
                "backward() before the next forward, or run the second forward under torch.no_grad() AFTER backward)")
        slot = module._grad_slot()
        if slot is not None:
            # the kernels ACCUMULATE parameter gradients: straight into the parameter's own flat .grad buffer, which is what
            # the trainer is bound to (no temporary, no rebinding -> a captured step graph stays valid)
            gbuf, ret = slot, None
            module._bind(trainer)
            if ctx.n_derived:        # derived entries are not parameters: their gradient region is scratch
                for name in module._derived_names():
                    module.layout.view(gbuf, name).zero_()
        else:
            gbuf = ret = torch.zeros_like(module.flat_params.data)
            trainer.bind(module.flat_params.data, gbuf)     # rebound to the parameter's .grad on the next use (_bind)
        trainer.backward(grad_out.contiguous())
        gder = ()
        if ctx.n_derived:
            gder = tuple(module.layout.view(gbuf, name).clone() for name in module._derived_names())
            for name in module._derived_names():     # keep them out of the flat gradient's norm / Adam moments
                module.layout.view(gbuf, name).zero_()
        return (ret, None, None) + gder


class _FnoRolloutModule(nn.Module):
    def __init__(self, n_modes, in_channels, hidden_channels, lifting_channels, projection_channels,
                 out_channels, n_layers, context_size):
        super().__init__()
        if len(n_modes) not in (2, 3):
            raise ValueError("n_modes must have 2 entries (FNOModule / TFNO2DModule) or 3 (FNOContextModule)")
        self.n_modes = [int(m) for m in n_modes]
        self.in_channels, self.out_channels = int(in_channels), int(out_channels)
        self.hidden_channels, self.n_layers = int(hidden_channels), int(n_layers)
        self.lifting_channels, self.projection_channels = int(lifting_channels), int(projection_channels)
        self.context_size = int(context_size)
        # 3-D context form: the window is a volume, the lifting layer reads the D channels of it
        lift_in = self.in_channels if len(self.n_modes) == 3 else self.in_channels * max(1, self.context_size)
        self.layout = FnoParamLayout(lift_in, self.hidden_channels, self.lifting_channels, self.projection_channels,
                                     self.out_channels, self.n_layers, self.n_modes)
        flat = torch.empty(self.layout.total)
        self.layout.init_(flat)
        self.flat_params = nn.Parameter(flat)
        self.flat_grad = None
        self._trainers = {}

    # ---- reference-compatible checkpoints (keys `fno.*`)
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = self.layout.to_state_dict(self.flat_params.data, prefix=prefix + "fno.")
        if destination is not None:
            destination.update(sd)
            return destination
        return sd

    def load_state_dict(self, state_dict, strict=True, assign=False):
        self.layout.from_state_dict(self.flat_params.data, state_dict, prefix="fno.")
        return torch.nn.modules.module._IncompatibleKeys([], [])

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        for tr in self._trainers.values():
            tr.close()
        self._trainers = {}
        self.flat_grad = None
        return out

    def _ensure_grad(self):
        """The flat gradient buffer the trainers accumulate into.  An existing contiguous `.grad` of the flat parameter
        (e.g. the slice train_engine.flatten_parameters pointed it at) is ADOPTED, never replaced: whoever owns that buffer
        (GraphedTrainStep's optimizer, a data-parallel reducer) must see the BPTT gradients."""
        g = self.flat_params.grad
        if g is not None and g.is_contiguous() and g.shape == self.flat_params.shape and g.device == self.flat_params.device:
            self.flat_grad = g
        else:
            if (self.flat_grad is None or self.flat_grad.device != self.flat_params.device
                    or self.flat_grad.shape != self.flat_params.shape):
                self.flat_grad = torch.zeros_like(self.flat_params.data)
            self.flat_params.grad = self.flat_grad
        return self.flat_grad

    def _derived_names(self):
        return []

    def _grad_slot(self):
        g = self.flat_params.grad
        if g is not None and g.is_contiguous() and g.shape == self.flat_params.shape:
            self.flat_grad = g
            return g
        return None

    def _bind(self, tr):
        """(Re)bind a trainer to the CURRENT parameter / gradient storage.  `p.data = ...` (flatten_parameters, a
        checkpoint loader) moves the storage without going through _apply, so the pointers are compared on every use;
        rebinding drops the trainer's captured graph (dlwp_fno_trainer_bind)."""
        g = self._ensure_grad()
        if tr.params.data_ptr() != self.flat_params.data.data_ptr() or tr.grads.data_ptr() != g.data_ptr():
            tr.bind(self.flat_params.data, g)
        return tr

    def trainer(self, B, T, H, W, teacher_forcing_steps):
        key = (B, T, H, W, int(teacher_forcing_steps))
        tr = self._trainers.get(key)
        if tr is None:
            cfg = make_cfg(B, T, self.in_channels, H, W, self.context_size, int(teacher_forcing_steps),
                           self.hidden_channels, self.lifting_channels, self.projection_channels,
                           self.n_layers, self.n_modes, out_channels=self.out_channels)
            tr = FnoRolloutTrainer(cfg, self.flat_params.data, self._ensure_grad(), self.flat_params.device)
            self._trainers[key] = tr
        return self._bind(tr)

    def forward(self, x: torch.Tensor, teacher_forcing_steps: int = 10) -> torch.Tensor:
        B, T, D, H, W = x.shape
        tf = min(int(teacher_forcing_steps), T)
        tr = self.trainer(B, T, H, W, tf)
        tr.x.copy_(x)
        if torch.is_grad_enabled() and self.flat_params.requires_grad:
            return _RolloutFn.apply(self.flat_params, self, tr)
        tr.forward(keep_activations=False)
        return tr.out.clone()

    def io_buffers(self, B, T, H, W, teacher_forcing_steps):
        """The (x, y) device buffers the captured step of this shape reads: a loader that lands its host-to-device copy
        here (and passes them to train_step) saves the two device-to-device copies per step."""
        tr = self.trainer(B, T, H, W, min(int(teacher_forcing_steps), T))
        return tr.x, tr.y

    # ---- fused training step (forward + MSE + BPTT captured in one hipGraph, then Adam)
    def make_optimizer(self, lr=1e-3):
        return FusedAdam(self.flat_params.data, self._ensure_grad(), lr=lr)

    def train_step(self, x, y, teacher_forcing_steps, optimizer=None, use_graph=True, clip_max_norm=None,
                   grad_scale=1.0, allreduce=None):
        """nsbench/scripts/train.py:117-127 on device. Returns the (device) MSE loss tensor."""
        B, T, D, H, W = x.shape
        tr = self.trainer(B, T, H, W, min(int(teacher_forcing_steps), T))
        if x.data_ptr() != tr.x.data_ptr():      # batches staged straight into io_buffers() need no device copy
            tr.x.copy_(x)
        if y.data_ptr() != tr.y.data_ptr():
            tr.y.copy_(y)
        loss = tr.fwd_bwd(use_graph=use_graph)
        if allreduce is not None:
            allreduce(self.flat_grad)
        if optimizer is not None:
            if clip_max_norm is not None:
                optimizer.clip_grad_norm_(clip_max_norm, grad_scale=grad_scale)
            optimizer.step(grad_scale=grad_scale)
        return loss


class TFNO2DModule(_FnoRolloutModule):
    """nsbench/models/fno/fno.py:193-250 (dense FNO inside despite the name)."""

    def __init__(self, n_modes, in_channels, hidden_channels, lifting_channels, projection_channels, out_channels,
                 n_layers, max_n_modes=None, rank=1.0, bias=True, context_size=10, **kwargs):
        super().__init__(n_modes, in_channels, hidden_channels, lifting_channels, projection_channels,
                         out_channels, n_layers, context_size)


class FNOModule(_FnoRolloutModule):
    """nsbench/models/fno/fno.py:10-41: one frame in, one frame out (context_size 1 rollout)."""

    def __init__(self, n_modes, in_channels, hidden_channels, lifting_channels, projection_channels, out_channels,
                 n_layers, bias=True, **kwargs):
        super().__init__(n_modes, in_channels, hidden_channels, lifting_channels, projection_channels,
                         out_channels, n_layers, 1)

    def forward(self, x: torch.Tensor, teacher_forcing_steps: int = 50) -> torch.Tensor:
        return super().forward(x, teacher_forcing_steps)


class FNOContextModule(_FnoRolloutModule):
    """nsbench/models/fno/fno.py:44-100 (the shipped configs/model/fno.yaml): a 3-D (time, y, x) FNO over the context window
    [B, D, ctx, H, W] whose last time slice is the prediction; the context length IS n_modes[0] (:54 -- the YAML's
    `context_size` is swallowed by **kwargs).  On libdlwpmi the (time, y) transform pair is one separable
    dense DFT: the block kernels see a (ctx * H) x W image with n_modes[0] * n_modes[1] row frequencies
    (DLWP_FNO_FORM_NS_CONTEXT3D, include/dlwpmi.h)."""

    def __init__(self, n_modes, in_channels, hidden_channels, lifting_channels, projection_channels, out_channels, n_layers,
                 max_n_modes=None, bias=True, **kwargs):
        if len(n_modes) != 3:
            raise ValueError("FNOContextModule takes three n_modes (time, y, x)")
        super().__init__(n_modes, in_channels, hidden_channels, lifting_channels, projection_channels, out_channels, n_layers,
                         int(n_modes[0]))

    def forward(self, x: torch.Tensor, teacher_forcing_steps: int = 15) -> torch.Tensor:
        return super().forward(x, teacher_forcing_steps)

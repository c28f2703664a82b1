"""Drop-in counterpart of the reference's WeatherBench FNO rollout module.

Reference: src/dlwpbench/models/fno/fno.py:12-106 (FNO2DModule).  Same constructor kwargs and
`forward(constants[B,1,Cc,H,W]|None, prescribed[B,T,Cp,H,W]|None, prognostic[B,T,Cg,H,W])
-> [B,T-ctx,Cg,H,W]`; state_dict keys under `fno.`.

The reference's multi-step loop is broken as published (list.to(), fno.py:91-95, and out.cpu() at :104;
SURVEY.md App. B-1); this module implements the clean intent (UNet.forward, dlwpbench/models/unet/
unet.py:64-111): predictions stay on device and feed the following steps.
"""
import torch

from ..fno_engine import FnoRolloutTrainer, make_cfg
from ..nsbench.fno import _FnoRolloutModule, _RolloutFn


class FNO2DModule(_FnoRolloutModule):
    def __init__(self, n_modes=[12, 12], constant_channels=4, prescribed_channels=1, prognostic_channels=8,
                 hidden_channels=32, lifting_channels=256, projection_channels=256, n_layers=4, max_n_modes=None,
                 bias=True, context_size=10, **kwargs):
        self.constant_channels = int(constant_channels)
        self.prescribed_channels = int(prescribed_channels)
        self.prognostic_channels = int(prognostic_channels)
        in_total = self.constant_channels + (self.prescribed_channels + self.prognostic_channels) * int(context_size)
        # the base class sizes the lifting layer from in_channels * context_size: pass the total with context 1
        super().__init__(n_modes, in_total, hidden_channels, lifting_channels, projection_channels,
                         self.prognostic_channels, n_layers, 1)
        self.context_size = int(context_size)

    def trainer(self, B, T, H, W, teacher_forcing_steps=0):
        key = ("dlwp", B, T, H, W)
        tr = self._trainers.get(key)
        if tr is None:
            cfg = make_cfg(B, T, self.prognostic_channels, H, W, self.context_size, 0, self.hidden_channels,
                           self.lifting_channels, self.projection_channels, self.n_layers, self.n_modes,
                           out_channels=self.prognostic_channels, form=1,
                           constant_channels=self.constant_channels, prescribed_channels=self.prescribed_channels)
            tr = FnoRolloutTrainer(cfg, self.flat_params.data, self._ensure_grad(), self.flat_params.device)
            self._trainers[key] = tr
        return self._bind(tr)

    def _load(self, tr, constants, prescribed, prognostic):
        tr.x.copy_(prognostic)
        if tr.constants is not None:
            tr.constants.copy_(constants)
        if tr.prescribed is not None:
            tr.prescribed.copy_(prescribed)

    def forward(self, constants: torch.Tensor = None, prescribed: torch.Tensor = None,
                prognostic: torch.Tensor = None) -> torch.Tensor:
        B, T, _, H, W = prognostic.shape
        tr = self.trainer(B, T, H, W)
        self._load(tr, constants, prescribed, prognostic)
        if torch.is_grad_enabled() and self.flat_params.requires_grad:
            return _RolloutFn.apply(self.flat_params, self, tr)
        tr.forward(keep_activations=False)
        return tr.out.clone()

    def train_step(self, constants, prescribed, prognostic, target, optimizer=None, use_graph=True,
                   clip_max_norm=None, grad_scale=1.0, allreduce=None):
        """dlwpbench/scripts/train.py:126-139 (one micro-batch) on device; returns the MSE loss tensor."""
        B, T, _, H, W = prognostic.shape
        tr = self.trainer(B, T, H, W)
        self._load(tr, constants, prescribed, prognostic)
        tr.y.copy_(target)
        loss = tr.fwd_bwd(use_graph=use_graph)
        if allreduce is not None:
            allreduce(self.flat_grad)
        if optimizer is not None:
            if clip_max_norm is not None:
                optimizer.clip_grad_norm_(clip_max_norm, grad_scale=grad_scale)
            optimizer.step(grad_scale=grad_scale)
        return loss


class _CompositeAdam:
    """FusedAdam on the flat buffer + one FusedAdam per Tucker tensor (same hyper-parameters)."""

    def __init__(self, main, extra):
        self.main, self.extra = main, extra

    def clip_grad_norm_(self, max_norm, grad_scale=1.0):
        """torch.nn.utils.clip_grad_norm_ over the module's parameters = the flat buffer (whose derived dense spectral slices
        carry a zeroed gradient by now, TFNO2DModule.train_step) + every Tucker core / factor: ONE global norm
        (dlwpbench/scripts/train.py:230-232, `clip_gradients: True` in configs/training/default.yaml:3), one scale for all."""
        from .. import lib as L
        lib, ss = self.main.lib, self.main.sumsq
        ss.zero_()
        opts = [self.main] + list(self.extra)
        for o in opts:                               # dlwp_sumsq accumulates into the one-float buffer
            L.check(lib.dlwp_sumsq(L.ptr(o.grads), o.grads.numel(), L.ptr(ss), L.stream()))
        for o in opts:
            L.check(lib.dlwp_clip_scale(L.ptr(o.grads), o.grads.numel(), L.ptr(ss), grad_scale, max_norm, L.stream()))

    def step(self, grad_scale=1.0, zero_grad=True):
        self.main.step(grad_scale=grad_scale, zero_grad=zero_grad)
        for opt in self.extra:
            opt.step(grad_scale=grad_scale, zero_grad=zero_grad)


class TFNO2DModule(FNO2DModule):
    """dlwpbench/models/fno/fno.py:109-146: FNO2DModule whose spectral weights are Tucker-factorised
    (neuralop TFNO, `rank` = fraction of the dense parameter count; configs/model/fno.yaml:11).  The factors are
    the parameters; the dense mode-major weights inside the flat buffer are derived from them once per step by
    libdlwpmi's complex mode-product kernels and their gradient is pushed back through the same kernels."""

    def __init__(self, n_modes=[12, 12], constant_channels=4, prescribed_channels=1, prognostic_channels=8,
                 hidden_channels=32, lifting_channels=256, projection_channels=256, n_layers=4, max_n_modes=None,
                 bias=True, context_size=10, rank=0.5, **kwargs):
        super().__init__(n_modes, constant_channels, prescribed_channels, prognostic_channels, hidden_channels,
                         lifting_channels, projection_channels, n_layers, max_n_modes, bias, context_size)
        from ..tucker import TuckerSpectralWeight
        std = (2.0 / (2 * self.hidden_channels)) ** 0.5
        self.tucker = torch.nn.ModuleList([
            TuckerSpectralWeight(self.hidden_channels, self.hidden_channels, self.layout.m1, self.layout.m2c, rank, std)
            for _ in range(self.n_layers)])

    def _spec_names(self):
        return [f"fno_blocks.convs.weight.{l}" for l in range(self.n_layers)]

    def _refresh_dense(self):
        dense = [tw.dense_mode_major() for tw in self.tucker]
        with torch.no_grad():
            for name, d in zip(self._spec_names(), dense):
                self.layout.view(self.flat_params.data, name).copy_(d)
        return dense

    def _derived_names(self):
        return self._spec_names()

    def forward(self, constants=None, prescribed=None, prognostic=None):
        dense = self._refresh_dense()
        if torch.is_grad_enabled() and self.flat_params.requires_grad:
            # autograd path (train_engine.GraphedTrainStep, train_loop.train_dlwp): the BPTT kernels leave the gradient of
            # the dense weights in the flat gradient buffer; _RolloutFn hands it to the mode-product graph of the factors
            B, T, _, H, W = prognostic.shape
            tr = self.trainer(B, T, H, W)
            self._load(tr, constants, prescribed, prognostic)
            return _RolloutFn.apply(self.flat_params, self, tr, *dense)
        return super().forward(constants, prescribed, prognostic)

    def make_optimizer(self, lr=1e-3):
        from ..fno_engine import FusedAdam
        extra = []
        for p in self.tucker.parameters():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            extra.append(FusedAdam(p.data.view(-1), p.grad.view(-1), lr=lr))
        return _CompositeAdam(super().make_optimizer(lr), extra)

    def train_step(self, constants, prescribed, prognostic, target, optimizer=None, use_graph=True,
                   clip_max_norm=None, grad_scale=1.0, allreduce=None):
        dense = self._refresh_dense()
        loss = super().train_step(constants, prescribed, prognostic, target, optimizer=None, use_graph=use_graph)
        grads = [self.layout.view(self.flat_grad, name).clone() for name in self._spec_names()]
        torch.autograd.backward(dense, grads)
        # the dense spectral weights are derived tensors, not parameters: their gradient has been handed to the factors and
        # must neither count in the clipping norm nor move the (re-derived every step) dense copy
        for name in self._spec_names():
            self.layout.view(self.flat_grad, name).zero_()
        if allreduce is not None:
            allreduce(self.flat_grad)
            for p in self.tucker.parameters():
                allreduce(p.grad)
        if optimizer is not None:
            if clip_max_norm is not None:
                optimizer.clip_grad_norm_(clip_max_norm, grad_scale=grad_scale)
            optimizer.step(grad_scale=grad_scale)
        return loss

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = super().state_dict(prefix=prefix)
        for l in range(self.n_layers):
            sd.pop(f"{prefix}fno.fno_blocks.convs.weight.{l}.tensor", None)
            sd[f"{prefix}fno.fno_blocks.convs.weight.{l}.core"] = self.tucker[l].core.detach().clone()
            for k, f in enumerate(self.tucker[l].factors):
                sd[f"{prefix}fno.fno_blocks.convs.weight.{l}.factors.{k}"] = f.detach().clone()
        if destination is not None:
            destination.update(sd)
            return destination
        return sd

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
        return tr

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

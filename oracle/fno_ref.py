"""TEST INFRASTRUCTURE ONLY — CPU oracle for the FNO rollout path.  Never imported by the
product (dlwp_benchmark_amd/); only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may use it.

PARITY UNPINNED: the arithmetic of FNO/TFNO lives in the third-party package `neuralop`
(neuraloperator git 05c01c3, /root/reference/README.md:29-40), which is neither vendored in
/root/reference nor installed here.  `FNO`/`SpectralConv` below restate its published
algorithm with torch.fft; they are anchored on the reference's own call
sites and on the parameter-count identity of tests/test_oracle_fno.py (the widths of
src/nsbench/scripts/train_commands.txt:83-91 reproduce the 5k..32M budgets of
src/nsbench/scripts/plot_results.py:58).  The rollout drivers ARE in-tree and are restated
line by line:
    ns_rollout          <- src/nsbench/models/fno/fno.py:217-250 (TFNO2DModule.forward)
    ns_rollout_single   <- src/nsbench/models/fno/fno.py:29-41   (FNOModule.forward)
    train_step          <- src/nsbench/scripts/train.py:113-127  (closure: MSE, backward, Adam)
"""
import math

import torch
import torch.nn.functional as F


def _mode_slices(size, n_mode):
    """neuralop SpectralConv slicing of an fftshift-ed axis (App. A-1)."""
    start = size - min(size, n_mode)
    if start:
        return slice(start // 2, -start // 2)  # (-start)//2 floors: rows lo .. size-ceil(start/2)
    return slice(None)


def spectral_conv(x, weight, bias, n_modes):
    """x [B,Ci,H,W] real; weight complex [Ci,Co,m1,m2c] (m2c = n_modes[1]//2+1); bias [Co] or None."""
    B, Ci, H, W = x.shape
    m1, m2c = n_modes[0], n_modes[1] // 2 + 1
    X = torch.fft.rfftn(x, dim=(-2, -1), norm="forward")
    X = torch.fft.fftshift(X, dim=-2)
    out = torch.zeros(B, weight.shape[1], H, W // 2 + 1, dtype=X.dtype)
    rows = _mode_slices(H, m1)
    cols = slice(None, min(m2c, W // 2 + 1))
    out[:, :, rows, cols] = torch.einsum("bixy,ioxy->boxy", X[:, :, rows, cols], weight)
    out = torch.fft.fftshift(out, dim=-2)
    y = torch.fft.irfftn(out, s=(H, W), dim=(-2, -1), norm="forward")
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    return y


def pw_mlp(x, w1, b1, w2, b2):
    """neuralop MLP(n_layers=2): 1x1 conv -> GELU -> 1x1 conv on [B,C,H,W]."""
    h = F.gelu(F.conv2d(x, w1[:, :, None, None], b1))
    return F.conv2d(h, w2[:, :, None, None], b2)


def fno_block(x, wspec, wskip, bias, n_modes):
    """pre-activation of one FNO block: SpectralConv(x) + linear skip (bias-free 1x1 conv)."""
    return spectral_conv(x, wspec, bias, n_modes) + F.conv2d(x, wskip[:, :, None, None])


class FNO:
    """Dense 2-D FNO (neuralop.models.FNO as constructed at nsbench/models/fno/fno.py:205-215)."""

    def __init__(self, n_modes, in_channels, hidden_channels, lifting_channels, projection_channels,
                 out_channels, n_layers, seed=1234, dtype=torch.float32):
        self.n_modes = list(n_modes)
        self.n_layers = n_layers
        self.hidden = hidden_channels
        g = torch.Generator().manual_seed(seed)
        m1, m2c = n_modes[0], n_modes[1] // 2 + 1

        def conv_init(co, ci):
            bound = 1.0 / math.sqrt(ci)
            w = (torch.rand(co, ci, generator=g, dtype=torch.float64) * 2 - 1) * bound
            b = (torch.rand(co, generator=g, dtype=torch.float64) * 2 - 1) * bound
            return w.to(dtype), b.to(dtype)

        p = {}
        p["lifting.fcs.0.weight"], p["lifting.fcs.0.bias"] = conv_init(lifting_channels, in_channels)
        p["lifting.fcs.1.weight"], p["lifting.fcs.1.bias"] = conv_init(hidden_channels, lifting_channels)
        p["projection.fcs.0.weight"], p["projection.fcs.0.bias"] = conv_init(projection_channels, hidden_channels)
        p["projection.fcs.1.weight"], p["projection.fcs.1.bias"] = conv_init(out_channels, projection_channels)
        std = (2.0 / (hidden_channels + hidden_channels)) ** 0.5
        for l in range(n_layers):
            w = torch.randn(hidden_channels, hidden_channels, m1, m2c, 2, generator=g, dtype=torch.float64) * std
            p[f"fno_blocks.convs.weight.{l}"] = torch.view_as_complex(w.contiguous()).to(
                torch.complex64 if dtype == torch.float32 else torch.complex128)
            p[f"fno_blocks.fno_skips.{l}.weight"] = conv_init(hidden_channels, hidden_channels)[0]
            p[f"fno_blocks.convs.bias.{l}"] = (torch.randn(hidden_channels, generator=g, dtype=torch.float64) * std).to(dtype)
        self.params = p

    def parameters(self):
        return list(self.params.values())

    def requires_grad_(self, flag=True):
        for v in self.params.values():
            v.requires_grad_(flag)
        return self

    def n_params(self):
        return sum(v.numel() * (2 if v.is_complex() else 1) for v in self.params.values())

    def __call__(self, x):
        p = self.params
        h = pw_mlp(x, p["lifting.fcs.0.weight"], p["lifting.fcs.0.bias"],
                   p["lifting.fcs.1.weight"], p["lifting.fcs.1.bias"])
        for l in range(self.n_layers):
            h = fno_block(h, p[f"fno_blocks.convs.weight.{l}"], p[f"fno_blocks.fno_skips.{l}.weight"],
                          p[f"fno_blocks.convs.bias.{l}"], self.n_modes)
            if l < self.n_layers - 1:
                h = F.gelu(h)
        return pw_mlp(h, p["projection.fcs.0.weight"], p["projection.fcs.0.bias"],
                      p["projection.fcs.1.weight"], p["projection.fcs.1.bias"])


def ns_rollout(net, x, teacher_forcing_steps, context_size):
    """Line-by-line restatement of TFNO2DModule.forward (nsbench/models/fno/fno.py:217-250)."""
    outs = []
    out = None
    for t in range(x.shape[1]):
        if t < teacher_forcing_steps:
            x_t = x[:, max(0, t - (context_size - 1)):t + 1]
        else:
            if context_size == 0:
                x_t = out
            else:
                ts = max(0, (teacher_forcing_steps - t - 1) + context_size)
                x_obs = x[:, teacher_forcing_steps - ts:teacher_forcing_steps]
                x_out = torch.stack(outs[-(context_size - ts):], dim=1)
                x_t = torch.cat([x_obs, x_out], dim=1)
        if t < context_size - 1:
            out = x_t[:, -1]
        else:
            out = net(x_t.flatten(start_dim=1, end_dim=2))
        outs.append(out)
    return torch.stack(outs, dim=1)


def ns_rollout_single(net, x, teacher_forcing_steps):
    """FNOModule.forward (nsbench/models/fno/fno.py:29-41)."""
    outs = []
    x_t = None
    for t in range(x.shape[1]):
        x_t = x[:, t] if t < teacher_forcing_steps else x_t
        x_t = net(x_t)
        outs.append(x_t)
    return torch.stack(outs, dim=1)


def train_step(model, x, y, teacher_forcing_steps, context_size, optimizer=None):
    """One closure of nsbench/scripts/train.py:117-127 (clip disabled as in train_commands.txt:83)."""
    for p in model.parameters():
        p.grad = None
    y_hat = ns_rollout(model, x, teacher_forcing_steps, context_size)
    loss = F.mse_loss(y_hat, y)
    loss.backward()
    if optimizer is not None:
        optimizer.step()
    return loss.detach(), y_hat.detach()


# ---- layout converters between the oracle (neuralop-like) tensors and libdlwpmi's flat buffer
def spec_to_mode_major(w):
    """complex [Ci,Co,m1,m2c] -> real [m1,m2c,Ci,Co,2]"""
    return torch.view_as_real(w.permute(2, 3, 0, 1).contiguous()).contiguous()


def spec_from_mode_major(w):
    """real [m1,m2c,Ci,Co,2] -> complex [Ci,Co,m1,m2c]"""
    return torch.view_as_complex(w.contiguous()).permute(2, 3, 0, 1).contiguous()


def dlwp_prepare_inputs(constants, prescribed, prognostic):
    """FNO2DModule._prepare_inputs (dlwpbench/models/fno/fno.py:49-62)."""
    tensors = []
    if constants is not None:
        tensors.append(constants[:, 0])
    if prescribed is not None:
        tensors.append(prescribed.flatten(1, 2))
    if prognostic is not None:
        tensors.append(prognostic.flatten(1, 2))
    return torch.cat(tensors, dim=1)


def dlwp_rollout(net, constants, prescribed, prognostic, context_size):
    """FNO2DModule.forward (dlwpbench/models/fno/fno.py:64-106) in its clean, on-device form (the published
    loop calls .to() on a list, App. B-1; the intent is UNet.forward, dlwpbench/models/unet/unet.py:64-111)."""
    outs = []
    ctx = context_size
    for t in range(ctx, prognostic.shape[1]):
        t_start = max(0, t - ctx)
        if t == ctx:
            prognostic_t = prognostic[:, t_start:t]
        else:
            prognostic_t = torch.cat([prognostic[:, t_start:ctx], torch.stack(outs, dim=1)[:, -ctx:]], dim=1)
        x_t = dlwp_prepare_inputs(constants, prescribed[:, t - ctx:t] if prescribed is not None else None, prognostic_t)
        outs.append(prognostic_t[:, -1] + net(x_t))
    return torch.stack(outs, dim=1)


# ---- 3-D (time, y, x) FNO of nsbench FNOContextModule (src/nsbench/models/fno/fno.py:44-100: `FNO(n_modes=[12, 12, 12], ...)`
# applied to the context window as a [B, D, T, H, W] volume, output = last time slice).  Same third-party arithmetic as the
# 2-D case (neuralop SpectralConv is dimension-generic, App. A-1; PARITY UNPINNED), one more transformed axis.
def spectral_conv3d(x, weight, bias, n_modes):
    """x [B,Ci,T,H,W] real; weight complex [Ci,Co,m0,m1,m2c]; bias [Co] or None."""
    B, Ci, T, H, W = x.shape
    m0, m1, m2c = n_modes[0], n_modes[1], n_modes[2] // 2 + 1
    X = torch.fft.rfftn(x, dim=(-3, -2, -1), norm="forward")
    X = torch.fft.fftshift(X, dim=(-3, -2))
    out = torch.zeros(B, weight.shape[1], T, H, W // 2 + 1, dtype=X.dtype)
    s0, s1, s2 = _mode_slices(T, m0), _mode_slices(H, m1), slice(None, min(m2c, W // 2 + 1))
    out[:, :, s0, s1, s2] = torch.einsum("bixyz,ioxyz->boxyz", X[:, :, s0, s1, s2], weight)
    # fftshift on the way back too, as in the 2-D form above: the upstream SpectralConv of the pinned neuralop commit applies
    # torch.fft.fftshift in both directions (recalled, not verifiable here: neuralop is absent); on even sizes -- every shipped
    # and tested configuration -- fftshift and ifftshift are the same permutation
    out = torch.fft.fftshift(out, dim=(-3, -2))
    y = torch.fft.irfftn(out, s=(T, H, W), dim=(-3, -2, -1), norm="forward")
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1, 1)
    return y


class FNO3d:
    """Dense 3-D FNO (neuralop.models.FNO with three n_modes, as constructed at nsbench/models/fno/fno.py:56-65)."""

    def __init__(self, n_modes, in_channels, hidden_channels, lifting_channels, projection_channels, out_channels, n_layers,
                 seed=1234, dtype=torch.float32):
        self.n_modes, self.n_layers, self.hidden = list(n_modes), n_layers, hidden_channels
        g = torch.Generator().manual_seed(seed)
        m0, m1, m2c = n_modes[0], n_modes[1], n_modes[2] // 2 + 1

        def conv_init(co, ci):
            bound = 1.0 / math.sqrt(ci)
            w = (torch.rand(co, ci, generator=g, dtype=torch.float64) * 2 - 1) * bound
            b = (torch.rand(co, generator=g, dtype=torch.float64) * 2 - 1) * bound
            return w.to(dtype), b.to(dtype)

        p = {}
        p["lifting.fcs.0.weight"], p["lifting.fcs.0.bias"] = conv_init(lifting_channels, in_channels)
        p["lifting.fcs.1.weight"], p["lifting.fcs.1.bias"] = conv_init(hidden_channels, lifting_channels)
        p["projection.fcs.0.weight"], p["projection.fcs.0.bias"] = conv_init(projection_channels, hidden_channels)
        p["projection.fcs.1.weight"], p["projection.fcs.1.bias"] = conv_init(out_channels, projection_channels)
        std = (2.0 / (hidden_channels + hidden_channels)) ** 0.5
        for l in range(n_layers):
            w = torch.randn(hidden_channels, hidden_channels, m0, m1, m2c, 2, generator=g, dtype=torch.float64) * std
            p[f"fno_blocks.convs.weight.{l}"] = torch.view_as_complex(w.contiguous()).to(
                torch.complex64 if dtype == torch.float32 else torch.complex128)
            p[f"fno_blocks.fno_skips.{l}.weight"] = conv_init(hidden_channels, hidden_channels)[0]
            p[f"fno_blocks.convs.bias.{l}"] = (torch.randn(hidden_channels, generator=g, dtype=torch.float64) * std).to(dtype)
        self.params = p

    def parameters(self):
        return list(self.params.values())

    def requires_grad_(self, flag=True):
        for v in self.params.values():
            v.requires_grad_(flag)
        return self

    def __call__(self, x):
        """x [B, Cin, T, H, W] -> [B, Cout, T, H, W]"""
        p = self.params
        c3 = lambda t, w, b: F.conv3d(t, w[:, :, None, None, None], b)   # noqa: E731
        h = c3(F.gelu(c3(x, p["lifting.fcs.0.weight"], p["lifting.fcs.0.bias"])), p["lifting.fcs.1.weight"], p["lifting.fcs.1.bias"])
        for l in range(self.n_layers):
            h = spectral_conv3d(h, p[f"fno_blocks.convs.weight.{l}"], p[f"fno_blocks.convs.bias.{l}"], self.n_modes) + \
                c3(h, p[f"fno_blocks.fno_skips.{l}.weight"], None)
            if l < self.n_layers - 1:
                h = F.gelu(h)
        return c3(F.gelu(c3(h, p["projection.fcs.0.weight"], p["projection.fcs.0.bias"])), p["projection.fcs.1.weight"],
                  p["projection.fcs.1.bias"])


def ns_rollout_context(net, x, teacher_forcing_steps, context_size):
    """Line-by-line restatement of FNOContextModule.forward (nsbench/models/fno/fno.py:67-100): the window is transposed to
    [B, D, T, H, W], run through the 3-D net, and the LAST time slice is the prediction."""
    outs, out = [], None
    for t in range(x.shape[1]):
        if t < teacher_forcing_steps:
            x_t = x[:, max(0, t - (context_size - 1)):t + 1]
        else:
            if context_size == 0:
                x_t = out
            else:
                ts = max(0, (teacher_forcing_steps - t - 1) + context_size)
                x_obs = x[:, teacher_forcing_steps - ts:teacher_forcing_steps]
                x_out = torch.stack(outs[-(context_size - ts):], dim=1)
                x_t = torch.cat([x_obs, x_out], dim=1)
        x_t = torch.transpose(x_t, 1, 2)
        out = x_t[:, :, -1] if t < context_size - 1 else net(x_t)[:, :, -1]
        outs.append(out)
    return torch.stack(outs, dim=1)

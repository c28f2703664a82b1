"""The dlwpbench autoregressive rollout shared by every model family (R-dlwp).

Reference: the clean form of the loop, UNet.forward (src/dlwpbench/models/unet/unet.py:64-111) /
SFNONet.forward (models/fourcastnet/fourcastnet.py:475-527); the copies in AFNONet / SwinTransformer /
PanguWeather / FNO2DModule (fourcastnet.py:309-361, swin_transformer.py:694-737, panguweather.py:457-502,
fno.py:64-106) call `.to()` on a Python list and move every prediction to the host (`out.cpu()`), so they raise at
the second lead time.  Predictions stay on the device here.
"""
import torch

from ..rollout_ops import advance


def prepare_inputs(constants, prescribed, prognostic):
    """[B, Cc + T*Cp + T*Cg, H, W]: constants[:, 0], then prescribed and prognostic as "b (t c) h w"."""
    parts = [] if constants is None else [constants[:, 0]]
    if prescribed is not None:
        parts.append(prescribed.flatten(1, 2))
    if prognostic is not None:
        parts.append(prognostic.flatten(1, 2))
    return torch.cat(parts, dim=1)


def rollout(one_step, context_size, constants, prescribed, prognostic, patch=None):
    """out[:, t - ctx] = prog_t[:, -1] + one_step(x_t) for t in [ctx, T), prog_t = the last ctx frames (observed, then predicted).
    The window slides through rollout_ops.advance: one kernel per lead time instead of stack + cat + add (and one backward kernel
    instead of a gradient accumulation per reader of each predicted frame).  patch=(ph, pw): one_step returns the patch
    tokens of a linear head [B, h*w, ph*pw*C] instead of a frame (rollout_ops.advance un-patches them)."""
    outs, ctx, T = [], context_size, prognostic.shape[1]
    win, flat = prognostic[:, 0:ctx], None
    for t in range(ctx, T):
        parts = [] if constants is None else [constants[:, 0]]
        if prescribed is not None:
            parts.append(prescribed[:, t - ctx:t].flatten(1, 2))
        parts.append(flat if flat is not None else win.flatten(1, 2))
        x_t = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
        win, flat, out = advance(win, one_step(x_t), want_next=t + 1 < T, patch=patch)
        outs.append(out)
    return torch.stack(outs, dim=1)

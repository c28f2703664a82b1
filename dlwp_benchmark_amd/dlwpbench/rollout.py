"""The dlwpbench autoregressive rollout shared by every model family (R-dlwp).

Reference: the clean form of the loop, UNet.forward (src/dlwpbench/models/unet/unet.py:64-111) /
SFNONet.forward (models/fourcastnet/fourcastnet.py:475-527); the copies in AFNONet / SwinTransformer /
PanguWeather / FNO2DModule (fourcastnet.py:309-361, swin_transformer.py:694-737, panguweather.py:457-502,
fno.py:64-106) call `.to()` on a Python list and move every prediction to the host (`out.cpu()`), so they raise at
the second lead time.  Predictions stay on the device here.
"""
import torch


def prepare_inputs(constants, prescribed, prognostic):
    """[B, Cc + T*Cp + T*Cg, H, W]: constants[:, 0], then prescribed and prognostic as "b (t c) h w"."""
    parts = [] if constants is None else [constants[:, 0]]
    if prescribed is not None:
        parts.append(prescribed.flatten(1, 2))
    if prognostic is not None:
        parts.append(prognostic.flatten(1, 2))
    return torch.cat(parts, dim=1)


def rollout(one_step, context_size, constants, prescribed, prognostic):
    """out[:, t - ctx] = prog_t[:, -1] + one_step(x_t) for t in [ctx, T)."""
    outs, ctx = [], context_size
    for t in range(ctx, prognostic.shape[1]):
        if t == ctx:
            prog_t = prognostic[:, max(0, t - ctx):t]
        else:
            prog_t = torch.cat([prognostic[:, max(0, t - ctx):ctx], torch.stack(outs, dim=1)[:, -ctx:]], dim=1)
        x_t = prepare_inputs(constants, prescribed[:, t - ctx:t] if prescribed is not None else None, prog_t)
        outs.append(prog_t[:, -1] + one_step(x_t))
    return torch.stack(outs, dim=1)

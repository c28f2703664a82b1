"""The sliding input window of the autoregressive rollouts, advanced by one libdlwpmi kernel per lead time.

Reference loops (rebuilt from Python lists on each step -- stack + cat + residual add forward, and one gradient accumulation per
reader of every predicted frame backward):
  nsbench   AFNONet.forward  src/nsbench/models/fourcastnet/fourcastnet.py:262-300
            SwinTransformer.forward  src/nsbench/models/swintransformer/swin_transformer.py:597-640
  dlwpbench UNet.forward  src/dlwpbench/models/unet/unet.py:64-111 (the clean form; see dlwpbench/rollout.py)

`advance(win, delta)` returns (next window, the same window flattened over (time, channel) for the network, the new frame).  It is
one autograd node with three outputs, so the gradients arriving from the next advance, from the network and from the loss meet in
ONE backward kernel (dlwp_window_advance_bwd) instead of in autograd's accumulation adds.
"""
import torch

from . import lib as L


def _batch_view(t):
    """(tensor, batch stride in floats) such that a sample's block is contiguous; copies only if it is not."""
    if t is None:
        return None, 0
    if t.dtype != torch.float32:
        t = t.float()
    # the sample block itself must be dense whatever the batch size: at B = 1 a channels-last gradient (the patch embedding's input
    # gradient seen through cat's narrow) used to slip through a `B == 1` shortcut and was read as if it were [C, H, W] -- wrong
    # gradients for every multi-lead-time rollout at batch 1 (found by tests/test_gpu_round6.py, round 6)
    if t[0].is_contiguous():
        return t, (t.stride(0) if t.shape[0] > 1 else t[0].numel())
    t = t.contiguous()
    return t, t[0].numel()


def _p(t):
    """Device address of a tensor whose samples are contiguous blocks (the batch stride travels separately)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise L.DlwpError("libdlwpmi needs CUDA/HIP tensors (no CPU fallback)")
    return t.data_ptr()


class _AdvanceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, win, delta, want_next, patch):
        B, c = win.shape[0], win.shape[1]
        frame_shape = win.shape[2:]
        F = 1
        for s in frame_shape:
            F *= s
        w, wbs = _batch_view(win)
        d = delta.contiguous().float()
        assert d.numel() == B * F, (tuple(delta.shape), tuple(win.shape))
        nxt = torch.empty((B, c) + tuple(frame_shape), device=win.device) if want_next else None
        out = torch.empty((B,) + tuple(frame_shape), device=win.device)
        if patch is None:
            lay = (0, 0, 0, 0, 0, 0)
        else:                                           # delta holds patch tokens [B, h, w, ph, pw, D]
            D, H, W = frame_shape
            lay = (1, D, H, W, patch[0], patch[1])
        L.check(L.load().dlwp_window_advance_fwd(_p(w), wbs, L.ptr(d), L.ptr(nxt), L.ptr(out), B, c, F, *lay, L.stream()))
        ctx.cfg = (B, c, F, lay, tuple(delta.shape), tuple(win.shape))
        if not want_next:
            return None, None, out
        return nxt, nxt.view(B, c * frame_shape[0], *frame_shape[1:]), out

    @staticmethod
    def backward(ctx, g_next, g_flat, g_out):
        B, c, F, lay, dshape, wshape = ctx.cfg
        gn = g_next.contiguous().float() if g_next is not None else None
        gf, fbs = _batch_view(g_flat)
        go, obs = _batch_view(g_out)
        g_delta = torch.empty(dshape, device=(g_out if g_out is not None else (g_next if g_next is not None else g_flat)).device)
        g_win = torch.empty(wshape, device=g_delta.device) if ctx.needs_input_grad[0] else None
        L.check(L.load().dlwp_window_advance_bwd(L.ptr(gn), _p(gf), fbs, _p(go), obs, L.ptr(g_win), L.ptr(g_delta), B, c, F,
                                                 *lay, L.stream()))
        return g_win, g_delta, None, None


def advance(win, delta, want_next=True, patch=None):
    """win [B, ctx, D, H, W], delta [B, D, H, W] (or, with patch=(ph, pw), the head's patch tokens [B, H/ph * W/pw, ph*pw*D])
    -> (next window [B, ctx, D, H, W] | None, the same as [B, ctx*D, H, W] | None, out = win[:, -1] + delta)."""
    return _AdvanceFn.apply(win, delta, bool(want_next), patch)


def ns_rollout(one_step, x, teacher_forcing_steps, context_size, patch=None):
    """The nsbench rollout (fourcastnet.py:262-300 / swin_transformer.py:597-640): for every t the network sees the last
    `context_size` frames -- observed ones while t < teacher_forcing_steps, its own predictions afterwards -- and predicts a
    residual to the newest of them.  one_step(x_t [B, ctx*D, H, W]) -> delta.  Returns [B, T, D, H, W]."""
    T, ctx, tf = x.shape[1], context_size, teacher_forcing_steps
    if ctx < 1:
        raise ValueError("context_size must be >= 1 (the reference's context_size == 0 branch indexes a missing time axis)")
    outs, out, chain = [], None, None
    for t in range(T):
        if chain is not None:                    # closed loop on a full window: the previous advance built it
            win, flat = chain
        else:
            if t < tf:
                win = x[:, max(0, t - (ctx - 1)):t + 1]
            else:                                # first closed-loop step (or a window still shorter than ctx)
                ts = max(0, (tf - t - 1) + ctx)
                win = torch.cat([x[:, tf - ts:tf], torch.stack(outs[-(ctx - ts):], dim=1)], dim=1)
            flat = None
        chain = None
        if t < ctx - 1:
            out = win[:, -1]
        else:
            delta = one_step(flat if flat is not None else win.flatten(1, 2))
            want_next = t + 1 < T and t + 1 >= tf and win.shape[1] == ctx
            nxt, nflat, out = advance(win, delta, want_next, patch)
            if want_next:
                chain = (nxt, nflat)
        outs.append(out)
    return torch.stack(outs, dim=1)

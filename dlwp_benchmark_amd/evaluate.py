"""Evaluation rollout and metrics on libdlwpmi.

nsbench (scripts/evaluate.py:26-85, 232-257): forward rollout without gradients over the test batches, then RMSE overall /
in teacher forcing / in closed loop and the accumulated error "Frob".  The reference slices its xarray dataset by
*label* -- `sel(time=slice(0, tf))`, `sel(time=slice(tf, T))` on the coordinate 0..T-1 (:109) -- so the teacher-forcing
window is [0, tf] inclusive and the closed-loop window [tf, T-1]: step tf belongs to both.  Reproduced as is.

dlwpbench (scripts/evaluate.py:494-546): latitude-weighted RMSE per (lead time, variable), w = cos(lat) / mean(cos(lat))
(Rasp et al. 2020 eq. 2), and the anomaly correlation coefficient against a climatology (eq. A1).

All reductions run in one kernel (dlwp_error_moments); only the [5, G] moment table reaches the host.
"""
import math

import torch

from . import lib as L


def error_moments(outputs, targets, climatology=None, row_weights=None):
    """outputs / targets / climatology [B, G, H, W] on the GPU -> moments [5, G] (see dlwpmi.h)."""
    B, G, H, W = outputs.shape
    m = torch.zeros(5, G, device=outputs.device)
    # the contiguous copies stay bound to locals until the launch is enqueued: a temporary freed between two ptr() calls
    # could hand its block to the next .contiguous() and alias both arguments
    o, t = outputs.contiguous(), targets.contiguous()
    c = climatology.contiguous() if climatology is not None else None
    w = row_weights.contiguous() if row_weights is not None else None
    L.check(L.load().dlwp_error_moments(L.ptr(o), L.ptr(t), L.ptr(c), L.ptr(w), B, G, H, W, L.ptr(m), L.stream()))
    return m


def ns_metrics(outputs, targets, teacher_forcing_steps):
    """outputs / targets [B, T, D, H, W] -> dict(rmse, rmse_tf, rmse_cl, frob, frob_tf, frob_cl) as Python floats."""
    B, T, D, H, W = outputs.shape
    m = error_moments(outputs.reshape(B, T, D * H, W), targets.reshape(B, T, D * H, W)).double().cpu()
    n = B * D * H * W                                 # elements per time step
    tf = int(teacher_forcing_steps)
    win = {"": (0, T), "_tf": (0, min(tf + 1, T)), "_cl": (min(tf, T), T)}    # label-inclusive slices, see module docstring
    out = {}
    for tag, (a, b) in win.items():
        cnt = max(b - a, 0) * n
        out["rmse" + tag] = math.sqrt(m[0, a:b].sum().item() / cnt) if cnt else float("nan")
        out["frob" + tag] = (m[1, a:b] / n).sum().item()
    return out


@torch.no_grad()
def evaluate_ns(model, batches, teacher_forcing_steps):
    """evaluate_model (:26-85) without the NetCDF detour: rollout every (x, y) batch, accumulate the moments on device."""
    tot, shape = None, None
    for x, y in batches:
        y_hat = model(x, teacher_forcing_steps)
        B, T, D, H, W = y_hat.shape
        m = error_moments(y_hat.reshape(B, T, D * H, W), y.reshape(B, T, D * H, W))
        tot = m if tot is None else tot + m
        shape = (shape[0] + B, T, D, H, W) if shape else (B, T, D, H, W)
    B, T, D, H, W = shape
    n, tf, m = B * D * H * W, int(teacher_forcing_steps), tot.double().cpu()
    out = {}
    for tag, (a, b) in {"": (0, T), "_tf": (0, min(tf + 1, T)), "_cl": (min(tf, T), T)}.items():
        cnt = max(b - a, 0) * n
        out["rmse" + tag] = math.sqrt(m[0, a:b].sum().item() / cnt) if cnt else float("nan")
        out["frob" + tag] = (m[1, a:b] / n).sum().item()
    return out


def lat_weights(lats_deg, device=None):
    lats = torch.deg2rad(torch.as_tensor(lats_deg, dtype=torch.float64))
    w = torch.cos(lats) / torch.cos(lats).mean()
    return w.float().to(device) if device is not None else w.float()


def dlwp_metrics(outputs, targets, lats_deg, climatology=None):
    """outputs / targets (/ climatology) [B, T, V, H, W] -> rmse [T, V] (and acc [T, V]) tensors on the host."""
    B, T, V, H, W = outputs.shape
    w = lat_weights(lats_deg, outputs.device)
    clim = climatology.reshape(B, T * V, H, W) if climatology is not None else None
    m = error_moments(outputs.reshape(B, T * V, H, W), targets.reshape(B, T * V, H, W), clim, w).double().cpu()
    res = {"rmse": torch.sqrt(m[0] / (B * H * W)).reshape(T, V)}
    if climatology is not None:
        res["acc"] = (m[2] / torch.sqrt(m[3] * m[4])).reshape(T, V)
    return res

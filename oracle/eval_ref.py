"""TEST INFRASTRUCTURE ONLY — CPU oracle for the evaluation metrics.  Never imported by the product.

Restated from the reference's compute_metrics functions; xarray is absent from this image, so the label-based slicing
it relies on is spelled out with numpy indices (PARITY UNPINNED for that reading; the arithmetic itself is elementary):
    ns_metrics    <- src/nsbench/scripts/evaluate.py:232-257  (time coordinate = 0..T-1 at :109, so
                     sel(time=slice(0, tf)) is [0, tf] inclusive and sel(time=slice(tf, T)) is [tf, T-1])
    dlwp_metrics  <- src/dlwpbench/scripts/evaluate.py:513-546 (lat weights :516-518, RMSE :527-528, ACC :539-545)
"""
import numpy as np


def ns_metrics(outputs, targets, tf):
    diff = np.asarray(outputs, dtype=np.float64) - np.asarray(targets, dtype=np.float64)   # [B,T,D,H,W]
    T = diff.shape[1]
    sl = {"": slice(0, T), "_tf": slice(0, tf + 1), "_cl": slice(tf, T)}
    out = {}
    for tag, s in sl.items():
        d = diff[:, s]
        out["rmse" + tag] = float(np.sqrt((d ** 2).mean())) if d.size else float("nan")
        out["frob" + tag] = float(np.sqrt(d ** 2).mean(axis=(0, 2, 3, 4)).sum())
    return out


def dlwp_metrics(outputs, targets, lats_deg, climatology=None):
    o, t = np.asarray(outputs, dtype=np.float64), np.asarray(targets, dtype=np.float64)             # [B,T,V,H,W]
    lats = np.deg2rad(np.asarray(lats_deg, dtype=np.float64))
    w = (np.cos(lats) / np.mean(np.cos(lats)))[None, None, None, :, None]
    res = {"rmse": np.sqrt((w * (o - t) ** 2).mean(axis=(0, 3, 4)))}
    if climatology is not None:
        c = np.asarray(climatology, dtype=np.float64)
        oc, tc = o - c, t - c
        nom = (w * oc * tc).mean(axis=(0, 3, 4))
        den = np.sqrt((w * oc ** 2).mean(axis=(0, 3, 4)) * (w * tc ** 2).mean(axis=(0, 3, 4)))
        res["acc"] = nom / den
    return res

"""Short version of tools/bf16_training_quality.py (profiles/r06_bf16_training_quality.json): the benchmarked bf16 arithmetic (bf16
operands + bf16 storage) must TRAIN like the fp32 path on a learnable synthetic forecasting task -- same data stream, same initial
weights, closed-loop RMSE on held-out fields.  At this training length (250 steps, error ~ 15 % of the field's standard deviation)
the full experiment found the seed-paired ratio at 0.98 - 1.00 on average, but a SINGLE pair moves by +- 5 % from execution to execution
(float atomics reorder the weight-gradient sums and the loss is falling fast at this point of the training: 0.126 after 250 steps, 0.041
after 500), so the bar here is 10 % -- wide enough not to flake, narrow enough for a wrong gradient path (tens of per cent).  (At errors
below ~ 2 % the 8-bit mantissa of the bf16 matrix operands starts to show: that regime is the experiment's job, not this test's.)"""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("bf16_training_quality", os.path.join(ROOT, "tools", "bf16_training_quality.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_sfno_c3_widths_bf16_path_trains_like_the_fp32_path(cuda):
    q = _tool()
    try:
        res = {mode: q.run("sfno", mode, seed=0, steps=250, device=cuda) for mode in ("fp32", "bf16")}
    finally:
        q.set_mode("fp32")
    pers = res["fp32"]["persistence_rmse"]
    for mode, r in res.items():
        assert r["closed_loop_rmse"] < 0.3 * pers, (mode, r["closed_loop_rmse"], pers)          # the task is being learned
    ratio = res["bf16"]["closed_loop_rmse"] / res["fp32"]["closed_loop_rmse"]
    assert 0.90 <= ratio <= 1.10, (ratio, res["bf16"]["closed_loop_rmse"], res["fp32"]["closed_loop_rmse"])


def test_afno_fcn_widths_bf16_storage_trains_like_the_fp32_path(cuda):
    """FourCastNet widths (E = 768, 16 blocks) on the rFFT2 path, depth 6: bf16 operands + bf16 storage with the (default) fp32 spectrum
    window against the fp32 path, lead-1 RMSE after 150 single-step training steps."""
    q = _tool()
    try:
        res = {mode: q.run("afno", mode, seed=0, steps=150, device=cuda) for mode in ("fp32", "bf16_fp32spectra")}
    finally:
        q.set_mode("fp32")
    for mode, r in res.items():
        assert r["lead1_rmse"] < 0.75 * r["persistence_lead1_rmse"], (mode, r["lead1_rmse"], r["persistence_lead1_rmse"])
    ratio = res["bf16_fp32spectra"]["lead1_rmse"] / res["fp32"]["lead1_rmse"]
    assert 0.93 <= ratio <= 1.07, (ratio, res)          # (the full experiment: 1.003 after 187 steps, 1.007 after 375, over three seeds)


@pytest.mark.parametrize("family", ["swin", "pangu"])
def test_c4_families_bf16_path_trains_like_the_fp32_path(cuda, family):
    """The C4 families at their benchmarked widths on a 56 x 112 grid (profiles/r06_bf16_training_quality_c4.json: seed-paired ratios 0.97 -
    1.04 at 500 steps, single pairs 0.94 - 1.04): bf16 window-attention tensors, the LayerNorm backward's scaled bf16 second output under
    stochastic depth and the transposed weight copies are all on this path.  A wrong gradient path shows as tens of per cent."""
    q = _tool()
    try:
        res = {mode: q.run(family, mode, seed=0, steps=500, device=cuda) for mode in ("fp32", "bf16")}
    finally:
        q.set_mode("fp32")
    for mode, r in res.items():
        assert r["closed_loop_rmse"] < 0.6 * r["persistence_rmse"], (mode, r["closed_loop_rmse"], r["persistence_rmse"])
    ratio = res["bf16"]["closed_loop_rmse"] / res["fp32"]["closed_loop_rmse"]
    assert 0.88 <= ratio <= 1.12, (ratio, res["bf16"]["closed_loop_rmse"], res["fp32"]["closed_loop_rmse"])

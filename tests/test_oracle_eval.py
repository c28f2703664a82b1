"""Hand-checked cases for the metrics oracle (oracle/eval_ref.py)."""
import numpy as np

from oracle import eval_ref


def test_ns_metrics_windows_are_label_inclusive():
    # error = time index, one pixel: rmse over a window = sqrt(mean(t^2)), frob = sum |t|
    T = 6
    o = np.arange(T, dtype=np.float64).reshape(1, T, 1, 1, 1)
    m = eval_ref.ns_metrics(o, np.zeros_like(o), tf=2)
    assert abs(m["rmse"] - np.sqrt(np.mean(np.arange(6) ** 2))) < 1e-12
    assert abs(m["rmse_tf"] - np.sqrt(np.mean(np.array([0, 1, 2]) ** 2))) < 1e-12     # steps 0..2 inclusive
    assert abs(m["rmse_cl"] - np.sqrt(np.mean(np.array([2, 3, 4, 5]) ** 2))) < 1e-12  # step 2 is in both windows
    assert m["frob"] == 15 and m["frob_tf"] == 3 and m["frob_cl"] == 14


def test_dlwp_metrics_weighting_and_acc():
    lats = np.array([-60.0, 0.0, 60.0])
    w = np.cos(np.deg2rad(lats))
    w = w / w.mean()
    o = np.zeros((1, 1, 1, 3, 2))
    t = np.zeros_like(o)
    o[0, 0, 0, :, :] = np.array([[1, 1], [2, 2], [3, 3]])
    m = eval_ref.dlwp_metrics(o, t, lats, climatology=np.zeros_like(o) + 0.5)
    assert abs(m["rmse"][0, 0] - np.sqrt(np.mean(w * np.array([1, 4, 9])))) < 1e-12
    oc, tc = np.array([0.5, 1.5, 2.5]), np.array([-0.5, -0.5, -0.5])
    acc = np.mean(w * oc * tc) / np.sqrt(np.mean(w * oc ** 2) * np.mean(w * tc ** 2))
    assert abs(m["acc"][0, 0] - acc) < 1e-12

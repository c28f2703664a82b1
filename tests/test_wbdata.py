"""WeatherBench sample assembly (dlwp half of SURVEY.md §8 row D-shard) against the reference's index arithmetic, restated by
hand from src/dlwpbench/data/datasets/datasets.py:320-398 on fields whose values encode (variable, time)."""
import numpy as np
import torch

from dlwp_benchmark_amd import wbdata


def coded_fields(n_time, H=4, W=8):
    """value = 1000 * variable_id + time index, so every returned frame can be traced."""
    t = np.arange(n_time, dtype=np.float32)[:, None, None] * np.ones((1, H, W), dtype=np.float32)
    return {"t2m": 1000 + t, "u10": 2000 + t, "z": {500: 3000 + t, 700: 4000 + t}, "tisr": 5000 + t,
            "orography": np.full((H, W), 7.0, dtype=np.float32), "lsm": np.full((H, W), 8.0, dtype=np.float32)}


PROG = {"t2m": [], "z": [500, 700], "u10": []}


def test_len_and_windows():
    L, ctx = 5, 2
    ds = wbdata.WeatherBenchArrays(coded_fields(23), PROG, ["tisr"], ["orography", "lsm"], sequence_length=L, context_size=ctx)
    assert len(ds) == (23 - L) // L                                   # datasets.py:322-323
    constants, prescribed, prognostic, target = ds[2]
    t0 = 2 * L                                                        # item * sequence_length (:335)
    assert constants.shape == (1, 2, 4, 8) and constants[0, 0, 0, 0] == 7.0 and constants[0, 1, 0, 0] == 8.0
    assert prescribed.shape == (L, 1, 4, 8)
    assert np.array_equal(prescribed[:, 0, 0, 0], 5000 + np.arange(t0, t0 + L))
    # channel order = dict order, levels expanded in place (:368-384)
    assert prognostic.shape == (L, 4, 4, 8)
    assert np.array_equal(prognostic[0, :, 0, 0], [1000 + t0, 3000 + t0, 4000 + t0, 2000 + t0])
    assert np.array_equal(prognostic[:, 0, 0, 0], 1000 + np.arange(t0, t0 + L))
    # target = window[1:][ctx:] -> first target frame is t0 + 1 + ctx (the off-by-one of App. B-9, kept)
    assert target.shape == (L - ctx, 4, 4, 8)
    assert np.array_equal(target[:, 0, 0, 0], 1000 + np.arange(t0 + 1 + ctx, t0 + L + 1))
    assert prognostic.dtype == np.float32 and target.dtype == np.float32


def test_absent_inputs_are_nan_dummies_and_collate_to_none():
    ds = wbdata.WeatherBenchArrays(coded_fields(12), {"t2m": []}, sequence_length=3)
    c, p, g, t = ds[0]
    assert isinstance(c, float) and np.isnan(c) and isinstance(p, float) and np.isnan(p)     # :318, :366
    batch = wbdata.to_device_batch([ds[0], ds[1]], "cpu")
    assert batch[0] is None and batch[1] is None
    assert batch[2].shape == (2, 3, 1, 4, 8) and batch[3].shape == (2, 2, 1, 4, 8)


def test_normalisation_uses_per_level_statistics():
    stats = {"t2m": {"mean": 1000.0, "std": 2.0}, "z": {"level": {500: {"mean": 3000.0, "std": 4.0}, 700: {"mean": 4000.0, "std": 8.0}}},
             "u10": {"mean": 2000.0, "std": 1.0}, "tisr": {"mean": 5000.0, "std": 5.0}, "orography": {"mean": 7.0, "std": 1.0},
             "lsm": {"mean": 0.0, "std": 2.0}}
    ds = wbdata.WeatherBenchArrays(coded_fields(12), PROG, ["tisr"], ["orography", "lsm"], sequence_length=3, normalize=True,
                                   stats=stats)
    c, p, g, t = ds[1]
    assert c[0, 0, 0, 0] == 0.0 and c[0, 1, 0, 0] == 4.0
    assert np.allclose(p[:, 0, 0, 0], np.arange(3, 6) / 5.0)
    assert np.allclose(g[0, :, 0, 0], [3 / 2.0, 3 / 4.0, 3 / 8.0, 3.0])


def test_noise_is_added_to_inputs_only_and_seeded_per_item():
    ds = wbdata.WeatherBenchArrays(coded_fields(12), {"t2m": []}, sequence_length=3, noise=0.5, seed=7)
    clean = wbdata.WeatherBenchArrays(coded_fields(12), {"t2m": []}, sequence_length=3)
    a, b = ds[1], ds[1]
    assert np.array_equal(a[2], b[2])                                 # same item -> same noise whatever rank draws it
    assert not np.array_equal(a[2], ds[0][2] + 3)                     # different items -> different streams
    assert np.array_equal(a[3], clean[1][3])                          # targets are noise free (:392-393)
    d = a[2] - clean[1][2]
    assert 0.1 < d.std() < 1.5


def test_sharding_is_a_partition_independent_of_world_size():
    fields, prog, presc, const = wbdata.synthetic_fields(64, 8, 16)
    ds = wbdata.WeatherBenchArrays(fields, prog, presc, const, sequence_length=5, normalize=True)
    n = len(ds)
    one = wbdata.shard_batches(ds, epoch=3, rank=0, world=1, batch=1).ravel()
    two = np.concatenate([wbdata.shard_batches(ds, 3, r, 2, 1).ravel() for r in range(2)])
    assert sorted(one.tolist()) == list(range(n)) and sorted(two.tolist()) == sorted(one.tolist())[: len(two)] or set(two) <= set(one)
    c, p, g, t = ds[int(one[0])]
    assert c.shape == (1, 4, 8, 16) and p.shape == (5, 1, 8, 16) and g.shape == (5, 8, 8, 16) and t.shape == (4, 8, 8, 16)
    assert abs(float(g.mean())) < 1.0 and 0.3 < float(g.std()) < 3.0   # z-scored synthetic fields


def test_zero_fill_past_the_end_of_the_record():
    # a window that starts inside the record but is shorter than the sequence is padded with zeros (:386-389)
    ds = wbdata.WeatherBenchArrays(coded_fields(7), {"t2m": []}, sequence_length=5)
    ds.n_time = 100                                                   # pretend the index space is longer than the stored data
    c, p, g, t = ds[1]                                                # t0 = 5: only 2 frames exist
    assert g.shape[0] == 4 and np.array_equal(g[:, 0, 0, 0], [1005, 1006, 0, 0])


def test_full_sample_equals_hand_built_tensors():
    """Every element of a sample against tensors built by explicit loops over the reference's index arithmetic
    (datasets.py:335-395): item * L windows, dict-ordered variables with levels expanded in place, per-level z-scores,
    inputs = window[:-1], targets = window[1:][ctx:] (first target two steps after the last context frame, App. B-9)."""
    rng = np.random.default_rng(3)
    n_time, H, W, L, ctx = 31, 3, 5, 6, 2
    fields = {"t2m": rng.normal(280, 10, (n_time, H, W)).astype(np.float32),
              "z": {500: rng.normal(54000, 3000, (n_time, H, W)).astype(np.float32),
                    850: rng.normal(14000, 1000, (n_time, H, W)).astype(np.float32)},
              "tisr": rng.uniform(0, 4e6, (n_time, H, W)).astype(np.float32), "lsm": rng.uniform(0, 1, (H, W)).astype(np.float32)}
    prog = {"z": [850, 500], "t2m": []}
    stats = {"t2m": {"mean": 278.0, "std": 21.0}, "z": {"level": {500: {"mean": 54000.0, "std": 3300.0}, 850: {"mean": 13700.0, "std": 1470.0}}},
             "tisr": {"mean": 1074504.0, "std": 1439846.0}, "lsm": {"mean": 0.33, "std": 0.45}}
    ds = wbdata.WeatherBenchArrays(fields, prog, ["tisr"], ["lsm"], sequence_length=L, normalize=True, context_size=ctx, stats=stats)
    for item in range(len(ds)):
        t0 = item * L
        exp_presc = np.zeros((L, 1, H, W), np.float32)
        for t in range(L):
            exp_presc[t, 0] = (fields["tisr"][t0 + t] - stats["tisr"]["mean"]) / stats["tisr"]["std"]
        window = np.zeros((L + 1, 3, H, W), np.float32)
        for t in range(L + 1):
            window[t, 0] = (fields["z"][850][t0 + t] - stats["z"]["level"][850]["mean"]) / stats["z"]["level"][850]["std"]
            window[t, 1] = (fields["z"][500][t0 + t] - stats["z"]["level"][500]["mean"]) / stats["z"]["level"][500]["std"]
            window[t, 2] = (fields["t2m"][t0 + t] - stats["t2m"]["mean"]) / stats["t2m"]["std"]
        c, p, g, t_ = ds[item]
        assert np.allclose(c, ((fields["lsm"] - 0.33) / 0.45)[None, None], atol=1e-6)
        assert np.allclose(p, exp_presc, atol=1e-6)
        assert np.allclose(g, window[:-1], atol=1e-6)
        assert np.allclose(t_, window[1:][ctx:], atol=1e-6) and t_.shape[0] == L - ctx


# ---- the init_dates branch (evaluation): label-inclusive slices, 2017 fill of the prescribed variable, zero fill ---------------
def _dated_fields(start, n_time, dt_h, H=2, W=3):
    """coded fields on a real time axis: value = 1000 * variable + index along the time axis."""
    import pandas as pd
    times = pd.date_range(start=start, periods=n_time, freq=f"{dt_h}h")
    return coded_fields(n_time, H, W), times


def _restate_with_pandas(fields, times, init_date, L, dt_h, ctx, prog):
    """What xarray does in datasets.py:339-395, with pandas label slicing (inclusive on both ends, like xarray's)."""
    import pandas as pd
    idx = pd.Series(np.arange(len(times)), index=times)
    d0 = pd.Timestamp(init_date)
    sel = idx.loc[d0:d0 + pd.Timedelta(f"{L * dt_h}h")].to_numpy()
    presc = fields["tisr"][sel]
    if L > len(sel):
        diff = L - len(sel)
        dates = pd.date_range(start=d0, end=d0 + pd.Timedelta(f"{L * dt_h}h"), freq=f"{dt_h}h")
        tmp = []
        for date in dates[-diff:]:
            date = date.replace(year=2017, day=28) if date.month == 2 and date.day > 28 else date.replace(year=2017)
            tmp.append(fields["tisr"][idx.loc[date]])
        presc = np.concatenate((presc, np.array(tmp)))
    sel2 = idx.loc[d0:d0 + pd.Timedelta(f"{(L + 1) * dt_h}h")].to_numpy()
    chans = []
    for p, levels in prog.items():
        if levels:
            chans += [fields[p][l][sel2] for l in levels]
        else:
            chans.append(fields[p][sel2])
    prognostic = np.float32(np.stack(chans, axis=1))
    if len(prognostic) < L:
        prognostic = np.concatenate((prognostic, np.zeros((L - len(prognostic), *prognostic.shape[1:]), dtype=np.float32)), axis=0)
    return np.float32(presc)[:, None], prognostic[:-1], prognostic[1:][ctx:]


def test_init_dates_branch_matches_label_inclusive_slicing():
    L, dt_h, ctx = 6, 6, 1
    fields, times = _dated_fields("2017-01-01", 4 * 365 * 2, dt_h)          # 2017-01-01 ... 2018-12-31, 6-hourly
    inits = wbdata.make_biweekly_inits("2017-01-01", "2018-12-31", sequence_length=L, timedelta=dt_h)
    ds = wbdata.WeatherBenchArrays(fields, PROG, ["tisr"], ["orography"], sequence_length=L, context_size=ctx, times=times.to_numpy(),
                                   init_dates=inits, timedelta=dt_h)
    assert len(ds) == len(inits)
    for item in (0, 1, len(inits) // 2, len(inits) - 1):
        c, p, g, t = ds[item]
        p_ref, g_ref, t_ref = _restate_with_pandas(fields, times, inits[item], L, dt_h, ctx, PROG)
        assert p.shape == (L + 1, 1, 2, 3) and g.shape == (L + 1, 4, 2, 3) and t.shape == (L + 1 - ctx, 4, 2, 3)      # inclusive slices: one frame more
        assert np.array_equal(p, p_ref) and np.array_equal(g, g_ref) and np.array_equal(t, t_ref)


def test_make_biweekly_inits_equals_the_pandas_construction():
    import pandas as pd
    for start, end, L, dt_h in (("2017-01-01", "2018-12-31", 57, 6), ("2017-01-01", "2017-03-01", 15, 6), ("2016-02-25", "2016-06-30", 20, 12)):
        t1 = pd.date_range(start=start, end=pd.Timestamp(end) - pd.Timedelta(hours=L * dt_h), freq="7D")
        t2 = pd.date_range(start=pd.Timestamp(start) + pd.Timedelta(days=3), end=pd.Timestamp(end) - pd.Timedelta(hours=L * dt_h), freq="7D")
        want = t1.append(t2).sort_values().to_numpy()
        got = wbdata.make_biweekly_inits(start, end, L, dt_h)
        assert np.array_equal(got.astype("datetime64[ns]"), want.astype("datetime64[ns]"))


def test_init_dates_past_the_record_fills_prescribed_from_2017_and_prognostic_with_zeros():
    """An initialisation date three frames before the record ends (2018-12-31 18:00): the prescribed variable continues with the same
    calendar dates of 2017 (datasets.py:348-360), the prognostic window is zero-filled up to sequence_length frames (:386-389)."""
    L, dt_h, ctx = 8, 6, 1
    fields, times = _dated_fields("2017-01-01", 4 * 365 * 2, dt_h)
    init = np.array([times[-3].to_numpy(), np.datetime64("2018-02-27T00")])
    ds = wbdata.WeatherBenchArrays(fields, PROG, ["tisr"], None, sequence_length=L, context_size=ctx, times=times.to_numpy(), init_dates=init,
                                   timedelta=dt_h)
    for item in range(2):
        c, p, g, t = ds[item]
        p_ref, g_ref, t_ref = _restate_with_pandas(fields, times, init[item], L, dt_h, ctx, PROG)
        assert np.array_equal(p, p_ref) and np.array_equal(g, g_ref) and np.array_equal(t, t_ref)
    c, p, g, t = ds[0]
    assert p.shape[0] == L                                            # 3 frames of the record + 5 from 2017
    last = 4 * 365 * 2 - 1
    assert p[0, 0, 0, 0] == 5000 + last - 2 and p[2, 0, 0, 0] == 5000 + last
    assert p[3, 0, 0, 0] == 5000 + 1                                  # the LAST five dates of the inclusive range d0 .. d0 + 8 dt start at 2019-01-01 06:00 -> 2017-01-01 06:00 = index 1
                                                                      # (the reference skips 2019-01-01 00:00 that way: reproduced, not fixed)
    assert g.shape[0] == L - 1 and np.all(g[3:] == 0)                 # 3 real frames + zero fill to L, minus the shifted last one


def test_netcdf3_files_of_the_weatherbench_layout_load_into_the_sample_assembly(tmp_path):
    """Classic NetCDF files written with scipy (the one NetCDF flavour this image can read or write): two yearly files of a surface
    variable, one level-resolved file, a constants file; time in "hours since 1900-01-01" as WeatherBench stores it."""
    import pandas as pd
    from scipy.io import netcdf_file
    H, W = 3, 4

    def write(path, name, times, data, levels=None):
        with netcdf_file(str(path), "w") as f:
            f.createDimension("time", len(times)); f.createDimension("lat", H); f.createDimension("lon", W)
            tv = f.createVariable("time", "i4", ("time",))
            tv[:] = ((times - pd.Timestamp("1900-01-01")) / pd.Timedelta("1h")).astype(np.int32)
            tv.units = "hours since 1900-01-01 00:00:00.0"
            if levels is None:
                f.createVariable(name, "f4", ("time", "lat", "lon"))[:] = data
            else:
                f.createDimension("level", len(levels))
                f.createVariable("level", "i4", ("level",))[:] = np.array(levels, dtype=np.int32)
                f.createVariable(name, "f4", ("time", "level", "lat", "lon"))[:] = data
    t16 = pd.date_range("2016-12-30", "2016-12-31 18:00", freq="6h")
    t17 = pd.date_range("2017-01-01", "2017-01-03 18:00", freq="6h")
    ones = np.ones((1, H, W), dtype=np.float32)
    write(tmp_path / "t2m_2016.nc", "t2m", t16, (100 + np.arange(len(t16), dtype=np.float32))[:, None, None] * ones)
    write(tmp_path / "t2m_2017.nc", "t2m", t17, (200 + np.arange(len(t17), dtype=np.float32))[:, None, None] * ones)
    zz = np.stack([(300 + np.arange(len(t16) + len(t17), dtype=np.float32))[:, None, None] * ones,
                   (400 + np.arange(len(t16) + len(t17), dtype=np.float32))[:, None, None] * ones], axis=1)
    write(tmp_path / "z_all.nc", "z", t16.append(t17), zz, levels=[500, 700])
    with netcdf_file(str(tmp_path / "constants.nc"), "w") as f:
        f.createDimension("lat", H); f.createDimension("lon", W)
        f.createVariable("orography", "f4", ("lat", "lon"))[:] = np.full((H, W), 7.0, dtype=np.float32)
    prog = {"t2m": [], "z": [500, 700]}
    paths = [str(p) for p in tmp_path.glob("*.nc")]
    fields, times = wbdata.load_netcdf3_fields(paths, prog, [], ["orography"], start_date="2016-12-31", stop_date="2017-01-02T18", timedelta=2)
    want_t = pd.date_range("2016-12-31", "2017-01-02 18:00", freq="12h").to_numpy().astype("datetime64[h]")
    assert np.array_equal(times, want_t)                                  # label-inclusive, every second frame
    assert np.array_equal(fields["t2m"][:, 0, 0], [104, 106, 200, 202, 204, 206])
    assert np.array_equal(fields["z"][700][:, 0, 0], 400 + np.arange(4, 16, 2))
    ds = wbdata.WeatherBenchArrays(fields, prog, None, ["orography"], sequence_length=2, context_size=1, times=times)
    c, p, g, t = ds[0]
    assert c.shape == (1, 1, H, W) and g.shape == (2, 3, H, W) and np.array_equal(g[:, 0, 0, 0], [104, 106])
    # an HDF5-based NetCDF-4 file is refused with a message that says why
    (tmp_path / "hdf5.nc").write_bytes(b"\\x89HDF\\r\\n\\x1a\\n" + b"\\0" * 64)
    try:
        wbdata.load_netcdf3_fields([str(tmp_path / "hdf5.nc")], prog)
        assert False, "an HDF5 file must be refused"
    except OSError as e:
        assert "NetCDF-3" in str(e)

"""WeatherBench sample assembly for the dlwpbench rollout step (SURVEY.md §8 row D-shard, dlwp half).

Reference: `WeatherBenchDataset.__len__` / `__getitem__` (src/dlwpbench/data/datasets/datasets.py:320-398) and the z-score
table `WeatherBenchDataset.STATISTICS` (:19-235).  The reference reads yearly NetCDF / zarr files through xarray, which
is not in this image (and file I/O is out of scope): this twin takes the already decoded fields as in-memory
arrays, one per variable, and reproduces what the training step sees -- the index arithmetic, the normalisation, the
noise, the dummy NaN for absent inputs and the target shift -- so that `train.py`'s batch tuple
`(constants, prescribed, prognostic, target)` (src/dlwpbench/scripts/train.py:116-121) has the reference's contents.

Reproduced as is: `target = prognostic[1:]` and the method returns `target[context_size:]`, while
the model's first output belongs to input frame `context_size`; the first prediction is therefore scored against the
field two steps after the last input frame.

Both branches of `__getitem__` are built: the randomly-sampled-initialisation branch (`init_dates is None`, the training path,
:327, :335, :368) and -- round 6 -- the `init_dates` branch of the evaluation (:339-360, :373-377; bi-weekly initialisation dates from
`make_biweekly_inits`, scripts/evaluate.py:56-68): xarray's label slices are INCLUSIVE on both ends, so `sel(time=slice(d, d + L dt))`
yields L + 1 frames of the prescribed variables and `slice(d, d + (L + 1) dt)` L + 2 of the prognostic ones while the record lasts;
prescribed frames past the end of the record are taken from the same calendar date of 2017 (29 February -> 28 February), prognostic
frames past the end are NOT padded unless fewer than L exist (zero fill, :386-389).  The calendar arithmetic is numpy datetime64
(the reference uses pandas; tests/test_wbdata.py checks this twin against pandas label slicing).
"""
import math

import numpy as np
import torch

from . import ddp

# mean / std of the variables of configs/data/weatherbench.yaml:23-47 (values: datasets.py:19-235; data, not code).
# Level-resolved variables carry {"level": {hPa: {...}}}; pass your own table for other variables.
STATISTICS = {
    "t": {"level": {850: {"mean": 274.518798828125, "std": 15.591468811035156}}},
    "t2m": {"mean": 278.44608, "std": 21.24761},
    "u10": {"mean": -0.09109575, "std": 5.547917},
    "v10": {"mean": 0.2246149, "std": 4.7760262},
    "z": {"level": {300: {"mean": 89399.90625, "std": 5087.197265625}, 500: {"mean": 54107.8671875, "std": 3349.03125},
                    700: {"mean": 28924.94921875, "std": 2132.567626953125},
                    1000: {"mean": 738.5037841796875, "std": 1069.619140625}}},
    "tisr": {"mean": 1074504.8, "std": 1439846.4},
    "orography": {"mean": 379.4976, "std": 859.87225},
    "lsm": {"mean": 0, "std": 1},
    "slt": {"mean": 0, "std": 1},
    "lat2d": {"mean": 0, "std": 51.936146191742026},
    "lon2d": {"mean": 177.1875, "std": 103.9103617607503},
}


class WeatherBenchArrays(torch.utils.data.Dataset):
    """`WeatherBenchDataset` over decoded arrays.

    fields:  {name: array [time, lat, lon]} for surface / prescribed variables, {name: {level: array [time, lat, lon]}}
             for level-resolved ones (what `self.ds[p]` / `.sel(level=l)` yield, datasets.py:371-380), and
             {name: array [lat, lon]} for constants; all on the `timedelta`-subsampled time axis (:287).
    The other arguments are the reference constructor's (:236-255)."""

    def __init__(self, fields, prognostic_variable_names_and_levels, prescribed_variable_names=None, constant_names=None,
                 sequence_length=15, noise=0.0, normalize=False, context_size=1, stats=None, seed=None, times=None, init_dates=None,
                 timedelta=6, **kwargs):
        """times: np.datetime64 array, the time coordinate of every [time, ...] field (needed with init_dates); init_dates: array of
        np.datetime64 initialisation dates (the evaluation branch, :257, :320-325); timedelta: hours between two frames (:246)."""
        self.fields = fields
        self.times = None if times is None else np.asarray(times).astype("datetime64[h]")
        self.init_dates = None if init_dates is None else np.asarray(init_dates).astype("datetime64[h]")
        self.timedelta = int(timedelta)
        if self.init_dates is not None and self.times is None:
            raise ValueError("init_dates needs the time coordinate of the fields (times=...)")
        self.prognostic_variable_names_and_levels = dict(prognostic_variable_names_and_levels)
        self.prescribed_variable_names = list(prescribed_variable_names or [])
        self.constant_names = list(constant_names or [])
        self.sequence_length, self.noise, self.normalize = int(sequence_length), float(noise), bool(normalize)
        self.context_size = int(context_size)
        self.stats = STATISTICS if stats is None else stats
        self.seed = seed           # None: numpy's global generator like the reference (:393); an int: per-item streams
        self.epoch = 0             # set_epoch(): part of the noise seed
        first = next(iter(self.prognostic_variable_names_and_levels))
        lv = self.prognostic_variable_names_and_levels[first]
        self.n_time = len(self.fields[first][lv[0]] if lv else self.fields[first])
        if self.constant_names:                                            # [1, #constants, lat, lon]  (:308-316)
            cs = [self._norm(np.asarray(self.fields[c]), self.stats[c]) if self.normalize else np.asarray(self.fields[c])
                  for c in self.constant_names]
            self.constants = np.expand_dims(np.float32(np.stack(cs)), axis=0)
        else:
            self.constants = torch.nan                                     # dummy when unused (:318)

    @staticmethod
    def _norm(a, st):
        return (a - st["mean"]) / st["std"]

    def set_epoch(self, epoch):
        """Seeded mode only: makes the noise stream of every item epoch-dependent (like ddp.ns_sample)."""
        self.epoch = int(epoch)

    def __len__(self):
        if self.init_dates is not None:
            return len(self.init_dates)                                                 # :324-325
        return (self.n_time - self.sequence_length) // self.sequence_length            # :322-323

    def _label_slice(self, start, stop):
        """Index range of `sel(time=slice(start, stop))`: xarray label slices include BOTH end points."""
        return int(np.searchsorted(self.times, start, side="left")), int(np.searchsorted(self.times, stop, side="right"))

    def _getitem_init_date(self, item):
        L_, dt = self.sequence_length, np.timedelta64(self.timedelta, "h")
        d0 = self.init_dates[item]
        if self.prescribed_variable_names:
            ps = []
            for p in self.prescribed_variable_names:
                a, b = self._label_slice(d0, d0 + L_ * dt)                              # :343-347 (inclusive: L + 1 frames while the record lasts)
                data = np.asarray(self.fields[p][a:b])
                if L_ > len(data):                                                      # :348-360: the record ends -- same calendar dates of 2017
                    diff = L_ - len(data)
                    dates = d0 + dt * np.arange(int((L_ * dt) / dt) + 1)                # pd.date_range(start, stop, freq = dt), inclusive
                    extra = []
                    for date in dates[-diff:]:
                        day = date.astype("datetime64[D]")
                        y, m, dd = str(day).split("-")
                        if m == "02" and int(dd) > 28:
                            dd = "28"
                        rep = np.datetime64(f"2017-{m}-{dd}") + (date - day)            # date.replace(year = 2017 [, day = 28]), hour kept
                        k = int(np.searchsorted(self.times, rep))
                        if k >= len(self.times) or self.times[k] != rep:
                            raise KeyError(f"prescribed variable {p!r}: {rep} is not in the record (the reference's ds.tisr.sel(time=date) raises too)")
                        extra.append(np.asarray(self.fields["tisr"][k]))                # (:358 reads `tisr` whatever the variable is called)
                    data = np.concatenate((data, np.asarray(extra)))
                ps.append(self._norm(data, self.stats[p]) if self.normalize else data)
            prescribed = np.float32(np.stack(ps, axis=1))
        else:
            prescribed = torch.nan
        prog = []
        a, b = self._label_slice(d0, d0 + (L_ + 1) * dt)                                # :374-377 (inclusive: L + 2 frames while the record lasts)
        for p, levels in self.prognostic_variable_names_and_levels.items():
            if levels:
                for l in levels:
                    x = np.asarray(self.fields[p][l][a:b])
                    prog.append(self._norm(x, self.stats[p]["level"][l]) if self.normalize else x)
            else:
                x = np.asarray(self.fields[p][a:b])
                prog.append(self._norm(x, self.stats[p]) if self.normalize else x)
        prognostic = np.float32(np.stack(prog, axis=1))
        if len(prognostic) < L_:                                                        # :386-389
            fill = np.zeros((L_ - len(prognostic), *prognostic.shape[1:]), dtype=np.float32)
            prognostic = np.concatenate((prognostic, fill), axis=0)
        target = prognostic[1:]
        eps = (np.random.randn(*prognostic[:-1].shape) if self.seed is None else
               np.random.default_rng([self.seed, int(self.epoch), int(item)]).standard_normal(prognostic[:-1].shape))
        prognostic = prognostic[:-1] + np.float32(eps * self.noise)
        return self.constants, prescribed, prognostic, target[self.context_size:]

    def __getitem__(self, item):
        if self.init_dates is not None:
            return self._getitem_init_date(item)
        L_ = self.sequence_length
        t0 = item * L_                                                                  # :335
        if self.prescribed_variable_names:                                              # [L, #prescribed, lat, lon]
            ps = []
            for p in self.prescribed_variable_names:
                a = np.asarray(self.fields[p][t0:t0 + L_])                              # :343
                ps.append(self._norm(a, self.stats[p]) if self.normalize else a)        # :362
            prescribed = np.float32(np.stack(ps, axis=1))
        else:
            prescribed = torch.nan                                                      # :366
        prog = []
        for p, levels in self.prognostic_variable_names_and_levels.items():            # [L + 1, #prognostic, lat, lon]
            if levels:
                for l in levels:                                                        # :378-382
                    a = np.asarray(self.fields[p][l][t0:t0 + L_ + 1])
                    prog.append(self._norm(a, self.stats[p]["level"][l]) if self.normalize else a)
            else:
                a = np.asarray(self.fields[p][t0:t0 + L_ + 1])                          # :371
                prog.append(self._norm(a, self.stats[p]) if self.normalize else a)
        prognostic = np.float32(np.stack(prog, axis=1))
        if len(prognostic) < L_:                                                        # zero fill past the record (:386-389)
            fill = np.zeros((L_ - len(prognostic), *prognostic.shape[1:]), dtype=np.float32)
            prognostic = np.concatenate((prognostic, fill), axis=0)
        target = prognostic[1:]                                                         # :392
        if self.seed is None:
            eps = np.random.randn(*prognostic[:-1].shape)
        else:                       # DDP: the noise of a sample must not depend on which rank draws it (ddp.py)
            # per-(epoch, item) stream: the perturbation of a sample changes from epoch to epoch (set_epoch) and does not
            # depend on the number of ranks
            eps = np.random.default_rng([self.seed, int(self.epoch), int(item)]).standard_normal(prognostic[:-1].shape)
        prognostic = prognostic[:-1] + np.float32(eps * self.noise)                     # :393
        return self.constants, prescribed, prognostic, target[self.context_size:]       # :395


def make_biweekly_inits(start="2017-01-01", end="2018-12-31", sequence_length=57, timedelta=6):
    """scripts/evaluate.py:56-68: two interleaved weekly series of initialisation dates (start and start + 3 days), each ending
    sequence_length * timedelta hours before `end`, merged and sorted.  numpy datetime64[ns] like `DatetimeIndex.to_numpy()`."""
    s0, e = np.datetime64(start, "h"), np.datetime64(end, "h") - np.timedelta64(int(sequence_length) * int(timedelta), "h")
    week = np.timedelta64(7 * 24, "h")
    series = []
    for first in (s0, s0 + np.timedelta64(3 * 24, "h")):
        n = int((e - first) / week) + 1 if e >= first else 0
        series.append(first + week * np.arange(n))
    return np.sort(np.concatenate(series)).astype("datetime64[ns]")


def load_netcdf3_fields(paths, prognostic_variable_names_and_levels, prescribed_variable_names=None, constant_names=None,
                        start_date=None, stop_date=None, timedelta=1):
    """The `fields` / `times` arguments of WeatherBenchArrays from NetCDF-3 CLASSIC files of the WeatherBench layout (one variable per file
    or several; dimensions time[, level], lat, lon; `time` in "hours since ..." as WeatherBench writes it) -- the part of
    `xr.open_mfdataset(fpaths).sel(time=slice(start_date, stop_date, timedelta))` (datasets.py:284-287) that this image can do:
    scipy.io.netcdf_file reads the classic format only; NetCDF-4 / HDF5 files and blosc-compressed zarr stores need netCDF4 / h5py / zarr,
    none of which is installed (convert with `nccopy -k classic` or `xarray.to_netcdf(engine="scipy")`).
    Returns (fields, times): surface variables [time, lat, lon], level-resolved ones {level: [time, lat, lon]}, constants [lat, lon]."""
    import re
    from scipy.io import netcdf_file
    want_t = set(prognostic_variable_names_and_levels) | set(prescribed_variable_names or [])
    want_c = set(constant_names or [])
    fields, times = {}, None
    for path in sorted(paths):
        try:
            f = netcdf_file(path, "r", mmap=False)
        except (TypeError, ValueError) as e:
            raise OSError(f"{path}: not a NetCDF-3 classic file (NetCDF-4 / HDF5 and zarr need libraries this image lacks): {e}") from e
        with f:
            tvals = None
            if "time" in f.variables:
                tv = f.variables["time"]
                units = tv.units.decode() if isinstance(tv.units, bytes) else str(tv.units)
                m = re.match(r"\s*(hours|days|seconds|minutes) since (\S+)(?:[ T](\S+))?", units)
                if not m:
                    raise OSError(f"{path}: time units {units!r} not understood")
                step = {"hours": "h", "days": "D", "seconds": "s", "minutes": "m"}[m.group(1)]
                origin = np.datetime64(m.group(2) + ("T" + m.group(3)[:8] if m.group(3) else ""))
                tvals = (origin.astype("datetime64[s]") + (np.array(tv[:], dtype=np.float64) * {"h": 3600, "D": 86400, "s": 1, "m": 60}[step]
                                                           ).astype("timedelta64[s]")).astype("datetime64[h]")
            for name, var in f.variables.items():
                if name not in want_t and name not in want_c:
                    continue
                data = np.array(var[:], dtype=np.float32)
                if hasattr(var, "scale_factor") or hasattr(var, "add_offset"):          # packed shorts, as ERA5 downloads are
                    data = data * np.float32(getattr(var, "scale_factor", 1.0)) + np.float32(getattr(var, "add_offset", 0.0))
                if name in want_c:
                    fields[name] = data.reshape(data.shape[-2:])
                    continue
                levels = np.array(f.variables["level"][:]).astype(int).tolist() if "level" in var.dimensions else None
                if levels is not None:
                    piece = {l: data[:, i] for i, l in enumerate(levels)}
                    prev = fields.get(name)
                    fields[name] = piece if prev is None else {l: np.concatenate((prev[l], piece[l])) for l in piece}
                else:
                    prev = fields.get(name)
                    fields[name] = data if prev is None else np.concatenate((prev, data))
                if tvals is not None and name == next(iter(prognostic_variable_names_and_levels)):
                    times = tvals if times is None else np.concatenate((times, tvals))
    if times is not None:                                                              # .sel(time=slice(start, stop, timedelta)): inclusive labels
        a = 0 if start_date is None else int(np.searchsorted(times, np.datetime64(start_date, "h"), side="left"))
        b = len(times) if stop_date is None else int(np.searchsorted(times, np.datetime64(stop_date, "h"), side="right"))
        sl = slice(a, b, int(timedelta))
        times = times[sl]
        for name in want_t & set(fields):
            fields[name] = {l: v[sl] for l, v in fields[name].items()} if isinstance(fields[name], dict) else fields[name][sl]
    return fields, times


def to_device_batch(items, device):
    """Collate `__getitem__` tuples like the default DataLoader collate + train.py:116-121: stacked tensors on `device`,
    absent inputs (NaN scalars) -> None."""
    def stack(k):
        vals = [it[k] for it in items]
        if not isinstance(vals[0], np.ndarray):
            return None
        return torch.from_numpy(np.stack(vals)).to(device)
    return stack(0), stack(1), stack(2), stack(3)


def shard_batches(dataset, epoch, rank, world, batch, seed=1234, drop_last=True):
    """The [n_iters, batch] item indices this rank trains on in `epoch` (one seeded permutation shared by all ranks)."""
    return ddp.shard_indices(len(dataset), epoch, rank, world, batch, seed=seed, drop_last=drop_last)


def synthetic_fields(n_time, height=32, width=64, prognostic=None, prescribed=("tisr",), constants=("orography", "lsm", "lat2d", "lon2d"),
                     seed=1234, stats=None):
    """Seeded smooth random fields with the statistics of the real variables (no dataset access on the GPU box): for
    benchmarks and tests of the loader -> rollout step path."""
    stats = STATISTICS if stats is None else stats
    prognostic = prognostic or {"t": [850], "t2m": [], "u10": [], "v10": [], "z": [300, 500, 700, 1000]}
    rng = np.random.default_rng(seed)
    lat = np.linspace(-90 + 90 / height, 90 - 90 / height, height)
    lon = np.linspace(0, 360, width, endpoint=False)

    def smooth(shape_t, st):
        k = rng.standard_normal((shape_t, 3, 3))
        phase = rng.uniform(0, 2 * math.pi, (shape_t, 3, 3))
        y = np.zeros((shape_t, height, width))
        for a in range(3):
            for b in range(3):
                y += k[:, a, b, None, None] * np.cos(np.deg2rad(lat)[None, :, None] * (a + 1) + np.deg2rad(lon)[None, None, :] * b
                                                    + phase[:, a, b, None, None])
        y = y / max(y.std(), 1e-12)
        return np.float32(st["mean"] + st["std"] * y)

    fields = {}
    for p, levels in prognostic.items():
        fields[p] = {l: smooth(n_time, stats[p]["level"][l]) for l in levels} if levels else smooth(n_time, stats[p])
    for p in prescribed:
        fields[p] = smooth(n_time, stats[p])
    for c in constants:
        if c == "lat2d":
            fields[c] = np.float32(np.repeat(lat[:, None], width, axis=1))
        elif c == "lon2d":
            fields[c] = np.float32(np.repeat(lon[None, :], height, axis=0))
        else:
            fields[c] = smooth(1, stats[c])[0]
    return fields, prognostic, list(prescribed), list(constants)

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

Only the randomly-sampled-initialisation branch (`init_dates is None`, the training path, :327, :335, :368) is on the hot
path; the `init_dates` branch (evaluation with calendar look-ups through pandas) is not built.
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
                 sequence_length=15, noise=0.0, normalize=False, context_size=1, stats=None, seed=None, **kwargs):
        self.fields = fields
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
        return (self.n_time - self.sequence_length) // self.sequence_length            # :322-323

    def __getitem__(self, item):
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

"""2-D incompressible Navier-Stokes data generator, restated on torch.fft and a NetCDF-free format.

Reference: src/nsbench/data/ns_generation/generate_ns_2d.py:27-130 (pseudo-spectral vorticity solver, Crank-Nicolson
for the viscous term, explicit non-linear term with 2/3 dealiasing), random_fields.py:8-64 (Gaussian random field initial
vorticity) and the driver :170-255 (forcing 0.1 (sin(f pi (x+y)) + cos(f pi (x+y))), record every T / record_steps).
The reference is written against the torch-1.6 API (`th.rfft(w0, 2, onesided=False)`, `th.ifft`), which no longer exists;
this restatement uses the full complex FFT for the forward transforms and the half-spectrum C2R transform for the
inverses, which is what the legacy `irfft(..., onesided=False, signal_sizes=...)` computed.  It is a data tool, not part
of the training hot path: it runs on whatever torch device it is given.

On-disk format: one `.npz` with `a` [N, s, s] (initial vorticity), `u` [N, T, 1, s, s] (recorded solution), `t` [T] and the
scalar attributes of the reference's NetCDF file (:224-233).  `NavierStokesNpz` mirrors NavierStokesDataset
(data/datasets/datasets.py:11-44): random crop of `sequence_length` frames, x = frames[:-1] (+ noise), y = frames[1:].
"""
import math
import os

import numpy as np
import torch


def _wavenumbers(n, device):
    k_max = n // 2
    k = torch.cat((torch.arange(0, k_max, device=device), torch.arange(-k_max, 0, device=device)), 0)
    k_y = k.repeat(n, 1).to(torch.float64)        # varies along the last axis
    return k_y.transpose(0, 1).contiguous(), k_y, k_max


class GaussianRF:
    """random_fields.py:8-64 (dim = 2): samples of N(0, sigma^2 (-Laplace + tau^2)^-alpha) on the periodic unit square."""

    def __init__(self, size, alpha=2.0, tau=3.0, sigma=None, device=None, generator=None):
        self.size, self.device, self.generator = size, device, generator
        if sigma is None:
            sigma = tau ** (0.5 * (2 * alpha - 2))
        k_x, k_y, _ = _wavenumbers(size, device)
        self.sqrt_eig = (size ** 2) * math.sqrt(2.0) * sigma * ((4 * math.pi ** 2 * (k_x ** 2 + k_y ** 2) + tau ** 2) ** (-alpha / 2.0))
        self.sqrt_eig[0, 0] = 0.0

    def sample(self, n):
        coeff = torch.randn(n, self.size, self.size, 2, device=self.device, generator=self.generator, dtype=torch.float64)
        c = torch.complex(self.sqrt_eig * coeff[..., 0], self.sqrt_eig * coeff[..., 1])
        return torch.fft.ifftn(c, dim=(-2, -1)).real.float()


def _c2r(spec, n):
    """Inverse transform of a full [.., n, n] spectrum the way the legacy C2R did it: from the half spectrum."""
    return torch.fft.irfft2(spec[..., : n // 2 + 1], s=(n, n))


def navier_stokes_2d(w0, f, visc, T, delta_t=1e-4, record_steps=1):
    """w0 [B, n, n] initial vorticity, f [n, n] (or [B, n, n]) forcing -> (sol [B, n, n, record_steps], sol_t)."""
    n = w0.shape[-1]
    dev = w0.device
    steps = math.ceil(T / delta_t)
    w_h = torch.fft.fft2(w0.double())
    f_h = torch.fft.fft2(f.double())
    if f_h.dim() < w_h.dim():
        f_h = f_h.unsqueeze(0)
    record_time = math.floor(steps / record_steps)
    k_x, k_y, k_max = _wavenumbers(n, dev)
    lap = 4 * math.pi ** 2 * (k_x ** 2 + k_y ** 2)
    lap[0, 0] = 1.0
    dealias = ((k_y.abs() <= (2.0 / 3.0) * k_max) & (k_x.abs() <= (2.0 / 3.0) * k_max)).double().unsqueeze(0)
    ikx, iky = 2j * math.pi * k_x, 2j * math.pi * k_y
    sol = torch.zeros(*w0.shape, record_steps, device=dev)
    sol_t = torch.zeros(record_steps, device=dev)
    cn_num, cn_den = 1.0 - 0.5 * delta_t * visc * lap, 1.0 + 0.5 * delta_t * visc * lap
    c, t = 0, 0.0
    for j in range(steps):
        psi_h = w_h / lap                                   # stream function: Poisson equation
        q = _c2r(iky * psi_h, n)                            # u_x =  psi_y
        v = _c2r(-ikx * psi_h, n)                           # u_y = -psi_x
        w_x = _c2r(ikx * w_h, n)
        w_y = _c2r(iky * w_h, n)
        F_h = dealias * torch.fft.fft2(q * w_x + v * w_y)   # non-linear term, 2/3 rule
        w_h = (-delta_t * F_h + delta_t * f_h + cn_num * w_h) / cn_den
        t += delta_t
        if (j + 1) % record_time == 0 and c < record_steps:
            sol[..., c] = _c2r(w_h, n).float()
            sol_t[c] = t
            c += 1
    return sol, sol_t


def navier_stokes_2d_hip(w0, f, visc, T, delta_t=1e-4, record_steps=1):
    """The same solver on the GPU with every transform on libdlwpmi's LDS-staged rFFT2 / irFFT2 kernels (fft.py,
    csrc/fft2d.hip; channels-first, norm "backward") in fp32 -- the precision the reference's torch-1.6 solver ran in.  The
    state is the Hermitian HALF spectrum [B, n, n/2+1]; the four inverse transforms of a step (velocity components and
    vorticity gradients) are one batched call.  The point-wise spectral algebra between the transforms is torch elementwise
    work: this is the data tool of SURVEY.md §8f.2, not the training hot path."""
    from . import fft
    n, dev, B = w0.shape[-1], w0.device, w0.shape[0]
    steps = math.ceil(T / delta_t)

    def r2c(x):                                            # [B, C, n, n] -> complex [B, C, n, n/2+1]
        return torch.view_as_complex(fft.rfft2(x.float().contiguous(), "channels_first", "backward"))

    def c2r(X):
        return fft.irfft2(torch.view_as_real(X).contiguous(), n, "channels_first", "backward")

    w_h = r2c(w0[:, None])[:, 0]
    f_h = r2c((f if f.dim() == 3 else f[None]).float()[:, None].to(dev))[:, 0]
    record_time = math.floor(steps / record_steps)
    k_x, k_y, k_max = _wavenumbers(n, dev)
    k_x, k_y = k_x[:, : n // 2 + 1].float(), k_y[:, : n // 2 + 1].float()
    lap = 4 * math.pi ** 2 * (k_x ** 2 + k_y ** 2)
    lap[0, 0] = 1.0
    dealias = ((k_y.abs() <= (2.0 / 3.0) * k_max) & (k_x.abs() <= (2.0 / 3.0) * k_max)).float().unsqueeze(0)
    ikx, iky = torch.complex(torch.zeros_like(k_x), 2 * math.pi * k_x), torch.complex(torch.zeros_like(k_y), 2 * math.pi * k_y)
    sol = torch.zeros(*w0.shape, record_steps, device=dev)
    sol_t = torch.zeros(record_steps, device=dev)
    cn_num, cn_den = 1.0 - 0.5 * delta_t * visc * lap, 1.0 + 0.5 * delta_t * visc * lap
    c, t = 0, 0.0
    for j in range(steps):
        psi_h = w_h / lap
        fields = c2r(torch.stack((iky * psi_h, -ikx * psi_h, ikx * w_h, iky * w_h), dim=1))     # q, v, w_x, w_y
        F_h = dealias * r2c((fields[:, 0] * fields[:, 2] + fields[:, 1] * fields[:, 3])[:, None])[:, 0]
        w_h = (-delta_t * F_h + delta_t * f_h + cn_num * w_h) / cn_den
        t += delta_t
        if (j + 1) % record_time == 0 and c < record_steps:
            sol[..., c] = c2r(w_h[:, None])[:, 0]
            sol_t[c] = t
            c += 1
    return sol, sol_t


def forcing(resolution, forcing_multiplicator=2.0, device=None):
    t = torch.linspace(0, 1, resolution + 1, device=device, dtype=torch.float64)[:-1]
    X, Y = torch.meshgrid(t, t, indexing="ij")
    return 0.1 * (torch.sin(forcing_multiplicator * math.pi * (X + Y)) + torch.cos(forcing_multiplicator * math.pi * (X + Y)))


def generate_data(resolution=64, n_samples=1000, batch_size=50, max_simulation_time=50, delta_t=1e-3, record_steps=None,
                  viscosity=1e-3, alpha=2.5, tau=7.0, forcing_multiplicator=2.0, device="cpu", seed=None):
    """generate_ns_2d.generate_data (:170-221) without the NetCDF writer: returns dict(a, u, t, attrs)."""
    device = torch.device(device)
    record_steps = record_steps or max_simulation_time
    batch_size = min(n_samples, batch_size)
    gen = None
    if seed is not None:
        gen = torch.Generator(device=device).manual_seed(seed)
    grf = GaussianRF(resolution, alpha=alpha, tau=tau, device=device, generator=gen)
    f = forcing(resolution, forcing_multiplicator, device)
    a = torch.zeros(n_samples, resolution, resolution)
    u = torch.zeros(n_samples, record_steps, 1, resolution, resolution)
    sol_t = None
    for c in range(0, (n_samples // batch_size) * batch_size, batch_size):
        w0 = grf.sample(batch_size)
        solver = navier_stokes_2d_hip if device.type == "cuda" else navier_stokes_2d
        sol, sol_t = solver(w0, f, viscosity, max_simulation_time, delta_t, record_steps)
        a[c:c + batch_size] = w0.cpu()
        u[c:c + batch_size] = sol.permute(0, 3, 1, 2).unsqueeze(2).cpu()
    attrs = {"info": "Incompressible Navier-Stokes data", "viscosity": viscosity, "delta_t": "%.e" % delta_t,
             "simulation T": max_simulation_time, "recorded steps": record_steps}
    return {"a": a.numpy(), "u": u.numpy(), "t": sol_t.cpu().numpy(), "attrs": attrs}


def default_name(viscosity, n_samples, T, resolution):
    return f"ns_r{'%.e' % int(1 / viscosity)}_n{n_samples}_t{T}_s{resolution}.npz"   # reference naming (:250), .npz


def save(data, path):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez_compressed(path, a=data["a"], u=data["u"], t=data["t"], **{"attr_" + k.replace(" ", "_"): v for k, v in data["attrs"].items()})


def save_netcdf(data, path):
    """The reference's file layout (generate_ns_2d.py:223-260: coordinates sample / time / dim / height / width, variables a, u, t, the
    attributes) as NetCDF-3 CLASSIC through scipy.io.netcdf_file -- what `xarray.Dataset.to_netcdf` itself writes where the netCDF4
    library is absent, and what `xr.open_dataset` reads back either way.  (NetCDF-4 / HDF5 files need h5py or netCDF4: not in this image.)"""
    from scipy.io import netcdf_file
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    u, a, t = np.asarray(data["u"], dtype=np.float32), np.asarray(data["a"], dtype=np.float32), np.asarray(data["t"], dtype=np.float32)
    dims = dict(zip(("sample", "time", "dim", "height", "width"), u.shape))
    with netcdf_file(path, "w", version=2) as f:          # 64-bit offsets: the 1000-sample files exceed 2 GiB
        for name, n in dims.items():
            f.createDimension(name, n)
            v = f.createVariable(name, "i4", (name,))
            v[:] = np.arange(n, dtype=np.int32)
        f.createVariable("a", "f4", ("sample", "height", "width"))[:] = a
        f.createVariable("u", "f4", tuple(dims))[:] = u
        f.createVariable("t", "f4", ("time",))[:] = t
        for k, val in {"info": "Incompressible Navier-Stokes data", "a": "Initial condition", "u": "Solution", "t": "Time step in simulation",
                       **data.get("attrs", {})}.items():
            if not isinstance(val, (str, bytes)):          # the classic format has no 64-bit integers
                val = np.asarray(val)
                val = val.astype(np.int32) if val.dtype.kind in "iu" else val.astype(np.float64) if val.dtype.kind == "f" else str(val)
            setattr(f, k.replace(" ", "_"), val)


def load_u(data_path):
    """The solution array u [sample, time, dim, height, width] of a data file: .npz (this build's format) or NetCDF-3 classic (.nc)."""
    if str(data_path).endswith(".npz"):
        return np.load(data_path)["u"]
    from scipy.io import netcdf_file
    try:
        with netcdf_file(data_path, "r", mmap=False) as f:
            return np.array(f.variables["u"][:], dtype=np.float32)
    except (TypeError, ValueError) as e:          # scipy: "... is not a valid NetCDF 3 file" for an HDF5-based NetCDF-4 file
        raise OSError(f"{data_path}: only NetCDF-3 classic files can be read in this environment (scipy.io.netcdf_file; NetCDF-4 / HDF5 "
                      f"needs h5py or netCDF4, which the image does not have): {e}") from e


class NavierStokesNpz(torch.utils.data.Dataset):
    """NavierStokesDataset (datasets.py:11-44) on the .npz file or on a NetCDF-3 classic file of the reference's layout (load_u)."""

    def __init__(self, data_path, sequence_length=15, noise=0.0, normalize=False, downscale_factor=None):
        self.u = load_u(data_path)
        self.sequence_length, self.noise, self.normalize = sequence_length, noise, normalize
        self.mean, self.std = self.u.mean(), self.u.std()
        if downscale_factor:
            n, t, d, h, w = self.u.shape
            k = downscale_factor
            self.u = self.u.reshape(n, t, d, h // k, k, w // k, k).mean(axis=(4, 6))

    def __len__(self):
        return self.u.shape[0]

    def __getitem__(self, item):
        r = np.random.randint(0, self.u.shape[1] - self.sequence_length + 1)
        x = np.float32(self.u[item, r:r + self.sequence_length - 1])
        x = x + np.float32(np.random.randn(*x.shape) * self.noise)
        y = np.float32(self.u[item, 1 + r:r + self.sequence_length])
        return x, y


def main():
    import argparse
    ap = argparse.ArgumentParser(description="Incompressible Navier-Stokes data generation (.npz)")
    ap.add_argument("-r", "--resolution", type=int, default=64)
    ap.add_argument("-n", "--n_samples", type=int, default=1000)
    ap.add_argument("-b", "--batch-size", type=int, default=50)
    ap.add_argument("-t", "--max-simulation-time", type=int, default=50)
    ap.add_argument("--delta-t", type=float, default=1e-3)
    ap.add_argument("-s", "--record-steps", type=int, default=None)
    ap.add_argument("-v", "--viscosity", type=float, default=1e-3)
    ap.add_argument("-a", "--alpha", type=float, default=2.0)      # CLI default of the reference (:280)
    ap.add_argument("--tau", type=float, default=7.0)
    ap.add_argument("--forcing-multiplicator", type=float, default=2.0)
    ap.add_argument("-d", "--device", default="cpu")
    ap.add_argument("-p", "--dst-path", default=os.path.join("data", "npz", "navier-stokes"))
    ap.add_argument("--seed", type=int, default=None)
    a = ap.parse_args()
    data = generate_data(a.resolution, a.n_samples, a.batch_size, a.max_simulation_time, a.delta_t, a.record_steps,
                         a.viscosity, a.alpha, a.tau, a.forcing_multiplicator, a.device, a.seed)
    path = os.path.join(a.dst_path, default_name(a.viscosity, a.n_samples, a.max_simulation_time, a.resolution))
    save(data, path)
    print("wrote", path)


if __name__ == "__main__":
    main()

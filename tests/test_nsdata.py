"""Self-consistency of the Navier-Stokes data generator (the reference generator needs the torch-1.6 FFT
API and cannot run here, so there is nothing to pin against): one solver step against an independent numpy
implementation, enstrophy decay without forcing, first-order convergence in dt, the 2/3 dealiasing rule, field
statistics, and the dataset's windowing."""
import math

import numpy as np
import torch

from dlwp_benchmark_amd import nsdata


def numpy_step(w, f, visc, dt):
    """One step of generate_ns_2d.py:71-113 written independently with numpy's complex FFT."""
    n = w.shape[-1]
    k = np.fft.fftfreq(n, 1.0 / n)
    kx, ky = np.meshgrid(k, k, indexing="ij")
    kx[n // 2, :] = -(n // 2); ky[:, n // 2] = -(n // 2)                      # noqa: E702 (the reference's -k_max at Nyquist)
    lap = 4 * np.pi ** 2 * (kx ** 2 + ky ** 2)
    lap[0, 0] = 1.0
    wh, fh = np.fft.fft2(w), np.fft.fft2(f)
    psi = wh / lap
    c2r = lambda s: np.fft.irfft2(s[..., : n // 2 + 1], s=(n, n))            # noqa: E731
    q, v = c2r(2j * np.pi * ky * psi), c2r(-2j * np.pi * kx * psi)
    wx, wy = c2r(2j * np.pi * kx * wh), c2r(2j * np.pi * ky * wh)
    mask = (np.abs(ky) <= 2 / 3 * (n // 2)) & (np.abs(kx) <= 2 / 3 * (n // 2))
    Fh = mask * np.fft.fft2(q * wx + v * wy)
    wh = (-dt * Fh + dt * fh + (1 - 0.5 * dt * visc * lap) * wh) / (1 + 0.5 * dt * visc * lap)
    return c2r(wh)


def test_one_step_matches_independent_numpy_solver():
    g = torch.Generator().manual_seed(1)
    w0 = nsdata.GaussianRF(32, alpha=2.5, tau=7.0, generator=g).sample(2)
    f = nsdata.forcing(32)
    sol, t = nsdata.navier_stokes_2d(w0, f, 1e-3, 1e-3, delta_t=1e-3, record_steps=1)
    ref = np.stack([numpy_step(w0[i].double().numpy(), f.numpy(), 1e-3, 1e-3) for i in range(2)])
    np.testing.assert_allclose(sol[..., 0].numpy(), ref, rtol=0, atol=2e-6 * np.abs(ref).max())
    assert abs(t.item() - 1e-3) < 1e-9


def test_enstrophy_decays_without_forcing():
    g = torch.Generator().manual_seed(2)
    w0 = nsdata.GaussianRF(32, alpha=2.5, tau=7.0, generator=g).sample(1)
    sol, _ = nsdata.navier_stokes_2d(w0, torch.zeros(32, 32, dtype=torch.float64), 1e-2, 1.0, delta_t=5e-3, record_steps=10)
    ens = [(w0 ** 2).mean().item()] + [(sol[..., i] ** 2).mean().item() for i in range(10)]
    assert all(b < a for a, b in zip(ens, ens[1:]))
    assert abs(sol.mean().item()) < 1e-6            # the mean vorticity stays zero


def test_time_step_convergence_is_first_order():
    g = torch.Generator().manual_seed(3)
    w0 = nsdata.GaussianRF(32, alpha=2.5, tau=7.0, generator=g).sample(1)
    f = nsdata.forcing(32)
    run = lambda dt: nsdata.navier_stokes_2d(w0, f, 1e-3, 0.2, delta_t=dt, record_steps=1)[0][..., 0]   # noqa: E731
    a, b, c = run(4e-3), run(2e-3), run(1e-3)
    e1, e2 = (a - b).abs().max().item(), (b - c).abs().max().item()
    assert 1.6 < e1 / e2 < 2.6                      # explicit Euler on the advection term


def test_random_field_statistics_and_zero_mean():
    g = torch.Generator().manual_seed(4)
    grf = nsdata.GaussianRF(64, alpha=2.0, tau=7.0, generator=g)
    u = grf.sample(200)
    assert abs(u.mean().item()) < 5e-3 and u.mean(dim=(-2, -1)).abs().max().item() < 1e-5     # k = 0 mode removed
    # E[u^2] = (1 / (2 n^4)) * 2 * sum sqrt_eig^2 ... : real part of a complex field with independent re/im coefficients
    expect = (grf.sqrt_eig ** 2).sum().item() / 64 ** 4
    assert abs(u.pow(2).mean().item() / expect - 1.0) < 0.05


def test_generate_save_and_dataset_windowing(tmp_path):
    data = nsdata.generate_data(resolution=16, n_samples=4, batch_size=2, max_simulation_time=3, delta_t=1e-2, seed=5)
    assert data["u"].shape == (4, 3, 1, 16, 16) and data["a"].shape == (4, 16, 16)
    np.testing.assert_allclose(data["t"], [1.0, 2.0, 3.0], atol=1e-5)
    path = tmp_path / nsdata.default_name(1e-3, 4, 3, 16)
    nsdata.save(data, str(path))
    ds = nsdata.NavierStokesNpz(str(path), sequence_length=3)
    x, y = ds[1]
    assert x.shape == (2, 1, 16, 16) and y.shape == (2, 1, 16, 16)
    np.testing.assert_array_equal(x[1], y[0])       # y is x shifted by one frame


def test_netcdf3_file_of_the_reference_layout_round_trips(tmp_path):
    """nsdata.save_netcdf writes the reference's file layout (generate_ns_2d.py:223-260) as NetCDF-3 classic; NavierStokesNpz reads it back
    (scipy.io.netcdf_file: the NetCDF flavour xarray itself falls back to without the netCDF4 library)."""
    import numpy as np
    from dlwp_benchmark_amd import nsdata
    rng = np.random.default_rng(0)
    data = {"a": rng.standard_normal((3, 8, 8)).astype(np.float32), "u": rng.standard_normal((3, 6, 1, 8, 8)).astype(np.float32),
            "t": np.arange(6, dtype=np.float32), "attrs": {"viscosity": 1e-3, "simulation T": 6}}
    path = str(tmp_path / "ns.nc")
    nsdata.save_netcdf(data, path)
    assert np.array_equal(nsdata.load_u(path), data["u"])
    ds = nsdata.NavierStokesNpz(path, sequence_length=4)
    x, y = ds[1]
    assert len(ds) == 3 and x.shape == (3, 1, 8, 8) and y.shape == (3, 1, 8, 8)
    from scipy.io import netcdf_file
    with netcdf_file(path, "r", mmap=False) as f:
        assert set(f.dimensions) == {"sample", "time", "dim", "height", "width"} and f.variables["u"].dimensions == ("sample", "time", "dim", "height", "width")

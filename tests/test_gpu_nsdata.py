"""The Navier-Stokes data generator's GPU leg (nsdata.navier_stokes_2d_hip: fp32, every transform on libdlwpmi's rFFT2 /
irFFT2 kernels) against the float64 CPU solver of the same restatement (tests/test_nsdata.py checks that one against an
independent numpy step, enstrophy decay and dt convergence)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gpu_solver_tracks_the_cpu_solver(cuda):
    from dlwp_benchmark_amd import nsdata
    n, B = 64, 3
    g = torch.Generator().manual_seed(4)
    w0 = nsdata.GaussianRF(n, alpha=2.5, tau=7.0, generator=g).sample(B)
    f = nsdata.forcing(n)
    ref, t_ref = nsdata.navier_stokes_2d(w0, f, 1e-3, T=2.0, delta_t=5e-3, record_steps=4)
    got, t_got = nsdata.navier_stokes_2d_hip(w0.to(cuda), f.to(cuda), 1e-3, T=2.0, delta_t=5e-3, record_steps=4)
    assert torch.allclose(t_got.cpu(), t_ref, atol=1e-6)
    err = (got.cpu().double() - ref.double()).abs().max() / ref.double().abs().max()
    assert err <= 2e-4, err                      # fp32 transforms and algebra over 400 time steps vs float64


def test_generate_data_on_the_gpu_matches_the_cpu_generator(cuda):
    from dlwp_benchmark_amd import nsdata
    kw = dict(resolution=32, n_samples=4, batch_size=4, max_simulation_time=3, delta_t=1e-2)
    cpu = nsdata.generate_data(seed=5, **kw)
    # same initial fields: the GPU run re-uses the CPU generator's samples (the device generators draw different streams)
    w0 = torch.from_numpy(cpu["a"]).to(cuda)
    sol, _ = nsdata.navier_stokes_2d_hip(w0, nsdata.forcing(32, device=cuda), 1e-3, 3, 1e-2, 3)
    u = sol.permute(0, 3, 1, 2).unsqueeze(2).cpu()
    assert u.shape == cpu["u"].shape
    assert (u - torch.from_numpy(cpu["u"])).abs().max() / torch.from_numpy(cpu["u"]).abs().max() <= 2e-4

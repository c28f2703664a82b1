"""rFFT2 / irFFT2 kernels (csrc/fft2d.hip) against torch.fft on the CPU (the fp32 reference of a floating-point kernel):
channels-last "ortho" (the AFNO2D call, fourcastnet.py:84,123) and channels-first "forward" (neuralop's SpectralConv),
power-of-two, mixed-radix and prime-factor sizes up to BASELINE C5's 721 x 1440, forward values and the adjoints used as
backward passes.  Tolerance 1e-5 relative to the max norm."""
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-5


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


SHAPES = [(2, 64, 64, 8), (1, 128, 256, 16), (1, 32, 64, 6), (2, 30, 36, 4), (1, 103, 180, 4), (1, 45, 50, 2), (1, 721, 1440, 2),
          (3, 16, 9, 2), (1, 7, 11, 4),
          # the FourCastNet token grid 90 x 180 (C5): both axes on their compile-time plans (round 5), channel counts that leave
          # partial lane groups; each plan next to the run-time plan of the other axis
          (2, 90, 180, 70), (1, 90, 180, 768), (1, 90, 64, 6), (1, 36, 180, 34)]


@pytest.mark.parametrize("B,H,W,C", SHAPES)
@pytest.mark.parametrize("layout,norm", [("channels_last", "ortho"), ("channels_first", "forward"), ("channels_last", "backward")])
def test_rfft2_irfft2_match_torch(cuda, B, H, W, C, layout, norm):
    from dlwp_benchmark_amd import fft
    g = torch.Generator().manual_seed(H * 1000 + W)
    cl = layout == "channels_last"
    x = torch.randn((B, H, W, C) if cl else (B, C, H, W), generator=g)
    dims = (1, 2) if cl else (2, 3)
    Xr = torch.fft.rfft2(x.double(), dim=dims, norm=norm)
    X = fft.rfft2(x.to(cuda), layout, norm)
    assert X.shape[-1] == 2
    assert rel(torch.view_as_complex(X.cpu().contiguous()), Xr) <= TOL
    # inverse of a generic Hermitian half spectrum (imaginary parts of DC / Nyquist set at random: torch ignores them too)
    Y = torch.randn(*Xr.shape, 2, generator=g)
    yr = torch.fft.irfft2(torch.view_as_complex(Y.double()), s=(H, W), dim=dims, norm=norm)
    y = fft.irfft2(Y.to(cuda), W, layout, norm)
    assert rel(y, yr) <= TOL
    # round trip
    assert rel(fft.irfft2(X, W, layout, norm), x) <= 2 * TOL


@pytest.mark.parametrize("B,H,W,C", [(2, 16, 24, 4), (1, 64, 64, 8), (1, 21, 30, 2), (1, 103, 180, 2)])
@pytest.mark.parametrize("layout,norm", [("channels_last", "ortho"), ("channels_first", "forward")])
def test_fft_backward_passes_are_the_adjoints(cuda, B, H, W, C, layout, norm):
    """<rfft2 x, G> == <x, rfft2^H G> and the same for irfft2, against torch autograd on the CPU (real pairs)."""
    from dlwp_benchmark_amd import fft
    g = torch.Generator().manual_seed(7)
    cl = layout == "channels_last"
    dims = (1, 2) if cl else (2, 3)
    x = torch.randn((B, H, W, C) if cl else (B, C, H, W), generator=g)
    xr = x.double().requires_grad_(True)
    Xr = torch.view_as_real(torch.fft.rfft2(xr, dim=dims, norm=norm))
    G = torch.randn(Xr.shape, generator=g)
    (Xr * G.double()).sum().backward()
    xd = x.to(cuda).requires_grad_(True)
    X = fft.rfft2(xd, layout, norm)
    (X * G.to(cuda)).sum().backward()
    assert rel(xd.grad, xr.grad) <= TOL
    Y = torch.randn(Xr.shape, generator=g)
    Yr = Y.double().requires_grad_(True)
    yr = torch.fft.irfft2(torch.view_as_complex(Yr), s=(H, W), dim=dims, norm=norm)
    Gy = torch.randn(yr.shape, generator=g)
    (yr * Gy.double()).sum().backward()
    Yd = Y.to(cuda).requires_grad_(True)
    y = fft.irfft2(Yd, W, layout, norm)
    (y * Gy.to(cuda)).sum().backward()
    assert rel(Yd.grad, Yr.grad) <= TOL


@pytest.mark.parametrize("B,H,W,C,win,bs", [(2, 16, 24, 4, None, 0), (1, 90, 180, 8, (0, 90, 46), 0), (1, 103, 180, 2, (21, 83, 31), 0),
                                            (2, 7, 11, 4, (2, 6, 3), 0), (1, 45, 64, 40, (9, 37, 14), 8), (2, 16, 24, 12, None, 4),
                                            (1, 90, 180, 96, (0, 90, 46), 48), (2, 7, 11, 6, (2, 6, 3), 2)])
def test_planar_window_transforms_match_torch(cuda, B, H, W, C, win, bs):
    """rfft2_planar / irfft2_planar (a window of the half spectrum as [2 (re | im), B, rows, cols, C], the AFNO mixer's GEMM
    operand) against torch.fft on the CPU with the reference's slicing / zero-initialised inverse (fourcastnet.py:85-124):
    values both ways and both backward passes."""
    from dlwp_benchmark_amd import fft
    g = torch.Generator().manual_seed(H + W)
    r0, r1, c1 = win or (0, H, W // 2 + 1)
    x = torch.randn(B, H, W, C, generator=g)
    xr = x.double().requires_grad_(True)
    Xr = torch.view_as_real(torch.fft.rfft2(xr, dim=(1, 2), norm="ortho")[:, r0:r1, :c1]).permute(4, 0, 1, 2, 3)
    G = torch.randn(Xr.shape, generator=g)
    (Xr * G.double()).sum().backward()
    xd = x.to(cuda).requires_grad_(True)
    def to_layout(t):          # planar [2, B, R, c1, C] -> the layout under test
        return t.reshape(2, B, r1 - r0, c1, C // bs, bs).permute(1, 2, 3, 4, 0, 5) if bs else t

    def from_layout(t):
        return t.permute(4, 0, 1, 2, 3, 5).reshape(2, B, r1 - r0, c1, C) if bs else t

    X = fft.rfft2_planar(xd, "ortho", win, block=bs)
    assert X.shape == ((B, r1 - r0, c1, C // bs, 2, bs) if bs else (2, B, r1 - r0, c1, C))
    (X * to_layout(G.to(cuda))).sum().backward()
    X = from_layout(X)
    assert rel(X, Xr) <= TOL
    assert rel(xd.grad, xr.grad) <= TOL
    Y = torch.randn(Xr.shape, generator=g)
    Yr = Y.double().requires_grad_(True)
    full = torch.nn.functional.pad(Yr.permute(1, 2, 3, 4, 0), (0, 0, 0, 0, 0, W // 2 + 1 - c1, r0, H - r1))     # zero outside the window
    full = torch.view_as_complex(full.contiguous())
    yr = torch.fft.irfft2(full, s=(H, W), dim=(1, 2), norm="ortho")
    Gy = torch.randn(yr.shape, generator=g)
    (yr * Gy.double()).sum().backward()
    Yd = to_layout(Y.to(cuda)).contiguous().requires_grad_(True)
    y = fft.irfft2_planar(Yd, H, W, "ortho", win, block=bs)
    (y * Gy.to(cuda)).sum().backward()
    assert rel(y, yr) <= TOL
    assert rel(from_layout(Yd.grad), Yr.grad) <= TOL


@pytest.mark.parametrize("spw,sph", [(1, 2), (3, 4), (5, 6), (0, 0)])
def test_every_compile_time_plan_matches_torch(cuda, spw, sph):
    """The selectable compile-time plans of the 90 x 180 grid (DLWP_FFT_SPW / DLWP_FFT_SPH; 0 / 0 = DLWP_FFT_STATIC=0, the run-time
    plan) through the C ABI with a plan made under that tuning: rfft2 and irfft2 against torch.fft, 1e-5 of the max norm."""
    import ctypes as C
    from dlwp_benchmark_amd import lib as L
    lib = L.load()
    lib.dlwp_set_tuning.argtypes = [C.c_char_p, C.c_int]
    lib.dlwp_clear_tuning.argtypes = [C.c_char_p]
    B, H, W, Cc = 2, 90, 180, 44
    g = torch.Generator().manual_seed(spw * 10 + sph)
    x = torch.randn(B, H, W, Cc, generator=g)
    ref = torch.fft.rfft2(x.double(), dim=(1, 2), norm="ortho")
    knobs = {b"FFT_SPW": spw, b"FFT_SPH": sph} if spw else {b"FFT_STATIC": 0}
    plan = C.c_void_p()
    try:
        for k, v in knobs.items():
            lib.dlwp_set_tuning(k, v)
        L.check(lib.dlwp_fft_plan_create(H, W, C.byref(plan)))
    finally:
        for k in knobs:
            lib.dlwp_clear_tuning(k)
    try:
        xd = x.to(cuda)
        X = torch.empty(B, H, W // 2 + 1, Cc, 2, device=cuda)
        L.check(lib.dlwp_rfft2(plan, L.ptr(xd), L.ptr(X), B, Cc, 0, 1, 0, L.stream()))
        assert rel(torch.view_as_complex(X), ref) <= TOL
        y, work = torch.empty_like(xd), torch.empty_like(X)
        L.check(lib.dlwp_irfft2(plan, L.ptr(X), L.ptr(y), L.ptr(work), B, Cc, 0, 1, 0, L.stream()))
        assert rel(y, x) <= TOL
    finally:
        torch.cuda.synchronize()
        lib.dlwp_fft_plan_destroy(plan)


def test_irfft2_planar_with_two_residuals(cuda):
    """dlwp_irfft2_planar2: both fields ride the inverse transform's store -- equal to the one-residual call plus the second field."""
    from dlwp_benchmark_amd import fft, lib as L
    g = torch.Generator().manual_seed(12)
    B, H, W, Cc, bs = 1, 90, 180, 32, 8
    win = (0, 90, 46)
    X = torch.randn(B, 90, 46, Cc // bs, 2, bs, generator=g).to(cuda)
    r1 = torch.randn(B, H, W, Cc, generator=g).to(cuda)
    r2 = torch.randn(B, H, W, Cc, generator=g).to(cuda)
    one = fft._run_c2r_planar(X, H, W, win, bs, fft.NORMS["ortho"], 0, residual=r1)
    two = fft._run_c2r_planar(X, H, W, win, bs, fft.NORMS["ortho"], 0, residual=r1, residual2=r2)
    assert rel(two, one + r2) <= 1e-6
    with pytest.raises(L.DlwpError):
        fft._run_c2r_planar(X, H, W, win, bs, fft.NORMS["ortho"], 0, residual=None, residual2=r2)


@pytest.mark.parametrize("bs", [0, 8])
def test_rfft2_planar_masked_store_is_the_softshrink_derivative(cuda, bs):
    """dlwp_rfft2_planar_masked (adjoint transform of a gradient field, stored components zeroed where |mask| <= lam) equals the
    plain adjoint transform followed by dlwp_act_bwd(act = 3): bit for bit (the same transform, then a select)."""
    from dlwp_benchmark_amd import fft, lib as L
    g = torch.Generator().manual_seed(21 + bs)
    B, H, W, Cc = 2, 30, 36, 16
    win = (3, 27, 12)
    lam = 0.3
    gy = torch.randn(B, H, W, Cc, generator=g).to(cuda)
    plain = fft._run_r2c_planar(gy, win, bs, fft.NORMS["ortho"], 1)
    P = torch.randn(plain.shape, generator=g).to(cuda)
    masked = fft._run_r2c_planar(gy, win, bs, fft.NORMS["ortho"], 1, mask=P, lam=lam)
    want = torch.empty_like(plain)
    L.check(L.load().dlwp_act_bwd(L.ptr(P), L.ptr(plain), L.ptr(want), plain.numel(), 3, lam, L.stream()))
    assert torch.equal(masked, want)
    assert 0.1 < (masked == 0).float().mean().item() < 0.5          # |N(0,1)| <= 0.3 for ~24 % of the elements


@pytest.mark.parametrize("bs", [0, 8])
def test_planar_window_as_bf16_array(cuda, bs):
    """dlwp_rfft2_planar_ex / dlwp_irfft2_planar_ex with flags = 1: the written window is the fp32 window rounded once (masked or
    not, the mask read as bf16); the inverse transform of a bf16 window equals the fp32 entry on the same (widened) values."""
    from dlwp_benchmark_amd import fft
    BF = torch.bfloat16
    g = torch.Generator().manual_seed(31 + bs)
    B, H, W, Cc = 2, 30, 36, 16
    win = (3, 27, 12)
    x = torch.randn(B, H, W, Cc, generator=g).to(cuda)
    X32 = fft._run_r2c_planar(x, win, bs, fft.NORMS["ortho"], 0)
    X16 = fft._run_r2c_planar(x, win, bs, fft.NORMS["ortho"], 0, out_bf16=True)
    assert X16.dtype == BF and torch.equal(X16, X32.to(BF))
    P = torch.randn(X32.shape, generator=g).to(cuda).to(BF)
    m16 = fft._run_r2c_planar(x, win, bs, fft.NORMS["ortho"], 1, mask=P, lam=0.3, out_bf16=True)
    m32 = fft._run_r2c_planar(x, win, bs, fft.NORMS["ortho"], 1, mask=P.float(), lam=0.3)
    assert torch.equal(m16, m32.to(BF))
    r = torch.randn(B, H, W, Cc, generator=g).to(cuda)
    y16 = fft._run_c2r_planar(X16, H, W, win, bs, fft.NORMS["ortho"], 0, residual=r)
    y32 = fft._run_c2r_planar(X16.float(), H, W, win, bs, fft.NORMS["ortho"], 0, residual=r)
    assert torch.equal(y16, y32)

"""Real spherical harmonic transforms and the SFNO spectral convolution on libdlwpmi's batched MFMA GEMM.

Reference: `torch_harmonics.RealSHT / InverseRealSHT` and the "driscoll-healy" spectral layer used by
`torch_harmonics.examples.sfno.SphericalFourierNeuralOperatorNet`, which the reference constructs at
src/dlwpbench/models/fno/fno.py:183-200 and models/fourcastnet/fourcastnet.py:411-428.  torch-harmonics (git 13aa492,
README.md:42-52) is NOT in this image, so the algorithm is restated from its published definition (SURVEY.md App. A-2,
PARITY UNPINNED); correctness of the transforms is established analytically (tests/test_sht.py).

Definitions (colatitude nodes theta_k from the north pole, quadrature weights w_k, P~_l^m orthonormal associated
Legendre functions with Condon-Shortley phase):
    RealSHT:         X[l, m] = sum_k w_k P~_l^m(cos theta_k) * (2 pi / nlon) sum_n x[k, n] e^{-2 pi i m n / nlon}
    InverseRealSHT:  x[k, n] = irfft_n( sum_l P~_l^m(cos theta_k) X[l, m] ),  "forward" norm (no 1/nlon on the inverse)

MI355X mapping.  Tensors are channels-last, x [B, nlat, nlon, C], and the spectrum is degree-major,
X [lmax, B, mmax, 2 (re|im), C].  Each transform is two strided-batched GEMMs against small constant tables:
    longitude DFT    T[b, k][(m, re|im)][c] = F[(m, re|im)][n]        . x[b, k][n][c]        batch (b, k)
    Legendre         X[l][b, m][(re|im, c)] = W[m][l][k]              . T[b][k][m][(re|im, c)] batch (b, m)
(and the transposes for the inverse and for both backward passes), so the whole SFNO block -- transforms, the
per-degree complex weights (one batched GEMM over l on the [[Wr, Wi], [-Wi, Wr]] image), the 1x1-convolution skip and
MLP -- runs on one kernel family.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import lib as L
from .token_ops import _gemm_batched, _grad_slot


# ---- tables (float64 on the host, stored as fp32 buffers) -----------------------------------------------------------
def legendre_gauss_weights(n):
    """Gauss-Legendre nodes (ascending in cos theta) and weights on [-1, 1]."""
    x, w = np.polynomial.legendre.leggauss(n)
    return x, w


def clenshaw_curtiss_weights(n):
    """Clenshaw-Curtis nodes cos(pi j / (n-1)) (ascending, both poles included) and weights on [-1, 1]."""
    assert n > 1
    n1 = n - 1
    theta = np.pi * np.arange(n) / n1
    x = -np.cos(theta)                                  # ascending from -1 to 1
    w = np.zeros(n)
    for k in range(n):
        s = 0.0
        for j in range(1, n1 // 2 + 1):
            b = 1.0 if 2 * j == n1 else 2.0
            s += b / (4.0 * j * j - 1.0) * np.cos(2.0 * j * theta[k])
        c = 1.0 if k in (0, n1) else 2.0
        w[k] = c / n1 * (1.0 - s)
    return x, w


def legpoly(mmax, lmax, x):
    """Orthonormal associated Legendre functions P~_l^m(x) as [mmax, lmax, len(x)] (zero for l < m), with the
    Condon-Shortley phase: the three-term recurrence in l seeded on the diagonal."""
    nmax = max(mmax, lmax)
    vdm = np.zeros((nmax, nmax, len(x)), dtype=np.float64)
    vdm[0, 0, :] = 1.0 / np.sqrt(4.0 * np.pi)
    for l in range(1, nmax):
        vdm[l - 1, l, :] = np.sqrt(2 * l + 1) * x * vdm[l - 1, l - 1, :]
        vdm[l, l, :] = np.sqrt((2 * l + 1) * (1 + x) * (1 - x) / 2 / l) * vdm[l - 1, l - 1, :]
    for l in range(2, nmax):
        for m in range(0, l - 1):
            vdm[m, l, :] = (x * np.sqrt((2 * l - 1) / (l - m) * (2 * l + 1) / (l + m)) * vdm[m, l - 1, :]
                            - np.sqrt((l + m - 1) / (l - m) * (2 * l + 1) / (2 * l - 3) * (l - m - 1) / (l + m)) * vdm[m, l - 2, :])
    vdm = vdm[:mmax, :lmax].copy()
    vdm[1::2] *= -1.0
    return vdm


def sht_tables(nlat, nlon, lmax, mmax, grid):
    """(F [2 mmax, nlon], Wf [mmax, lmax, nlat], Pinv [mmax, lmax, nlat], G [nlon, 2 mmax]) in float64."""
    if grid == "legendre-gauss":
        cost, w = legendre_gauss_weights(nlat)
    elif grid == "equiangular":
        cost, w = clenshaw_curtiss_weights(nlat)
    else:
        raise NotImplementedError(f"grid {grid!r}: only 'legendre-gauss' and 'equiangular' are on the MI355X hot path")
    theta = np.flip(np.arccos(cost))                    # colatitude, north pole first
    pct = legpoly(mmax, lmax, np.cos(theta))
    Wf = pct * w[None, None, :]
    n = np.arange(nlon)
    m = np.arange(mmax)
    ang = 2.0 * np.pi * np.outer(m, n) / nlon           # [mmax, nlon]
    F = np.empty((2 * mmax, nlon))
    F[0::2] = 2.0 * np.pi / nlon * np.cos(ang)
    F[1::2] = -2.0 * np.pi / nlon * np.sin(ang)
    c = np.where((m == 0) | (2 * m == nlon), 1.0, 2.0)  # irfft doubles the interior orders
    G = np.empty((nlon, 2 * mmax))
    G[:, 0::2] = (c[:, None] * np.cos(ang)).T
    G[:, 1::2] = (-c[:, None] * np.sin(ang)).T
    return F, Wf, pct, G


# ---- y = table . x as a strided-batched GEMM, backward = transposed table -------------------------------------------
class _TableGemm(torch.autograd.Function):
    """y[z] = sum_p op(A_p[z]) . x_p[z] (+ residual) for every batch z = (z1, z2); A is a constant table.  `spec` holds
    the GEMM shape (M, N, K), A's leading dimension / transpose flag / batch strides, the row strides and batch strides
    of x and y, the output tensor shape and optionally `passes`: element offsets (oA, oX) of the K-slices that are
    accumulated (a contraction index spread over two planes of x).  The backward pass applies the transposed slices."""

    @staticmethod
    def forward(ctx, x, table, spec, residual=None, out_dtype=torch.float32, fork=False):
        # out_dtype bfloat16 (bf16 storage, lib.set_storage): the result only feeds another GEMM of the transform chain
        # fork: also return x itself (a second use of the input, e.g. the skip connection around a spectral filter); the gradient
        # arriving on that output is added in the epilogue of the input-gradient product instead of by an autograd accumulation
        x = x.contiguous()
        y = torch.empty(spec["out_shape"], device=x.device, dtype=out_dtype)
        passes = spec.get("passes", ((0, 0),))
        assert len(passes) == 1 or out_dtype == torch.float32, "multi-pass accumulation needs an fp32 output"
        for i, (oA, oX) in enumerate(passes):
            last = i == len(passes) - 1
            _gemm_batched(table, x, y, spec["M"], spec["N"], spec["K"], spec["lda"], spec["ldx"], spec["ldy"], spec["tA"], 0,
                          spec["nb1"], spec["nb2"], spec["sA"], spec["sX"], spec["sY"], accumulate=int(i > 0), oA=oA, oB=oX,
                          residual=residual.contiguous() if (residual is not None and last) else None, sR=spec["sY"])
        ctx.spec, ctx.table, ctx.in_shape, ctx.has_res, ctx.in_dtype = spec, table, x.shape, residual is not None, x.dtype
        ctx.fork = bool(fork)
        if fork:
            assert len(passes) == 1 and not (spec["nb2"] > 1 and spec["sX"][1] == 0) and x.dtype == torch.float32
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, gy, gskip=None):
        s = ctx.spec
        gy = gy.contiguous()
        gx = torch.empty(ctx.in_shape, device=gy.device, dtype=ctx.in_dtype)      # the gradient has its tensor's storage type
        shared = s["nb2"] > 1 and s["sX"][1] == 0      # x[z1] feeds every z2: its gradient is a sum over z2
        if gskip is not None:
            gskip = gskip.contiguous().float()
        for oA, oX in s.get("passes", ((0, 0),)):
            if not shared:
                _gemm_batched(ctx.table, gy, gx, s["K"], s["N"], s["M"], s["lda"], s["ldy"], s["ldx"], 1 - s["tA"], 0,
                              s["nb1"], s["nb2"], s["sA"], s["sY"], s["sX"], oA=oA, oC=oX, residual=gskip, sR=s["sX"])
                continue
            for z2 in range(s["nb2"]):
                _gemm_batched(ctx.table, gy, gx, s["K"], s["N"], s["M"], s["lda"], s["ldy"], s["ldx"], 1 - s["tA"], 0,
                              s["nb1"], 1, (s["sA"][0], 0), (s["sY"][0], 0), (s["sX"][0], 0), accumulate=int(z2 > 0),
                              oA=oA + z2 * s["sA"][1], oB=z2 * s["sY"][1], oC=oX)
        return gx, None, None, (gy if ctx.has_res else None), None, None


def _chain_dtype():
    """Storage type of the tensors that live only between two GEMMs of the transform chain (bf16 under lib.set_storage)."""
    return torch.bfloat16 if (L.storage_bf16() and L.SHADOW_ACTIVE) else torch.float32


def _table(mod, name, dtype):
    """The transform table `name` of module `mod` in the chain's storage type (a bf16 copy of the constant is made once)."""
    t = getattr(mod, name)
    if dtype == torch.float32:
        return t
    key = "_bf16_" + name
    c = mod.__dict__.get(key)
    if c is None or c.device != t.device:
        c = t.to(torch.bfloat16)
        mod.__dict__[key] = c
    return c


def _triangular_host(table):
    """1 when the host Legendre table [mmax, nlat, lmax] is exactly zero for degrees l < m, else 0 (see _triangular_flag)."""
    import numpy as np
    M, _, Lm = table.shape
    dead = np.arange(Lm)[None, :] < np.arange(M)[:, None]           # [M, L]: l < m
    return 1 if bool((np.transpose(table, (0, 2, 1))[dead] == 0).all()) else 0


def _triangular_flag(mod, name):
    """DLWP_SHT_TRIANGULAR for the bf16 synthesis kernel when the Legendre table `name` ([mmax, nlat, lmax], the kernel's S1t) is
    exactly zero for degrees l < m (associated Legendre functions: always, checked once per table on the host): every product with
    a spectrum entry of order m > l is then zero whatever the entry holds, so the kernel does not read those entries."""
    key = "_tri_" + name
    flag = mod.__dict__.get(key)         # set by RealSHT / InverseRealSHT.__init__ from the host tables (no device sync, capture-safe)
    if flag is None:                     # a module that did not go through __init__ (unpickled): one blocking check, then cached
        t = getattr(mod, name)                                   # [mmax, nlat, lmax]
        M, _, Lm = t.shape
        dead = torch.arange(Lm, device=t.device)[None, :] < torch.arange(M, device=t.device)[:, None]      # [M, L]: l < m
        flag = 1 if bool((t.permute(0, 2, 1)[dead] == 0).all()) else 0
        mod.__dict__[key] = flag
    return flag


def _fused_ok(nlat, nlon, C, mmax, lmax):
    """Single-launch kernels (csrc/sht_fused.hip, exact-f32 MFMA) or two strided-batched GEMMs?  Measured on the C3 step
    (B=4): with fp32 GEMM operands the fused transforms win (526 vs 512 samples/s); with bf16 GEMM operands the two table
    GEMMs run on the 16x faster bf16 MFMA and the fused fp32 kernels only tie at B=4 (656 vs 658) and lose at B=16 (1097 vs
    1180).  Hence: fused in fp32 mode, GEMMs in bf16 mode; DLWP_SHT_FUSED=0/1 overrides (A/B runs)."""
    lib = L.load()
    if not lib.dlwp_sht_fused_supported(nlat, nlon, C, mmax, lmax):
        return False
    env = os.environ.get("DLWP_SHT_FUSED")
    if env is not None:
        return env != "0"
    return lib.dlwp_get_gemm_precision() == 0


class _FusedAnalysis(torch.autograd.Function):
    """X = analysis(x; A1, A2) in one launch (csrc/sht_fused.hip); backward = synthesis with the transposed tables."""

    @staticmethod
    def forward(ctx, x, A1, A2, A2t, A1t, mmax, lmax):
        x = x.contiguous()
        B, K, N, C = x.shape
        X = torch.empty(lmax, B, mmax, 2, C, device=x.device)
        L.check(L.load().dlwp_sht_analysis(L.ptr(x), L.ptr(A1), L.ptr(A2), L.ptr(X), B, K, N, C, mmax, lmax, L.stream()))
        ctx.tabs, ctx.dims = (A2t, A1t), (B, K, N, C, mmax, lmax)
        return X

    @staticmethod
    def backward(ctx, gX):
        B, K, N, C, mmax, lmax = ctx.dims
        gX = gX.contiguous()
        gx = torch.empty(B, K, N, C, device=gX.device)
        L.check(L.load().dlwp_sht_synthesis(L.ptr(gX), L.ptr(ctx.tabs[0]), L.ptr(ctx.tabs[1]), L.ptr(gx), B, K, N, C, mmax,
                                            lmax, L.stream()))
        return gx, None, None, None, None, None, None


class _FusedSynthesis(torch.autograd.Function):
    """x = synthesis(X; S1, S2) in one launch; backward = analysis(A1 = S2, A2 = S1)."""

    @staticmethod
    def forward(ctx, X, S1t, S2t, S2, S1, nlat, nlon):
        X = X.contiguous()
        Lm, B, M, _, C = X.shape
        x = torch.empty(B, nlat, nlon, C, device=X.device)
        L.check(L.load().dlwp_sht_synthesis(L.ptr(X), L.ptr(S1t), L.ptr(S2t), L.ptr(x), B, nlat, nlon, C, M, Lm, L.stream()))
        ctx.tabs, ctx.dims = (S2, S1), (B, nlat, nlon, C, M, Lm)
        return x

    @staticmethod
    def backward(ctx, gx):
        B, K, N, C, M, Lm = ctx.dims
        gx = gx.contiguous()
        gX = torch.empty(Lm, B, M, 2, C, device=gx.device)
        L.check(L.load().dlwp_sht_analysis(L.ptr(gx), L.ptr(ctx.tabs[0]), L.ptr(ctx.tabs[1]), L.ptr(gX), B, K, N, C, M, Lm,
                                           L.stream()))
        return gX, None, None, None, None, None, None


def _bf16_fused_ok(nlat, nlon, C, mmax, lmax):
    """One-launch bf16 kernels (csrc/sht_bf16.hip) for the bf16-storage chain: x fp32 -> X bf16 and back.  C3 step at B = 4 with the
    two table GEMMs per transform: 3.25 ms; DLWP_SHT_BF16=0 keeps the GEMMs (A/B runs)."""
    if _chain_dtype() != torch.bfloat16 or os.environ.get("DLWP_SHT_BF16", "1") == "0":
        return False
    return L.load().dlwp_sht_bf16_supported(nlat, nlon, C, mmax, lmax) == 1


class _Bf16Analysis(torch.autograd.Function):
    """X (bf16) = analysis(x fp32; A1, A2) in one launch; backward = bf16 synthesis with the transposed tables, the skip gradient
    of a forked input (fork=True returns (X, x)) added in its output stage."""

    @staticmethod
    def forward(ctx, x, mod, names, dims, fork):
        x = x.contiguous().float()
        B, K, N, C = x.shape
        M, Lm = dims
        bf = torch.bfloat16
        X = torch.empty(Lm, B, M, 2, C, device=x.device, dtype=bf)
        L.check(L.load().dlwp_sht_analysis_bf16(L.ptr(x), L.ptr(_table(mod, names[0], bf)), L.ptr(_table(mod, names[1], bf)), L.ptr(X),
                                                B, K, N, C, M, Lm, L.stream()))
        ctx.mod, ctx.names, ctx.shape, ctx.dims, ctx.fork = mod, names, (B, K, N, C), dims, bool(fork)
        if fork:
            return X, x.view_as(x)
        return X

    @staticmethod
    def backward(ctx, gX, gskip=None):
        B, K, N, C = ctx.shape
        M, Lm = ctx.dims
        bf = torch.bfloat16
        gX = gX.contiguous()
        if gX.dtype != bf:
            gX = gX.to(bf)
        if gskip is not None:
            gskip = gskip.contiguous().float()
        gx = torch.empty(B, K, N, C, device=gX.device)
        L.check(L.load().dlwp_sht_synthesis_bf16_ex(L.ptr(gX), L.ptr(_table(ctx.mod, ctx.names[2], bf)), L.ptr(_table(ctx.mod, ctx.names[3], bf)),
                                                    L.ptr(gskip), L.ptr(gx), B, K, N, C, M, Lm, _triangular_flag(ctx.mod, ctx.names[2]),
                                                    L.stream()))
        return gx, None, None, None, None


class _Bf16Synthesis(torch.autograd.Function):
    """x = synthesis(X bf16; S1t, S2) in one launch; backward = bf16 analysis with the transposed tables.  field_bf16: x is written
    as a bf16 array (dlwp_sht_synthesis_bf16_ex, DLWP_SHT_FIELD_BF16) for a consumer that reads it once as a bf16 operand (the
    SFNO block tail); its gradient then arrives as a bf16 array too and is read as such."""

    @staticmethod
    def forward(ctx, X, mod, names, grid, field_bf16=False):
        bf = torch.bfloat16
        X = X.contiguous()
        if X.dtype != bf:
            X = X.to(bf)
        Lm, B, M, _, C = X.shape
        K, N = grid
        x = torch.empty(B, K, N, C, device=X.device, dtype=bf if field_bf16 else torch.float32)
        L.check(L.load().dlwp_sht_synthesis_bf16_ex(L.ptr(X), L.ptr(_table(mod, names[0], bf)), L.ptr(_table(mod, names[1], bf)), None,
                                                    L.ptr(x), B, K, N, C, M, Lm, _triangular_flag(mod, names[0]) | (2 if field_bf16 else 0),
                                                    L.stream()))
        ctx.mod, ctx.names, ctx.dims = mod, names, (B, K, N, C, M, Lm)
        return x

    @staticmethod
    def backward(ctx, gx):
        B, K, N, C, M, Lm = ctx.dims
        bf = torch.bfloat16
        gx = gx.contiguous()
        if gx.dtype != bf:
            gx = gx.float()
        gX = torch.empty(Lm, B, M, 2, C, device=gx.device, dtype=bf)
        L.check(L.load().dlwp_sht_analysis_bf16_ex(L.ptr(gx), L.ptr(_table(ctx.mod, ctx.names[2], bf)), L.ptr(_table(ctx.mod, ctx.names[3], bf)),
                                                   L.ptr(gX), B, K, N, C, M, Lm, 2 if gx.dtype == bf else 0, L.stream()))
        return gX, None, None, None, None


class RealSHT(nn.Module):
    """x [B, nlat, nlon, C] -> X [lmax, B, mmax, 2, C] (re, im planes).  One fused launch where the shape fits
    (dlwp_sht_fused_supported), two strided-batched GEMMs otherwise; `fused=False` forces the GEMM path."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="legendre-gauss", fused=True):
        super().__init__()
        self.nlat, self.nlon, self.grid, self.fused = nlat, nlon, grid, fused
        self.lmax = lmax or nlat
        self.mmax = mmax or nlon // 2 + 1
        F, Wf, _, _ = sht_tables(nlat, nlon, self.lmax, self.mmax, grid)
        self.register_buffer("dft", torch.from_numpy(F).float().contiguous(), persistent=False)
        self.register_buffer("weights", torch.from_numpy(Wf).float().contiguous(), persistent=False)
        # transposed copies for the fused backward (synthesis reads S1t [m][k][l], S2t [n][q])
        self.register_buffer("dft_t", torch.from_numpy(F.T.copy()).float().contiguous(), persistent=False)
        self.register_buffer("weights_t", torch.from_numpy(Wf.transpose(0, 2, 1).copy()).float().contiguous(), persistent=False)
        self._tri_weights_t = _triangular_host(Wf.transpose(0, 2, 1))       # next to the (non-persistent, never reloaded) table it describes

    def forward(self, x, fork=False):
        """fork=True: returns (X, x) -- x again, for a skip connection around the spectral filter; on the GEMM path the gradient
        coming back along it is added in the epilogue of the transform's input-gradient product."""
        B, K, N, C = x.shape
        assert K == self.nlat and N == self.nlon, "input grid does not match the transform"
        M, Lm = self.mmax, self.lmax
        if self.fused and x.is_cuda and x.dtype == torch.float32 and _bf16_fused_ok(K, N, C, M, Lm):
            # forward tables A1 = dft, A2 = weights; backward (a synthesis) S1t = weights_t, S2 = dft_t
            do_fork = bool(fork and x.requires_grad)
            res = _Bf16Analysis.apply(x, self, ("dft", "weights", "weights_t", "dft_t"), (M, Lm), do_fork)
            if do_fork:
                return res                                   # (X, x again: its gradient joins inside the backward synthesis)
            return (res, x) if fork else res
        if self.fused and x.is_cuda and _fused_ok(K, N, C, M, Lm):
            X = _FusedAnalysis.apply(x, self.dft, self.weights, self.weights_t, self.dft_t, M, Lm)
            return (X, x) if fork else X
        lon = dict(M=2 * M, N=C, K=N, lda=N, tA=0, ldx=C, ldy=C, nb1=B * K, nb2=1, sA=(0, 0), sX=(N * C, 0),
                   sY=(2 * M * C, 0), out_shape=(B, K, M, 2, C))
        cd = _chain_dtype()
        skip = None
        if fork and x.dtype == torch.float32 and x.requires_grad:
            t, skip = _TableGemm.apply(x, _table(self, "dft", cd), lon, None, cd, True)
        else:
            t = _TableGemm.apply(x, _table(self, "dft", cd), lon, None, cd)
        leg = dict(M=Lm, N=2 * C, K=K, lda=K, tA=0, ldx=2 * M * C, ldy=B * M * 2 * C, nb1=B, nb2=M, sA=(0, Lm * K),
                   sX=(K * 2 * M * C, 2 * C), sY=(M * 2 * C, 2 * C), out_shape=(Lm, B, M, 2, C))
        X = _TableGemm.apply(t, _table(self, "weights", cd), leg, None, cd)
        return (X, skip if skip is not None else x) if fork else X


class InverseRealSHT(nn.Module):
    """X [lmax, B, mmax, 2, C] -> x [B, nlat, nlon, C].  Fused / GEMM paths as in RealSHT."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="legendre-gauss", fused=True):
        super().__init__()
        self.nlat, self.nlon, self.grid, self.fused = nlat, nlon, grid, fused
        self.lmax = lmax or nlat
        self.mmax = mmax or nlon // 2 + 1
        _, _, P, G = sht_tables(nlat, nlon, self.lmax, self.mmax, grid)
        self.register_buffer("pct", torch.from_numpy(P).float().contiguous(), persistent=False)
        self.register_buffer("idft", torch.from_numpy(G).float().contiguous(), persistent=False)
        self.register_buffer("pct_t", torch.from_numpy(P.transpose(0, 2, 1).copy()).float().contiguous(), persistent=False)
        self._tri_pct_t = _triangular_host(P.transpose(0, 2, 1))
        self.register_buffer("idft_t", torch.from_numpy(G.T.copy()).float().contiguous(), persistent=False)

    def forward(self, X, field_bf16=False):
        """field_bf16: where the one-launch bf16 kernel runs, return the field as a bf16 array (for a consumer that rounds it to
        bf16 anyway: the SFNO block tail); the other paths ignore the request and return fp32."""
        Lm, B, M, _, C = X.shape
        assert Lm == self.lmax and M == self.mmax, "spectrum does not match the transform"
        K, N = self.nlat, self.nlon
        if self.fused and X.is_cuda and X.dtype == torch.bfloat16 and _bf16_fused_ok(K, N, C, M, Lm):
            # forward tables S1t = pct_t, S2 = idft; backward (an analysis) A1 = idft_t, A2 = pct
            return _Bf16Synthesis.apply(X, self, ("pct_t", "idft", "idft_t", "pct"), (K, N), bool(field_bf16))
        if self.fused and X.is_cuda and _fused_ok(K, N, C, M, Lm):
            return _FusedSynthesis.apply(X, self.pct_t, self.idft, self.idft_t, self.pct, K, N)
        leg = dict(M=K, N=2 * C, K=Lm, lda=K, tA=1, ldx=B * M * 2 * C, ldy=2 * M * C, nb1=B, nb2=M, sA=(0, Lm * K),
                   sX=(M * 2 * C, 2 * C), sY=(K * 2 * M * C, 2 * C), out_shape=(B, K, M, 2, C))
        cd = _chain_dtype()
        t = _TableGemm.apply(X, _table(self, "pct", cd), leg, None, cd)
        lon = dict(M=N, N=C, K=2 * M, lda=2 * M, tA=0, ldx=C, ldy=C, nb1=B * K, nb2=1, sA=(0, 0), sX=(2 * M * C, 0),
                   sY=(N * C, 0), out_shape=(B, K, N, C))
        return _TableGemm.apply(t, _table(self, "idft", cd), lon, None)       # back on the grid: fp32 (residual stream)


# ---- per-degree complex weights ---------------------------------------------------------------------------------
_wexp_scope = None          # {id(w): expanded image} while a spectral_weight_scope is open
_pending_folds = {}         # {id(w): (accumulated expanded gradient, gradient slot, shape)} during one backward pass


class spectral_weight_scope:
    """Within the scope the parameters cannot change (one rollout forward: every lead time applies the same weights), so
    the [[Wr, Wi], [-Wi, Wr]] image of each spectral weight is built once and shared by all net calls."""

    def __enter__(self):
        global _wexp_scope
        self._outer = _wexp_scope
        if _wexp_scope is None:
            _wexp_scope = {}
        return self

    def __exit__(self, *exc):
        global _wexp_scope
        _wexp_scope = self._outer
        return False


def _expanded_weight(w):
    lib = L.load()
    if _wexp_scope is not None and id(w) in _wexp_scope:
        return _wexp_scope[id(w)][0]
    Cin, Cout, Lm, _ = w.shape
    wexp = torch.empty(Lm, 2 * Cin, 2 * Cout, device=w.device)
    L.check(lib.dlwp_cweight_expand(L.ptr(w.contiguous()), L.ptr(wexp), Cin, Cout, Lm, L.stream()))
    if _chain_dtype() == torch.bfloat16:         # the GEMMs read a bf16 image (built once per rollout inside the scope)
        w16 = torch.empty_like(wexp, dtype=torch.bfloat16)
        L.check(lib.dlwp_cast_bf16(L.ptr(wexp), L.ptr(w16), wexp.numel(), L.stream()))
        wexp = w16
    if _wexp_scope is not None:
        _wexp_scope[id(w)] = (wexp, w)          # keeps w alive, so the id stays unique inside the scope
    return wexp


def _fold_pending():
    """End of a backward pass: fold every accumulated expanded gradient into its complex parameter's gradient slot."""
    lib = L.load()
    for gexp, slot, shape in _pending_folds.values():
        L.check(lib.dlwp_cweight_fold(L.ptr(gexp), L.ptr(slot), shape[0], shape[1], shape[2], L.stream()))
    _pending_folds.clear()


class _DHConvFn(torch.autograd.Function):
    """Y[l, r, (re|im, o)] = sum_i X[l, r, (re|im, i)] * W[i, o, l] (complex), r = (b, m) rows."""

    @staticmethod
    def forward(ctx, X, w):
        Lm, B, M, _, Cin = X.shape
        Cout = w.shape[1]
        assert w.shape == (Cin, Cout, Lm, 2), f"weight {tuple(w.shape)} does not match spectrum {tuple(X.shape)}"
        X = X.contiguous()
        wexp = _expanded_weight(w)
        R = B * M
        Y = torch.empty(Lm, B, M, 2, Cout, device=X.device, dtype=X.dtype)        # bf16 spectra stay bf16
        _gemm_batched(X, wexp, Y, R, 2 * Cout, 2 * Cin, 2 * Cin, 2 * Cout, 2 * Cout, 0, 0, Lm, 1, (R * 2 * Cin, 0),
                      (4 * Cin * Cout, 0), (R * 2 * Cout, 0))
        ctx.save_for_backward(X, wexp)
        ctx.wslot, ctx.wshape, ctx.wid = _grad_slot(w), w.shape, id(w)
        return Y

    @staticmethod
    def backward(ctx, gY):
        lib = L.load()
        X, wexp = ctx.saved_tensors
        Lm, B, M, _, Cin = X.shape
        Cout = ctx.wshape[1]
        R = B * M
        gY = gY.contiguous()
        gX = torch.empty_like(X)
        # gX[l] = gY[l] . wexp[l]^T
        _gemm_batched(gY, wexp, gX, R, 2 * Cin, 2 * Cout, 2 * Cout, 2 * Cout, 2 * Cin, 0, 1, Lm, 1, (R * 2 * Cout, 0),
                      (4 * Cin * Cout, 0), (R * 2 * Cin, 0))
        # gexp[l] = X[l]^T . gY[l].  With a preallocated gradient slot the expanded gradients of every application of
        # this weight in the backward pass (one per lead time) are summed by the GEMM itself and folded into the
        # complex parameter once, when the pass ends.
        first = ctx.wid not in _pending_folds
        if ctx.wslot is not None:
            if first:
                if not _pending_folds:
                    torch.autograd.Variable._execution_engine.queue_callback(_fold_pending)
                _pending_folds[ctx.wid] = (torch.empty_like(wexp, dtype=torch.float32), ctx.wslot, (Cin, Cout, Lm))
            gexp = _pending_folds[ctx.wid][0]
        else:
            gexp = torch.empty_like(wexp, dtype=torch.float32)
        _gemm_batched(X, gY, gexp, 2 * Cin, 2 * Cout, R, 2 * Cin, 2 * Cout, 2 * Cout, 1, 0, Lm, 1, (R * 2 * Cin, 0),
                      (R * 2 * Cout, 0), (4 * Cin * Cout, 0), accumulate=int(ctx.wslot is not None and not first))
        if ctx.wslot is not None:
            return gX, None
        gw = torch.zeros(ctx.wshape, device=X.device)
        L.check(lib.dlwp_cweight_fold(L.ptr(gexp), L.ptr(gw), Cin, Cout, Lm, L.stream()))
        return gX, gw


DHCONV_NATIVE = os.environ.get("DLWP_DHCONV_NATIVE", "1") != "0"      # env: A/B runs against the expanded-image GEMMs


def _dh_state(w):
    """Per rollout pass (spectral_weight_scope): the two fragment-order images of a spectral weight and the list that collects
    the lead times' (X, gY) pairs for the one weight-gradient product of the pass."""
    key = ("dhconv", id(w))
    if _wexp_scope is not None and key in _wexp_scope:
        return _wexp_scope[key][0]
    lib = L.load()
    Cin, Cout, Lm, _ = w.shape
    n = lib.dlwp_dhconv_image_elems(Cin, Cout, Lm)
    imgs = torch.empty(2, n, device=w.device, dtype=torch.bfloat16)
    L.check(lib.dlwp_dhconv_pack(L.ptr(w.detach().contiguous()), L.ptr(imgs[0]), L.ptr(imgs[1]), Cin, Cout, Lm, L.stream()))
    st = {"imgs": imgs, "uses": 0, "pending": []}
    if _wexp_scope is not None:
        _wexp_scope[key] = (st, w)
    return st


def prepack_spectral_weights(weights):
    """Inside a spectral_weight_scope: build the images of every spectral weight of a network (same shape) that has none yet in
    ONE launch (dlwp_dhconv_pack_many) instead of one per layer at its first use."""
    import ctypes as C
    if _wexp_scope is None or not weights or not DHCONV_NATIVE or _chain_dtype() != torch.bfloat16:
        return
    todo = [w for w in weights if ("dhconv", id(w)) not in _wexp_scope]
    if not todo or any(w.shape != todo[0].shape or not w.is_cuda for w in todo):
        return
    lib = L.load()
    Cin, Cout, Lm, _ = todo[0].shape
    if lib.dlwp_dhconv_supported(Cin, Cout, Lm) != 1 or len(todo) > 16:
        return
    n = lib.dlwp_dhconv_image_elems(Cin, Cout, Lm)
    imgs = torch.empty(len(todo), 2, n, device=todo[0].device, dtype=torch.bfloat16)
    srcs = [w.detach().contiguous() for w in todo]
    wp = (C.c_void_p * len(todo))(*[L.ptr(t) for t in srcs])
    fp = (C.c_void_p * len(todo))(*[L.ptr(imgs[i, 0]) for i in range(len(todo))])
    bp = (C.c_void_p * len(todo))(*[L.ptr(imgs[i, 1]) for i in range(len(todo))])
    L.check(lib.dlwp_dhconv_pack_many(wp, fp, bp, len(todo), Cin, Cout, Lm, L.stream()))
    for i, w in enumerate(todo):
        _wexp_scope[("dhconv", id(w))] = ({"imgs": imgs[i], "uses": 0, "pending": []}, w)


class _DHConvNativeFn(torch.autograd.Function):
    """dhconv on csrc/dhconv.hip (bf16 spectra): forward and input gradient one launch each on the un-expanded weight images; the
    weight gradient of ALL lead times of a pass as one product per degree, launched by the last backward pass through the weight
    and folded straight into the parameter's gradient."""

    @staticmethod
    def applies(X, w):
        return (DHCONV_NATIVE and X.is_cuda and X.dtype == torch.bfloat16 and _chain_dtype() == torch.bfloat16
                and L.load().dlwp_dhconv_supported(w.shape[0], w.shape[1], w.shape[2]) == 1)

    @staticmethod
    def forward(ctx, X, w, triangular):
        Lm, B, M, _, Cin = X.shape
        Cout = w.shape[1]
        assert w.shape == (Cin, Cout, Lm, 2), f"weight {tuple(w.shape)} does not match spectrum {tuple(X.shape)}"
        X = X.contiguous()
        st = _dh_state(w)
        Y = torch.empty(Lm, B, M, 2, Cout, device=X.device, dtype=torch.bfloat16)
        ctx.mm = M if triangular else 0                     # orders m > l known to be zero: skipped (and produced as zeros)
        L.check(L.load().dlwp_dhconv_apply(L.ptr(X), L.ptr(st["imgs"][0]), L.ptr(Y), Lm, B * M * 2, Cin, Cout, ctx.mm, 0, L.stream()))
        ctx.save_for_backward(X)
        if any(ctx.needs_input_grad):
            st["uses"] += 1
        ctx.st, ctx.wslot, ctx.wshape = st, _grad_slot(w), w.shape
        return Y

    @staticmethod
    def backward(ctx, gY):
        import ctypes as C
        lib = L.load()
        (X,) = ctx.saved_tensors
        Lm, B, M, _, Cin = X.shape
        Cout = ctx.wshape[1]
        st = ctx.st
        gY = gY.contiguous()
        if gY.dtype != torch.bfloat16:
            gY = gY.to(torch.bfloat16)
        gX = torch.empty_like(X)
        L.check(lib.dlwp_dhconv_apply(L.ptr(gY), L.ptr(st["imgs"][1]), L.ptr(gX), Lm, B * M * 2, Cout, Cin, ctx.mm, 1, L.stream()))
        st["pending"].append((X, gY))
        st["uses"] -= 1
        if st["uses"] > 0 and len(st["pending"]) < 8:
            return gX, None, None
        segs, st["pending"] = st["pending"], []
        xs = (C.c_void_p * len(segs))(*[L.ptr(a) for a, _ in segs])
        gs = (C.c_void_p * len(segs))(*[L.ptr(b) for _, b in segs])
        G = torch.empty(Lm, Cin, 2 * Cout, device=X.device)
        L.check(lib.dlwp_dhconv_wgrad(xs, gs, len(segs), L.ptr(G), Lm, B * M * 2, Cin, Cout, ctx.mm, L.stream()))
        gw = ctx.wslot if ctx.wslot is not None else torch.zeros(ctx.wshape, device=X.device)
        L.check(lib.dlwp_dhconv_fold(L.ptr(G), L.ptr(gw), Cin, Cout, Lm, L.stream()))
        return gX, (None if ctx.wslot is not None else gw), None


def dhconv(X, w, triangular=False):
    """Y[l, b, m, :, o] = sum_i X[l, b, m, :, i] * w[i, o, l] (complex).  triangular=True: the caller guarantees X[l, :, m] = 0 for
    m > l (a spectrum produced by RealSHT; likewise its gradient) -- the bf16 kernels then skip those orders."""
    if _DHConvNativeFn.applies(X, w):
        return _DHConvNativeFn.apply(X, w, bool(triangular))
    return _DHConvFn.apply(X, w)

import sys, torch
sys.path.insert(0, ".")
from dlwp_benchmark_amd.token_ops import InstanceNorm
dev = torch.device("cuda:0")
for (B, H, W, C) in [(1, 7, 9, 5), (1, 7, 9, 8), (1, 8, 8, 5), (2, 7, 9, 5), (1, 7, 9, 64)]:
    g = torch.Generator().manual_seed(8)
    x = (torch.randn(B, H, W, C, generator=g) * 2 + 3).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(C, generator=g).requires_grad_(True)
    gy = torch.randn(B, H, W, C, generator=g)
    yr = torch.nn.functional.instance_norm(x.permute(0, 3, 1, 2), weight=gamma, bias=beta, eps=1e-6).permute(0, 2, 3, 1)
    yr.backward(gy)
    m = InstanceNorm(C, eps=1e-6).to(dev)
    with torch.no_grad():
        m.weight.copy_(gamma); m.bias.copy_(beta)
    xd = x.detach().to(dev).requires_grad_(True)
    y = m(xd)
    y.backward(gy.to(dev))
    # manual formula on CPU
    xx = x.detach().double(); gg = gy.double()
    mu = xx.mean(dim=(1, 2), keepdim=True); var = xx.var(dim=(1, 2), unbiased=False, keepdim=True)
    rs = (var + 1e-6).rsqrt(); xh = (xx - mu) * rs
    man = rs * gamma.detach().double() * (gg - gg.mean(dim=(1, 2), keepdim=True) - xh * (gg * xh).mean(dim=(1, 2), keepdim=True))
    print((B, H, W, C), "y", (y.cpu() - yr).abs().max().item(), "gx vs torch", (xd.grad.cpu() - x.grad).abs().max().item(),
          "gx vs manual", (xd.grad.cpu().double() - man).abs().max().item(), "torch vs manual", (x.grad.double() - man).abs().max().item())

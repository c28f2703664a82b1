import sys, torch
sys.path.insert(0, ".")
from dlwp_benchmark_amd import dlwpbench
from dlwp_benchmark_amd.train_engine import GraphedTrainStep, mse_loss
dev = torch.device("cuda:0")
def make():
    torch.manual_seed(0)
    m = _make()
    for mod in m.modules():
        if isinstance(mod, dlwpbench.panguweather.DropPath) and DP is not None:
            mod.p = DP
    return m
DP = None if len(sys.argv) > 1 and sys.argv[1] == "droppath" else 0.0
def _make():
    return dlwpbench.PanguWeather(constant_channels=4, prescribed_channels=1, prognostic_channels=5, embed_dim=192,
                                  num_heads=(6, 12, 12, 6), window_size=(2, 6, 12), patch_size=(1, 1), n_lat=32, n_lon=64,
                                  context_size=1, drop_path_rate=0.0).to(dev).train()
g = torch.Generator().manual_seed(1234)
kw = dict(constants=torch.randn(1, 1, 4, 32, 64, generator=g).to(dev), prescribed=torch.randn(1, 5, 1, 32, 64, generator=g).to(dev),
          prognostic=torch.randn(1, 5, 5, 32, 64, generator=g).to(dev))
tgt = torch.randn(1, 4, 5, 32, 64, generator=g).to(dev)
ref = make()
opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
L0 = []
for _ in range(40):
    opt.zero_grad(set_to_none=True)
    loss = mse_loss(ref(**kw), tgt); loss.backward(); opt.step(); L0.append(round(loss.item(), 4))
print("eager/unfused", L0)
for ug in (False, True):
    m = make()
    st = GraphedTrainStep(m, kw, tgt, lr=1e-3, use_graph=ug)
    print("flat fused graph=%d" % ug, [round(st().item(), 4) for _ in range(40)])

import sys, time
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch
from dlwp_benchmark_amd import lib as L, nsbench
lib = L.load()
out = torch.zeros(4, dtype=torch.int64, device='cuda')
def probe(tag, iters=2000, blocks=256):
    L.check(lib.dlwp_debug_clock_probe(out.data_ptr(), iters, blocks, L.stream())); torch.cuda.synchronize()
    t, r = out[0].item(), out[1].item()
    print(f"{tag}: shader ticks {t}, real {r*10} ns -> {t/(r*10)*1000:.0f} MHz  ({t/iters:.2f} ticks/iter)")
probe("cold")
probe("again")
probe("long", iters=200000)
m = nsbench.TFNO2DModule(n_modes=[12,12], in_channels=1, hidden_channels=32, lifting_channels=256, projection_channels=256, out_channels=1, n_layers=4, context_size=10).cuda()
x = torch.randn(4,20,1,64,64, device='cuda'); opt = m.make_optimizer()
for _ in range(30): m.train_step(x, x, 10, optimizer=opt)
probe("after 30 steps (queued)")
torch.cuda.synchronize()
for _ in range(30): m.train_step(x, x, 10, optimizer=opt)
probe("after 60 steps")

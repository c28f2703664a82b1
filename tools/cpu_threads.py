import os, sys, time
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch
from oracle import fno_ref
import bench
w = bench.WORKLOAD
for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    net = fno_ref.FNO(w["n_modes"], 10, 32, 256, 256, 1, 4, seed=1234); net.requires_grad_(True)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    u = torch.randn(4, 21, 1, 64, 64); x, y = u[:, :-1].contiguous(), u[:, 1:].contiguous()
    fno_ref.train_step(net, x, y, 10, 10, optimizer=opt)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 4 and n < 20:
        fno_ref.train_step(net, x, y, 10, 10, optimizer=opt); n += 1
    dt = time.perf_counter() - t0
    print(f"threads={nt}: {4*n/dt:.2f} samples/s ({dt/n*1e3:.0f} ms/step)", flush=True)

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dlwp_benchmark_amd import dlwpbench, lib as L, token_ops as TO, window_ops as WO
w = bench.DLWP_WORKLOADS["swin"]
L.set_gemm_precision("bf16"); L.set_storage(w["storage"])
dev = torch.device("cuda", 0)
model = getattr(dlwpbench, w["cls"])(**w["model"]).to(dev).train()
B, H, W_, Cg, T = w["batch"], w["H"], w["W"], w["Cg"], w["T"]
kw = dict(constants=torch.randn(B, 1, 4, H, W_).to(dev), prescribed=torch.randn(B, T, 1, H, W_).to(dev), prognostic=torch.randn(B, T, Cg, H, W_).to(dev))
op = WO.partition
def part(x, spec, *a, **k):
    out = op(x, spec, *a, **k)
    print("partition in", x.dtype, tuple(x.shape), "out", out.dtype, tuple(out.shape), "identity", WO._identity(spec, tuple(spec.shift if not a else a[0])))
    return out
import dlwp_benchmark_amd.nsbench.swin_transformer as ST
ST.partition = part
blk = model.layers[0].blocks[0]
print("real_token_flow", getattr(blk, "real_token_flow", None), "act dtype", TO._act_dtype())
of = TO._LayerNormFn.forward
from dlwp_benchmark_amd.train_engine import GraphedTrainStep
target = torch.randn(B, T - 1, Cg, H, W_).to(dev)
step = GraphedTrainStep(model, kw, target, lr=1e-3, clip_max_norm=1e-3, use_graph=False)
print("act dtype now", TO._act_dtype())
ol = TO._LinearFn.forward
step._fwd_bwd()

"""Launch only the per-mode spectral kernel (fno_mix_fwd_kernel: H-axis step + complex contraction on MFMA) for rocprofv3."""
import sys
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch, bench
if len(sys.argv) > 1:
    bench.WORKLOAD["hidden_channels"] = int(sys.argv[1])
print(bench.mix_probe(torch.device('cuda:0'), 4, reps=50))

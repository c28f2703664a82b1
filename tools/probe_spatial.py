"""Launch only the roofline probe kernel (forward inner-block `spatial`) for PMC passes."""
import sys
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch, bench
print(bench.roofline_probe(torch.device('cuda:0'), 4, reps=50))

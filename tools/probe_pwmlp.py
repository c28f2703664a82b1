import sys
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch
from dlwp_benchmark_amd import lib as L
lib = L.load(); dev='cuda'
B,Cin,Ch,Cout,P = 4,10,256,32,4096
x=torch.randn(B,Cin,P,device=dev); w1=torch.randn(Ch,Cin,device=dev); b1=torch.randn(Ch,device=dev)
w2=torch.randn(Cout,Ch,device=dev); b2=torch.randn(Cout,device=dev); y=torch.empty(B,Cout,P,device=dev)
gy=torch.randn(B,Cout,P,device=dev); gx=torch.empty_like(x)
g=[torch.zeros_like(t) for t in (w1,b1,w2,b2)]
for it in range(10):
    L.check(lib.dlwp_pwmlp_fwd(L.ptr(x),L.ptr(w1),L.ptr(b1),L.ptr(w2),L.ptr(b2),L.ptr(y),B,Cin,Ch,Cout,P,L.stream()))
    L.check(lib.dlwp_pwmlp_bwd(L.ptr(x),L.ptr(w1),L.ptr(b1),L.ptr(w2),L.ptr(gy),L.ptr(gx),*[L.ptr(t) for t in g],B,Cin,Ch,Cout,P,L.stream()))
torch.cuda.synchronize()

import ctypes as C, sys
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch
lib = C.CDLL(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..', 'dlwp_benchmark_amd', 'libdlwpmi_stamps.so'))
V=C.c_void_p; I=C.c_int
lib.dlwp_pwmlp_fwd.argtypes=[V]*6+[I]*5+[V]; lib.dlwp_pwmlp_fwd.restype=I
lib.dlwp_debug_stamps_pwmlp.argtypes=[V]; lib.dlwp_debug_stamps_pwmlp.restype=I
dev='cuda'
for (B,Cin,Ch,Cout,P) in [(4,10,256,32,4096),(4,32,256,1,4096)]:
    x=torch.randn(B,Cin,P,device=dev); w1=torch.randn(Ch,Cin,device=dev); b1=torch.randn(Ch,device=dev)
    w2=torch.randn(Cout,Ch,device=dev); b2=torch.randn(Cout,device=dev); y=torch.empty(B,Cout,P,device=dev)
    for it in range(3):
        rc=lib.dlwp_pwmlp_fwd(x.data_ptr(),w1.data_ptr(),b1.data_ptr(),w2.data_ptr(),b2.data_ptr(),y.data_ptr(),B,Cin,Ch,Cout,P,None)
        torch.cuda.synchronize()
    buf=(C.c_ulonglong*32)()
    lib.dlwp_debug_stamps_pwmlp(buf)
    t=list(buf)[:9]
    print((B,Cin,Ch,Cout,P), "phase cycles:", [t[i+1]-t[i] for i in range(8)], "total", t[8]-t[0])
lib.dlwp_pwmlp_bwd.argtypes=[V]*10+[I]*5+[V]; lib.dlwp_pwmlp_bwd.restype=I
for (B,Cin,Ch,Cout,P) in [(4,10,256,32,4096),(4,32,256,1,4096)]:
    x=torch.randn(B,Cin,P,device=dev); w1=torch.randn(Ch,Cin,device=dev); b1=torch.randn(Ch,device=dev)
    w2=torch.randn(Cout,Ch,device=dev); gy=torch.randn(B,Cout,P,device=dev); gx=torch.empty_like(x)
    g=[torch.zeros_like(t) for t in (w1,b1,w2,torch.zeros(Cout,device=dev))]
    for it in range(3):
        rc=lib.dlwp_pwmlp_bwd(x.data_ptr(),w1.data_ptr(),b1.data_ptr(),w2.data_ptr(),gy.data_ptr(),gx.data_ptr(),*[t.data_ptr() for t in g],B,Cin,Ch,Cout,P,None)
        torch.cuda.synchronize()
    buf=(C.c_ulonglong*32)()
    lib.dlwp_debug_stamps_pwmlp(buf)
    t=list(buf)
    print("bwd",(B,Cin,Ch,Cout,P), "stamps 10..22 deltas:", [t[i+1]-t[i] for i in range(10,22)], "total", t[22]-t[10])

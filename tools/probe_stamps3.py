import ctypes as C, sys
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch
lib = C.CDLL(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..', 'dlwp_benchmark_amd', 'libdlwpmi_stamps.so'))
V=C.c_void_p; I=C.c_int
lib.dlwp_pwmlp_slab_floats.argtypes=[I]*5; lib.dlwp_pwmlp_slab_floats.restype=C.c_longlong
lib.dlwp_pwmlp_bwd_slab.argtypes=[V]*7+[I]*6+[V]
lib.dlwp_debug_stamps_pwmlp.argtypes=[V]
dev='cuda'
for (B,Cin,Ch,Cout,P) in [(4,10,256,32,4096),(4,32,256,1,4096)]:
    x=torch.randn(B,Cin,P,device=dev); w1=torch.randn(Ch,Cin,device=dev); b1=torch.randn(Ch,device=dev)
    w2=torch.randn(Cout,Ch,device=dev); gy=torch.randn(B,Cout,P,device=dev); gx=torch.empty_like(x)
    slab=torch.zeros(lib.dlwp_pwmlp_slab_floats(B,Cin,Ch,Cout,P),device=dev)
    for acc in (0,1):
        for it in range(3):
            lib.dlwp_pwmlp_bwd_slab(x.data_ptr(),w1.data_ptr(),b1.data_ptr(),w2.data_ptr(),gy.data_ptr(),gx.data_ptr(),slab.data_ptr(),acc,B,Cin,Ch,Cout,P,None)
            torch.cuda.synchronize()
        buf=(C.c_ulonglong*32)(); lib.dlwp_debug_stamps_pwmlp(buf); t=list(buf)
        names = {9: "entry", 10: "tiles staged+barrier", 11: "init", 12: "hb_begin(0)", 18: "4 pixel blocks", 19: "flush", 20: "main loop end (wave 0)",
                 21: "gx partials->LDS", 22: "barrier", 23: "gx reduce+store, db2, rows DFT"}
        ks = [9, 10, 11, 12, 18, 19, 20, 21, 22, 23]
        print("bwd slab acc=%d" % acc, (B, Cin, Ch, Cout, P), "cycles:", {names[k1]: t[k1] - t[k0] for k0, k1 in zip(ks, ks[1:])}, "total", t[23] - t[9],
              "wave ends (from entry):", [t[24 + w] - t[9] for w in range(8)])

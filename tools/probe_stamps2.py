import ctypes as C, sys
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch
lib = C.CDLL(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..', 'dlwp_benchmark_amd', 'libdlwpmi_stamps.so'))
V=C.c_void_p; I=C.c_int
lib.dlwp_fno_plan_create.argtypes=[I]*5+[C.POINTER(V)]
lib.dlwp_fno_block_workspace_bytes.argtypes=[V,I]; lib.dlwp_fno_block_workspace_bytes.restype=C.c_size_t
lib.dlwp_fno_block_fwd.argtypes=[V,V,I,V,V,V,V,V,I,V,V]
lib.dlwp_fno_block_bwd.argtypes=[V,V,I,V,V,V,V,V,V,V,V,I,V,V]
lib.dlwp_debug_stamps_fno.argtypes=[V]
dev='cuda'; B,Cc,H,W,m1,m2c=4,32,64,64,12,7
plan=V(); lib.dlwp_fno_plan_create(Cc,H,W,m1,m2c,C.byref(plan))
ws=torch.empty(lib.dlwp_fno_block_workspace_bytes(plan,B),dtype=torch.uint8,device=dev)
x=torch.randn(B,Cc,H,W,device=dev); w=torch.randn(m1,m2c,Cc,Cc,2,device=dev); k=torch.randn(Cc,Cc,device=dev); bb=torch.randn(Cc,device=dev)
pre=torch.empty_like(x); xhat=torch.empty(B,m1,m2c,Cc,2,device=dev); g=torch.randn_like(x); gx=torch.empty_like(x)
gw=torch.zeros_like(w); gk=torch.zeros_like(k); gb=torch.zeros_like(bb)
def stamps():
    buf=(C.c_ulonglong*32)(); lib.dlwp_debug_stamps_fno(buf); t=list(buf); return [t[i+1]-t[i] for i in range(12)], t[12]-t[0]
for it in range(3):
    lib.dlwp_fno_block_fwd(plan,x.data_ptr(),1,w.data_ptr(),k.data_ptr(),bb.data_ptr(),pre.data_ptr(),xhat.data_ptr(),B,ws.data_ptr(),None); torch.cuda.synchronize()
print("spatial fwd phases", *stamps())
buf=(C.c_ulonglong*32)(); lib.dlwp_debug_stamps_fno(buf); t=list(buf); print("mix fwd stamps 16..24 deltas", [t[i+1]-t[i] for i in range(16,24)], "total", t[24]-t[16])
for it in range(3):
    lib.dlwp_fno_block_bwd(plan,x.data_ptr(),1,w.data_ptr(),k.data_ptr(),g.data_ptr(),xhat.data_ptr(),gx.data_ptr(),gw.data_ptr(),gk.data_ptr(),gb.data_ptr(),B,ws.data_ptr(),None); torch.cuda.synchronize()
print("spatial bwd phases", *stamps())

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tstar_amd import _lib
lib=_lib.load(); s=torch.cuda.current_stream().cuda_stream
for B in (16,64,128):
    T,heads=577,12; D=768
    qkv=torch.randn(B*T,3*D,device='cuda'); out=torch.empty(B*T,D,device='cuda')
    for name,fn in (("f32",lambda: lib.tstar_attention_f32(qkv.data_ptr(),out.data_ptr(),B,T,heads,0,None,s)),("split",lambda: lib.tstar_attention_split(qkv.data_ptr(),out.data_ptr(),B,T,heads,s))):
        for _ in range(3): fn()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms=e0.elapsed_time(e1)/10
        print(f"attn {name:5s} B={B:3d} {ms:.3f} ms {4.0*B*heads*T*T*64/ms/1e9:.1f} TF")

import sys, json
sys.path.insert(0,'torch-darktable_amd'); sys.path.insert(0,'.')
import torch, torch_darktable as td
from torch_darktable import _native
from torch_darktable.synthetic import synthetic_bayer
dev=torch.device('cuda',0); w,h=4096,3072
bayer=synthetic_bayer(h,w,1234,dev)
packed=td.encode(bayer.reshape(-1))
gains=torch.tensor([1.5,1.0,1.2],device=dev)
rcd=td.RCD(dev,(w,h),td.BayerPattern.RGGB)
def t(f,name):
    for _ in range(3): f()
    torch.cuda.synchronize(); _native.profile_enable(True)
    for _ in range(10): f()
    torch.cuda.synchronize(); r=_native.profile_report(); _native.profile_enable(False)
    print(name, {k:round(v[1]/v[0]*1e3,1) for k,v in r.items()})
t(lambda: rcd.process(bayer),'plain f32')
t(lambda: rcd.process(bayer.half()),'plain f16 (incl. cast)')
t(lambda: rcd.process_packed(packed,None),'packed no gains f32')
t(lambda: rcd.process_packed(packed,gains),'packed gains f32')
t(lambda: rcd.process_packed(packed,gains,output_dtype=torch.float16),'packed gains f16')
t(lambda: rcd.process(td.apply_white_balance(td.decode12(packed).view(h,w),gains,td.BayerPattern.RGGB).unsqueeze(-1)),'three calls')

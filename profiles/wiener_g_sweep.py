import sys, os, time, json
sys.path.insert(0,'torch-darktable_amd'); sys.path.insert(0,'.')
import torch, torch_darktable as td
from torch_darktable import _native
from torch_darktable.synthetic import synthetic_bayer
dev=torch.device('cuda',0)
w,h=4096,3072
b=synthetic_bayer(h,w,1234,dev).half()
rgb=td.RCD(dev,(w,h),td.BayerPattern.RGGB).process(b)
wi=td.Wiener(dev,(w,h),overlap_factor=4,tile_size=32)
res={}
for G in [0, 26, 30, 32, 34, 36, 38, 40, 44]:
    if G: os.environ['TDK_WIENER_G']=str(G)
    for _ in range(3): wi.process_log_luminance(rgb,0.075)
    torch.cuda.synchronize()
    _native.profile_enable(True)
    for _ in range(10): wi.process_log_luminance(rgb,0.075)
    torch.cuda.synchronize()
    rep=_native.profile_report(); _native.profile_enable(False)
    res[G]={k:round(v[1]/v[0]*1e3,1) for k,v in rep.items()}
print(json.dumps(res,indent=0))

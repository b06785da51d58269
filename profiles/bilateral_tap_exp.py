import sys, json
sys.path.insert(0,'torch-darktable_amd'); sys.path.insert(0,'.')
import torch, torch_darktable as td
from torch_darktable import _native
from torch_darktable.synthetic import synthetic_rgb, synthetic_bayer
dev=torch.device('cuda',0); w,h=4096,3072
rgb=td.Wiener(dev,(w,h)).process_log_luminance(td.RCD(dev,(w,h),td.BayerPattern.RGGB).process(synthetic_bayer(h,w,1234,dev).half()),0.075)
bil=td.Bilateral(dev,(w,h),sigma_s=2.0,sigma_r=0.2)
lum=td.extension.extension._extract_luminance(rgb, False, 1e-6, torch.float32)
acc=td.tonemap.MetricsAccumulator(dev,8)
def t(f,name):
    for _ in range(3): f()
    torch.cuda.synchronize(); _native.profile_enable(True)
    for _ in range(10): f()
    torch.cuda.synchronize(); r=_native.profile_report(); _native.profile_enable(False)
    print(name, {k:round(v[1]/v[0]*1e3,1) for k,v in r.items()})
t(lambda: bil.process_rgb(rgb,0.4),'plain')
t(lambda: bil.process_rgb(rgb,0.4),'plain')
t(lambda: bil.process_rgb(rgb,0.4,luminance=lum),'lum')
t(lambda: (bil.process_rgb(rgb,0.4,metrics=acc), acc.finish()),'metrics')
t(lambda: (bil.process_rgb(rgb,0.4,luminance=lum,metrics=acc), acc.finish()),'both')

"""Experiment: clock64() deltas per phase of one workgroup of the bilateral tile kernel, from a library built with
-DTDK_EXPERIMENTS -DTDK_BIL_TIMING=1 (only bilateral.hip recompiled; the default build compiles no experiment hooks):  python profiles/bilateral_phase_exp.py variants/bil_timing.so
Slots: 8 set-up (sample window + table records), 0 its barrier, 1 splat, 2 blur x, 3 blur y, 4 z derivative, 5 slice + modify."""
import ctypes as C
import json
import sys

import torch

lib = C.CDLL(sys.argv[1])
f = lib.tdk_bilateral_rgb_lum
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_int, C.c_void_p]
lib.tdk_bilateral_rgb_workspace_bytes.restype = C.c_size_t
lib.tdk_bilateral_rgb_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float]
lib.tdk_last_error.restype = C.c_char_p
dev = torch.device('cuda', 0)
w, h = 4096, 3072
g = torch.Generator(device=dev).manual_seed(1)
rgb = (torch.rand(h, w, 3, generator=g, device=dev) * 0.8 + 0.1).half()
out = torch.empty_like(rgb)
lum = torch.rand(h, w, generator=g, device=dev) * 0.9 + 0.05
ws = torch.empty(max(lib.tdk_bilateral_rgb_workspace_bytes(w, h, 2.0, 0.2), 256), dtype=torch.uint8, device=dev)
run = lambda: f(rgb.data_ptr(), lum.data_ptr(), out.data_ptr(), ws.data_ptr(), w, h, 2.0, 0.2, 0.4, 0, 1e-6, 1, None)
for _ in range(3):
    assert run() == 0, lib.tdk_last_error()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    run()
b.record()
torch.cuda.synchronize()
res = {'us': round(a.elapsed_time(b) / 20 * 1e3, 1)}
if hasattr(lib, 'tdk_debug_bilateral_phase_cycles'):
    buf = (C.c_ulonglong * 16)()
    lib.tdk_debug_bilateral_phase_cycles(buf, 1)
    run()
    torch.cuda.synchronize()
    lib.tdk_debug_bilateral_phase_cycles(buf, 1)
    res["cycles"] = [int(buf[k]) for k in range(10)]
print(json.dumps(res))

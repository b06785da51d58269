"""Experiment: tdk_rcd device time (12 MP, fp16 and fp32 storage) of variant libraries, one process per library:
    python profiles/rcd_ab.py variants/a.so variants/b.so ..."""
import ctypes as C
import json
import subprocess
import sys


def child(path):
    import torch
    lib = C.CDLL(path)
    dev = torch.device('cuda', 0)
    w, h = 4096, 3072
    g = torch.Generator(device=dev).manual_seed(1)
    res = {}
    for name, dt, tag in (('f16', torch.float16, 1), ('f32', torch.float32, 0)):
        bayer = (torch.rand(h, w, generator=g, device=dev) * 0.9 + 0.05).to(dt)
        out = torch.empty(h, w, 3, dtype=dt, device=dev)
        run = lambda: lib.tdk_rcd(C.c_void_p(bayer.data_ptr()), C.c_void_p(out.data_ptr()), None, w, h, C.c_uint32(0x94949494), tag, None)
        for _ in range(3):
            assert run() == 0
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            run()
        b.record()
        torch.cuda.synchronize()
        res[name] = round(a.elapsed_time(b) / 20 * 1e3, 1)
    print(path, json.dumps(res), flush=True)


if __name__ == '__main__':
    if sys.argv[1] == '--child':
        child(sys.argv[2])
    else:
        for _ in range(2):
            for p in sys.argv[1:]:
                subprocess.run([sys.executable, __file__, '--child', p], stderr=subprocess.DEVNULL)

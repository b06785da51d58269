"""Experiment: where the RCD tile kernel spends its time.  Variant libraries built with -DTDK_RCD_STOP=k return
after phase k (results are wrong, only the duration matters):
    for k in 0..5: TDK_EXTRA_FLAGS="-DTDK_EXPERIMENTS -DTDK_RCD_STOP=$k" python torch-darktable_amd/build.py --force; cp .../libtdk_hip.so variants/rcd_stop$k.so
    python profiles/rcd_phase_exp.py variants/rcd_stop*.so variants/rcd_full.so
Each library is timed in its own process (12 MP, fp16 in / fp16 out and fp32 / fp32).
A library built with -DTDK_EXPERIMENTS -DTDK_RCD_TIMING=1 also reports clock64() deltas per phase of one workgroup (entries: 0 = end
barrier + load phase + prefetch issue, 1..6 = phases P1..P6), in cycles per tile."""
import ctypes as C
import json
import subprocess
import sys


def child(path):
    import torch
    lib = C.CDLL(path)
    lib.tdk_rcd.restype = C.c_int
    lib.tdk_rcd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_void_p]
    dev = torch.device('cuda', 0)
    w, h = 4096, 3072
    g = torch.Generator(device=dev).manual_seed(1)
    out = {}
    for name, dt, tag in (('f16', torch.float16, 1), ('f32', torch.float32, 0)):
        x = (torch.rand(h, w, generator=g, device=dev) * 0.9 + 0.05).to(dt)
        y = torch.empty(h, w, 3, dtype=dt, device=dev)
        run = lambda: lib.tdk_rcd(x.data_ptr(), y.data_ptr(), None, w, h, 0x94949494, tag, None)
        for _ in range(5):
            assert run() == 0
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30):
            run()
        b.record()
        torch.cuda.synchronize()
        out[name] = round(a.elapsed_time(b) / 30 * 1e3, 1)
        if hasattr(lib, 'tdk_debug_rcd_phase_cycles'):
            buf = (C.c_ulonglong * 16)()
            lib.tdk_debug_rcd_phase_cycles(buf, 1)   # reset
            run()
            torch.cuda.synchronize()
            lib.tdk_debug_rcd_phase_cycles(buf, 1)
            out[name + '_cycles_per_tile'] = [round(buf[k] / 12) for k in range(12)]   # workgroup 3 of 256 owns 12 tiles at 12 MP
    print(json.dumps(out))


if __name__ == '__main__':
    if sys.argv[1] == '--child':
        child(sys.argv[2])
    else:
        for path in sys.argv[1:]:
            r = subprocess.run([sys.executable, __file__, '--child', path], capture_output=True, text=True)
            print(path, r.stdout.strip() or r.stderr[-300:], flush=True)

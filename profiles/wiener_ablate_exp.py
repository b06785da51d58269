"""Experiment: where the y-streaming Wiener tile kernel spends its time.  Variant libraries built with
-DTDK_EXPERIMENTS -DTDK_YS_ABLATE=n leave one part out (results are wrong, only the duration matters):
  1 no row stages   2 no column pipeline   3 log-lightness conversion replaced by a sum   4 no gains   5 no column FFTs / gains
-DTDK_YS_TIMING=1 (variant 9, ys_timing.so) keeps everything and adds clock64() stamps per phase and wave of one workgroup.
    python profiles/wiener_ablate_exp.py build          # in the build container: variants/ys_ablate<n>.so + ys_full.so
    python profiles/wiener_ablate_exp.py variants/ys_*.so   # on the GPU box
Each library is timed in its own process: tdk_wiener (fp32 plane) and tdk_wiener_log_luminance (fp16 RGB) at 12 MP."""
import ctypes as C
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / 'torch-darktable_amd'


def build():
    sys.path.insert(0, str(PKG))
    import importlib.util
    spec = importlib.util.spec_from_file_location('tdk_build', PKG / 'build.py')
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build()  # default objects
    out = ROOT / 'variants'
    out.mkdir(exist_ok=True)
    others = [str(o) for o in (PKG / 'build').glob('*.o') if o.stem != 'wiener']
    for n in [0] + [int(a) for a in sys.argv[2:]] if len(sys.argv) > 2 else [0, 1, 2, 3, 4, 5, 9]:
        obj = out / f'wiener_{n}.o'
        flags = [f for f in b.CXXFLAGS] + (['-DTDK_EXPERIMENTS', f'-DTDK_YS_ABLATE={n}'] if n not in (0, 9) else []) + (['-DTDK_EXPERIMENTS', '-DTDK_YS_TIMING=1'] if n == 9 else [])
        subprocess.run([b.HIPCC, *flags, '-c', str(PKG / 'csrc' / 'wiener.hip'), '-o', str(obj)], check=True)
        lib = out / ('ys_timing.so' if n == 9 else (f'ys_ablate{n}.so' if n else 'ys_full.so'))
        subprocess.run([b.HIPCC, '-shared', '-fPIC', f'--offload-arch={b.ARCH}', '-o', str(lib), *others, str(obj)], check=True)
        obj.unlink()
        print(lib)


def child(path):
    import torch
    lib = C.CDLL(path)
    lib.tdk_wiener_workspace_bytes.restype = C.c_size_t
    lib.tdk_wiener_log_luminance_workspace_bytes.restype = C.c_size_t
    dev = torch.device('cuda', 0)
    w, h = 4096, 3072
    g = torch.Generator(device=dev).manual_seed(1)
    sig = torch.tensor([0.075], device=dev)
    plane = torch.rand(h, w, generator=g, device=dev) * 2 - 3
    rgb = (torch.rand(h, w, 3, generator=g, device=dev) * 0.8 + 0.1).half()
    out_p, out_rgb = torch.empty_like(plane), torch.empty_like(rgb)
    ws = torch.empty(max(lib.tdk_wiener_workspace_bytes(w, h, 1, 32, 4), lib.tdk_wiener_log_luminance_workspace_bytes(w, h, 32, 4)), dtype=torch.uint8, device=dev)
    V = C.c_void_p
    runs = {
        'plane_f32': lambda: lib.tdk_wiener(V(plane.data_ptr()), V(out_p.data_ptr()), V(ws.data_ptr()), w, h, 1, 32, 4, V(sig.data_ptr()), 0, None),
        'loglum_f16': lambda: lib.tdk_wiener_log_luminance(V(rgb.data_ptr()), V(out_rgb.data_ptr()), V(ws.data_ptr()), w, h, 32, 4, V(sig.data_ptr()), C.c_float(1e-4), 1, None),
    }
    res = {}
    for name, run in runs.items():
        for _ in range(3):
            assert run() == 0
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            run()
        b.record()
        torch.cuda.synchronize()
        res[name + '_us'] = round(a.elapsed_time(b) / 20 * 1e3, 1)
    if hasattr(lib, 'tdk_debug_ys_phase_cycles'):
        buf = (C.c_ulonglong * 32)()
        lib.tdk_debug_ys_phase_cycles(buf, 1)  # reset
        runs['plane_f32']()
        torch.cuda.synchronize()
        lib.tdk_debug_ys_phase_cycles(buf, 1)
        names = ['loads', 'column', 'fwd_row', 'inv_row', 'staging', 'barrier', 'empty']
        res['cycles_of_workgroup_200'] = {f'wave{w_}': {names[k]: int(buf[w_ * 8 + k]) for k in range(7)} for w_ in range(4)}
    print(json.dumps(res))


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build()
    elif sys.argv[1] == '--child':
        child(sys.argv[2])
    else:
        for path in sys.argv[1:]:
            r = subprocess.run([sys.executable, __file__, '--child', path], capture_output=True, text=True)
            print(os.path.basename(path), r.stdout.strip() or r.stderr[-400:], flush=True)

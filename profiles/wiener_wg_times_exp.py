"""Experiment: start / end (100 MHz wall clock) of every workgroup of the y-streaming Wiener tile kernel at 12 MP, C = 1.
    python profiles/build_variant.py wiener variants/ys_timing.so -DTDK_EXPERIMENTS -DTDK_YS_TIMING=1
    python profiles/wiener_wg_times_exp.py variants/ys_timing.so"""
import ctypes as C
import json
import sys


def main(path):
    import torch
    lib = C.CDLL(path)
    lib.tdk_wiener_workspace_bytes.restype = C.c_size_t
    dev = torch.device('cuda', 0)
    w, h = 4096, 3072
    g = torch.Generator(device=dev).manual_seed(1)
    sig = torch.tensor([0.075], device=dev)
    plane = torch.rand(h, w, generator=g, device=dev) * 2 - 3
    out = torch.empty_like(plane)
    ws = torch.empty(lib.tdk_wiener_workspace_bytes(w, h, 1, 32, 4), dtype=torch.uint8, device=dev)
    V = C.c_void_p
    run = lambda: lib.tdk_wiener(V(plane.data_ptr()), V(out.data_ptr()), V(ws.data_ptr()), w, h, 1, 32, 4, V(sig.data_ptr()), 0, None)
    for _ in range(3):
        assert run() == 0
    torch.cuda.synchronize()
    tb = (C.c_ulonglong * 4096)()
    lib.tdk_debug_ys_wg_times(tb)
    t = [(tb[2 * b], tb[2 * b + 1]) for b in range(2048) if tb[2 * b + 1] > tb[2 * b] > 0]
    # only the workgroups of the last launch: those whose start lies within 1 ms of the latest start
    latest = max(a for a, _ in t)
    t = [(a, e) for a, e in t if latest - a < 100000]
    t0 = min(a for a, _ in t)
    us = lambda x: round(x / 100.0, 1)
    life = sorted(us(e - a) for a, e in t)
    ends = sorted(us(e - t0) for _, e in t)
    n = len(t)
    print(json.dumps({'workgroups': n, 'launch_span_us': ends[-1], 'last_start_us': us(latest - t0),
                      'lifetime_us': {'min': life[0], 'p25': life[n // 4], 'median': life[n // 2], 'p75': life[3 * n // 4], 'max': life[-1]},
                      'end_us_percentiles': {q: ends[int(q * (n - 1))] for q in (0.1, 0.25, 0.5, 0.75, 0.9, 0.99, 1.0)}}))


if __name__ == '__main__':
    main(sys.argv[1])

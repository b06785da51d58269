"""Experiment: where a step of the register-blocked RCD strips (rq::rcd_quad) spends its time.
    python profiles/build_variant.py rcd variants/rcd_timing.so -DTDK_EXPERIMENTS -DTDK_RCD_TIMING=1
    python profiles/rcd_quad_phase_exp.py variants/rcd_timing.so
Clock deltas of wave 0 of workgroup 300 (mid-frame: INNER blocks), per phase, summed over its 22 steps: entry k = phase k up to
the barrier that closes it INCLUDING the wait, entry 8 + k = up to the arrival at that barrier (see RQ_MARK in tdk_rcd_quad.h)."""
import ctypes as C
import json
import sys


def main(path):
    import torch
    lib = C.CDLL(path)
    lib.tdk_rcd_ex.restype = C.c_int
    lib.tdk_rcd_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_uint, C.c_void_p]
    dev = torch.device('cuda', 0)
    w, h = 4096, 3072
    g = torch.Generator(device=dev).manual_seed(1)
    x32 = torch.rand(h, w, generator=g, device=dev) * 0.9 + 0.05
    for name, dt, tag, flags in (('f16 approximate quad', torch.float16, 1, 2), ('f16 exact quad', torch.float16, 1, 2 | 4), ('f32 quad', torch.float32, 0, 2)):
        x = x32.to(dt)
        y = torch.empty(h, w, 3, dtype=dt, device=dev)
        run = lambda: lib.tdk_rcd_ex(x.data_ptr(), y.data_ptr(), None, w, h, 0x94949494, tag, flags, None)
        for _ in range(3):
            assert run() == 0
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 16)()
        lib.tdk_debug_rcd_phase_cycles(buf, 1)
        run()
        torch.cuda.synchronize()
        lib.tdk_debug_rcd_phase_cycles(buf, 1)
        per_step = [round(buf[k] / 22) for k in range(16)]
        print(json.dumps({'case': name, 'cycles_per_step_incl_wait': per_step[:6], 'sum': sum(per_step[:6]),
                          'cycles_to_barrier_arrival': per_step[9:13]}))
        if hasattr(lib, 'tdk_debug_rcd_wg_times'):
            # start / end of every workgroup (100 MHz wall clock): launch span, workgroup lifetimes by strip / segment class
            tb = (C.c_ulonglong * 2048)()
            lib.tdk_debug_rcd_wg_times(tb)
            nstrips, nwg = 38, 760
            t = [(tb[2 * b], tb[2 * b + 1]) for b in range(nwg)]
            t0 = min(a for a, _ in t)
            us = lambda ticks: round(ticks / 100.0, 1)
            life = [us(e - a) for a, e in t]
            cls = {'interior': [], 'border strip (0, 37)': [], 'first / last segment': []}
            nsegs = nwg // nstrips

            def strip_seg(b):  # rs::strip_map (tdk_rcd_stream.h): border strips first, then the first / last segments, then the rest
                ni, e1, e2 = nstrips - 2, 2 * nsegs, 2 * (nstrips - 2)
                if b < e1:
                    return (nstrips - 1 if b & 1 else 0), b >> 1
                if b < e1 + e2:
                    i = b - e1
                    return 1 + (i >> 1), (nsegs - 1 if i & 1 else 0)
                i = b - e1 - e2
                return 1 + i % ni, 1 + i // ni

            for b in range(nwg):
                strip, seg = strip_seg(b)
                key = 'border strip (0, 37)' if strip in (0, nstrips - 1) else ('first / last segment' if seg in (0, nwg // nstrips - 1) else 'interior')
                cls[key].append(life[b])
            print(json.dumps({'case': name, 'launch_span_us': us(max(e for _, e in t) - t0), 'last_start_us': us(max(a for a, _ in t) - t0),
                              'lifetime_us': {k: {'n': len(v), 'min': min(v), 'median': sorted(v)[len(v) // 2], 'max': max(v)} for k, v in cls.items()},
                              'median_lifetime_by_launch_order_256s': [sorted(life[k:k + 256])[len(life[k:k + 256]) // 2] for k in range(0, nwg, 256)],
                              'end_us_percentiles': [us(sorted(e for _, e in t)[int(q * (nwg - 1))] - t0) for q in (0.1, 0.5, 0.9, 0.99, 1.0)]}))


if __name__ == '__main__':
    main(sys.argv[1])

"""RCD at 12 MP: device time of the strip kernels by arithmetic flavour (float16 results: approximate by default, exact with
TDK_RCD_EXACT) and by variant (rs::rcd_stream / rq::rcd_quad = TDK_RCD_CONCURRENT), plus how far the approximate float16 result
is from the exact one on the whole frame.
    python profiles/rcd_flavours.py [--iters 20] > profiles/r05/experiments/rcd_flavours.txt"""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'torch-darktable_amd'))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--save', default='', help='write the float16 approximate and the float32 results to this file')
    ap.add_argument('--compare', default='', help='compare with the results another library (TDK_LIB_PATH) saved there')
    a = ap.parse_args()
    import torch_darktable as td
    from torch_darktable import _native
    from torch_darktable import torch_darktable_extension as ext
    from torch_darktable.synthetic import synthetic_bayer

    dev = torch.device('cuda', 0)
    w, h = 4096, 3072
    b32 = synthetic_bayer(h, w, 1234, dev)
    b16 = b32.half()
    ws = td.RCD(dev, (w, h), td.BayerPattern.RGGB)

    def timed(x, exact, concurrent):
        def call():
            with ext.verification_paths(rcd_exact=exact), ext.concurrent_frames(concurrent):
                return ws.process(x)
        out = call()
        torch.cuda.synchronize()
        _native.profile_enable(True)
        for _ in range(a.iters):
            call()
        torch.cuda.synchronize()
        rep = _native.profile_report()
        _native.profile_enable(False)
        return out, {k: round(v[1] / v[0] * 1e3, 1) for k, v in rep.items()}

    res = {}
    for name, x, exact, conc in (('f16 approximate, rcd_stream', b16, False, False), ('f16 approximate, rcd_quad', b16, False, True),
                                 ('f16 exact, rcd_stream', b16, True, False), ('f16 exact, rcd_quad', b16, True, True),
                                 ('f32, rcd_stream', b32, False, False), ('f32, rcd_quad', b32, False, True)):
        out, us = timed(x, exact, conc)
        res[name] = out
        print(json.dumps({'case': name, 'us': us}), flush=True)
    fast, exact = res['f16 approximate, rcd_stream'].float(), res['f16 exact, rcd_stream'].float()
    d = (fast - exact).abs()
    ulp = torch.exp2(torch.floor(torch.log2(torch.maximum(fast.abs(), exact.abs()).clamp_min(2.0 ** -14))) - 10)
    print(json.dumps({'whole frame, approximate vs exact float16': {
        'values_differing': float((d > 0).float().mean()), 'pixels_beyond_1_ulp16': int((d > ulp).any(-1).sum()), 'max_abs': float(d.max()),
        'quad_equals_stream': bool(torch.equal(res['f16 approximate, rcd_stream'], res['f16 approximate, rcd_quad']))}}), flush=True)


    if a.save:
        torch.save({'f16': res['f16 approximate, rcd_quad'].cpu(), 'f32': res['f32, rcd_quad'].cpu()}, a.save)
    if a.compare:
        ref = torch.load(a.compare)
        print(json.dumps({'equal to the saved results': {'f16 approximate quad': bool(torch.equal(ref['f16'], res['f16 approximate, rcd_quad'].cpu())),
                                                         'f32 quad': bool(torch.equal(ref['f32'], res['f32, rcd_quad'].cpu()))}}), flush=True)


if __name__ == '__main__':
    main()

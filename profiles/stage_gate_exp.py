"""Experiment: does a fixed software pipeline beat free-running streams?  The 12 MP chain on N streams as bench.py runs it, with stage k of
frame n + 1 made to wait (event) for stage k of frame n -- for every stage, for the RCD only, or not at all.
    python profiles/stage_gate_exp.py [--streams 3] [--steps 100]"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'torch-darktable_amd'))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--streams', type=int, default=3)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--modes', default='free,rcd,all,heavy,free')
    ap.add_argument('--sleep', type=float, default=0.0, help='idle seconds before each timed region (after its warm-up steps)')
    ap.add_argument('--warm', type=int, default=5)
    ap.add_argument('--prio', default='', help='comma list of stream priorities (-1 = high, 0 = normal), e.g. -1,0,0')
    a = ap.parse_args()
    import torch_darktable as td
    from torch_darktable.synthetic import synthetic_bayer
    from torch_darktable.torch_darktable_extension import concurrent_frames

    dev = torch.device('cuda', 0)
    w, h, frames = 4096, 3072, 8
    inputs = [synthetic_bayer(h, w, seed=1234 + i, device=dev).half() for i in range(frames)]
    params = td.TonemapParameters(gamma=0.75, intensity=2.0, light_adapt=1.0, vibrance=0.0)

    def make():
        return dict(rcd=td.RCD(dev, (w, h), td.BayerPattern.RGGB), wiener=td.Wiener(dev, (w, h), overlap_factor=4, tile_size=32),
                    bil=td.Bilateral(dev, (w, h), sigma_s=2.0, sigma_r=0.2), lum=torch.empty((h, w), dtype=torch.float32, device=dev),
                    acc=td.tonemap.MetricsAccumulator(dev, stride=8))

    chains = [make() for _ in range(a.streams)]
    prio = [int(x) for x in a.prio.split(',')] if a.prio else [0] * a.streams
    streams = [torch.cuda.Stream(dev, priority=prio[k % len(prio)]) for k in range(a.streams)]
    print('stream priorities', prio, torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else '')
    ap_modes = a.modes.split(',')
    for mode in ap_modes:
        nxt = [0]
        prev = [None] * 4

        def step():
            with concurrent_frames():
                for b in inputs:
                    k = nxt[0]
                    nxt[0] = (k + 1) % a.streams
                    c, s = chains[k], streams[k]
                    gate = {'free': (), 'rcd': (0,), 'all': (0, 1, 2, 3), 'heavy': (0, 1, 2)}[mode]
                    with torch.cuda.stream(s):
                        def stage(i, fn):
                            if i in gate and prev[i] is not None:
                                s.wait_event(prev[i])
                            r = fn()
                            if i in gate:
                                e = torch.cuda.Event()
                                e.record(s)
                                prev[i] = e
                            return r
                        x = stage(0, lambda: c['rcd'].process(b))
                        x = stage(1, lambda: c['wiener'].process_log_luminance(x, 0.075, luminance_out=c['lum']))
                        x = stage(2, lambda: c['bil'].process_rgb(x, 0.4, luminance=c['lum'], metrics=c['acc']))
                        stage(3, lambda: td.reinhard_tonemap(x, c['acc'].finish(), params))

        for _ in range(a.warm):
            step()
        torch.cuda.synchronize()
        if a.sleep:
            time.sleep(a.sleep)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'{mode:6s} streams {a.streams}: {frames * a.steps * w * h / 1e6 / dt:9.1f} MP/s  {dt / a.steps * 1e3:.4f} ms/step', flush=True)


if __name__ == '__main__':
    main()

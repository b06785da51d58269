"""RAW mosaic -> JPEG bytes on the host, per GPU: the chain bench.py times (RCD -> Wiener log-L -> bilateral -> Reinhard, Lab hand-over,
12 MP, float16 storage) followed by Jpeg.encode (4:2:2, quality 94, baseline) of the tone-mapped frame.  One host thread per HIP stream,
each with its own chain, coder and workspace (Jpeg.encode synchronises its stream for the Huffman statistics and the length; the
library call releases the GIL, so the other threads keep issuing).

  python3 profiles/raw_to_jpeg.py [--threads 3] [--frames 24] [--repeats 5]
"""
import argparse
import json
import sys
import threading
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'torch-darktable_amd'))
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import torch_darktable as td  # noqa: E402
from torch_darktable.synthetic import synthetic_bayer  # noqa: E402
from torch_darktable.torch_darktable_extension import concurrent_frames  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--threads', type=int, default=3)
    ap.add_argument('--frames', type=int, default=24, help='frames per repeat, split over the threads')
    ap.add_argument('--repeats', type=int, default=5)
    ap.add_argument('--overlap', action='store_true', help='two streams per thread: the chain of the next frame is queued before the current frame is encoded')
    ap.add_argument('--no-jpeg', action='store_true', help='the chain alone through the same threads (what the encoder adds = the difference)')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    w, h = bench.W12, bench.H12
    frames = [synthetic_bayer(h, w, 100 + i, dev).half() for i in range(4)]
    torch.cuda.synchronize()
    sizes = []

    def worker(k, n, barrier, out):
        stream = torch.cuda.Stream(dev)
        jstream = torch.cuda.Stream(dev)   # --overlap: the coder's own stream; frame i + 1's chain is queued before frame i is encoded
        _, chain = bench.build_pipeline(td, dev, w, h, 'f16', 'isp', 'lab')
        coder = td.Jpeg()

        def encode(u8, first):
            data = coder.encode(u8, 94, td.InputFormat.RGBI, td.Subsampling.CSS_422, False)
            if first:
                out.append(int(data.numel()))

        with concurrent_frames(a.threads > 1 or a.overlap):
            for rep in range(a.repeats + 1):   # repeat 0 = warm-up (workspaces, LDS limits)
                barrier.wait()
                pending = None
                for i in range(n):
                    with torch.cuda.stream(stream):
                        u8 = chain(frames[(k + i) % len(frames)])
                        if not a.no_jpeg and not a.overlap:
                            encode(u8, rep == 0 and i == 0)
                    if a.no_jpeg or not a.overlap:
                        continue
                    done = torch.cuda.Event()
                    done.record(stream)
                    if pending is not None:   # the host blocks in the encoder of frame i - 1 while the GPU already holds frame i's chain
                        with torch.cuda.stream(jstream):
                            jstream.wait_event(pending[1])
                            encode(pending[0], False)
                    pending = (u8, done)
                if pending is not None:
                    with torch.cuda.stream(jstream):
                        jstream.wait_event(pending[1])
                        encode(pending[0], rep == 0)
                stream.synchronize()
                jstream.synchronize()
                barrier.wait()

    per = [a.frames // a.threads + (1 if k < a.frames % a.threads else 0) for k in range(a.threads)]
    barrier = threading.Barrier(a.threads + 1)
    ts = [threading.Thread(target=worker, args=(k, per[k], barrier, sizes)) for k in range(a.threads)]
    for t in ts:
        t.start()
    times = []
    for rep in range(a.repeats + 1):
        barrier.wait()
        t0 = time.perf_counter()
        barrier.wait()
        if rep:
            times.append(time.perf_counter() - t0)
    for t in ts:
        t.join()
    best, mean = min(times), sum(times) / len(times)
    mp = w * h * a.frames / 1e6
    print(json.dumps({'what': 'chain only' if a.no_jpeg else 'RAW -> JPEG bytes on the host', 'threads': a.threads, 'overlap': bool(a.overlap), 'frames_per_repeat': a.frames,
                      'ms_per_frame_mean': round(mean / a.frames * 1e3, 4), 'ms_per_frame_best': round(best / a.frames * 1e3, 4),
                      'MP_per_s_mean': round(mp / mean, 1), 'MP_per_s_best': round(mp / best, 1), 'frames_per_s_mean': round(a.frames / mean, 1),
                      'jpeg_bytes': sizes[:1]}))


if __name__ == '__main__':
    main()

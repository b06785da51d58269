#!/usr/bin/env python3
"""Headline benchmark: megapixels/s of the full RAW ISP on synthetic 12 MP RGGB frames.

  python bench.py --gpus N --steps K --warmup W [--workload isp|rcd] [--storage f16|f32]

One step = one pass of the hot path over one batch of device-resident synthetic frames per GPU:
  isp (BASELINE.json configs[2], the configuration the metric is quoted on):
      8 x 4096x3072 RGGB, fp16 storage / fp32 arithmetic,
      RCD -> Wiener.process_log_luminance(0.075; K=32, ov=4) -> Bilateral.process_rgb(sigma_s=2,
      sigma_r=0.2, detail=0.4) -> compute_image_metrics(stride 8) -> reinhard_tonemap(gamma .75,
      intensity 2.0, light_adapt 1.0) -> uint8    (ImageProcessingSettings defaults,
      reference torch_darktable/pipeline/config.py:114-146; "nlmeans" in BASELINE.json has no
      counterpart in the reference, its denoiser is the tiled-FFT Wiener filter)
  rcd (configs[1]): one 4096x3072 fp32 frame through RCD.process.

Multi-GPU: one process per GPU (torch.distributed.run), frames are independent, so every rank
processes its own batch -- weak scaling, no data-path collective; the only communication is the
barrier and the max-over-ranks of the timing.

Prints ONE JSON line on rank 0 (contract in the task description): metric/value/unit...,
"roofline" for the dominant kernel (per-kernel device time measured live with HIP events on the
launch stream by the library's tdk_profile_* timer), and "cpu_baseline" (the strict-fp32 CPU
oracle, i.e. a port -- the reference has no CPU path -- timed on a bounded sample).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / 'torch-darktable_amd'))
sys.path.insert(0, str(ROOT))

VALU_CUS, VALU_CLOCK_GHZ = 256, 2.4  # MI355X: 256 CUs x 4 SIMDs, 2.4 GHz peak engine clock (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured float4 copy
FP32_VECTOR_PEAK_TFLOPS = 157.3

W12, H12 = 4096, 3072


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', choices=['isp', 'rcd'], default='isp')
    ap.add_argument('--storage', choices=['f16', 'f32'], default=None, help='image storage type (default: f16 for isp, f32 for rcd)')
    ap.add_argument('--frames', type=int, default=None, help='frames per GPU per step (default 8 for isp, 1 for rcd)')
    ap.add_argument('--width', type=int, default=W12)
    ap.add_argument('--height', type=int, default=H12)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-dist', action='store_true', help='rehearsal: create the RCCL process group and run the barriers / max-reduce even with one rank')
    ap.add_argument('--no-kernel-timer', action='store_true', help='leave the per-kernel event timer off in the timed region')
    return ap.parse_args()


# Algorithmic (compulsory) bytes per pixel of each kernel family at its op boundary:
# (bytes in + bytes out) with s = bytes per stored sample (2 for f16, 4 for f32).  SURVEY.md 8(d).
def algorithmic_bytes_per_px(kernel: str, s: int) -> float | None:
    table = {
        'tdk_rcd': 1 * s + 3 * s,                    # bayer in, rgb out
        # on the fused Wiener.process_log_luminance path the op boundary is RGB in, RGB out (the
        # log-luminance planes are internal): SURVEY.md 8(d) "denoise 6 + 6 = 12 B/px" at fp16
        'tdk_wiener(tiles)': 3 * s + 3 * s,
        'tdk_wiener(finish)': 1 * s + 1 * s,
        'tdk_wiener(finish+modify)': 3 * s + 3 * s,
        'tdk_bilateral(slice+modify)': 3 * s + 3 * s,
        'tdk_bilateral(tiles)': 3 * s + 3 * s,       # fused op on the process_rgb path: RGB in, RGB out
        'tdk_bilateral(splat)': 1 * s + 1 * s,       # the bilateral op on one plane
        'tdk_bilateral(blur_xy)': 1 * s + 1 * s,
        'tdk_bilateral(blur_z)': 1 * s + 1 * s,
        'tdk_bilateral(slice)': 1 * s + 1 * s,
        'tdk_compute_luminance': 3 * s + 1 * s,
        'tdk_modify_luminance': 3 * s + 1 * s + 3 * s,
        'tdk_tonemap': 3 * s + 3,
        'tdk_image_metrics_accumulate': 0.0,
    }
    return table.get(kernel)


def build_pipeline(td, dev, w, h, storage, workload):
    import torch

    dtype = torch.float16 if storage == 'f16' else torch.float32
    rcd = td.RCD(dev, (w, h), td.BayerPattern.RGGB)
    if workload == 'rcd':
        return dtype, lambda bayer: rcd.process(bayer)
    wiener = td.Wiener(dev, (w, h), overlap_factor=4, tile_size=32)
    bilateral = td.Bilateral(dev, (w, h), sigma_s=2.0, sigma_r=0.2)
    params = td.TonemapParameters(gamma=0.75, intensity=2.0, light_adapt=1.0, vibrance=0.0)

    def frame(bayer):
        rgb = rcd.process(bayer)
        rgb = wiener.process_log_luminance(rgb, 0.075)
        rgb = bilateral.process_rgb(rgb, 0.4)
        metrics = td.compute_image_metrics([rgb], stride=8)  # per-frame statistics (moving_average = 1)
        return td.reinhard_tonemap(rgb, metrics, params)

    return dtype, frame


def cpu_baseline(workload, threads):
    """The oracle chain (a CPU port: the reference has no CPU path) on a bounded sample."""
    import numpy as np

    sys.path.insert(0, str(ROOT / 'oracle'))
    os.environ['OMP_NUM_THREADS'] = str(threads)
    import tdk_oracle as O
    from torch_darktable.synthetic import synthetic_bayer

    O.build()
    sw, sh = (2048, 1536) if workload == 'isp' else (4096, 3072)
    bayer = synthetic_bayer(sh, sw, seed=1234, device='cpu').numpy()

    def run():
        rgb = O.rcd(bayer, O.RGGB)
        if workload == 'rcd':
            return rgb
        ll = O.compute_luminance(rgb, log=True, eps=1e-4)
        rgb = O.modify_luminance(rgb, O.wiener(ll[:, :, None], 0.075, 32, 4)[:, :, 0], log=True)
        rgb = O.modify_luminance(rgb, O.bilateral(O.compute_luminance(rgb), 2.0, 0.2, 0.4))
        m = O.image_metrics([rgb], 8)
        return O.tonemap('reinhard', rgb, m, 0.75, 2.0, 1.0, 0.0)

    run()  # warm-up (page-in, OpenMP pool)
    reps, t0 = 0, time.perf_counter()
    while reps < 3 or (time.perf_counter() - t0 < 8.0 and reps < 20):
        run()
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return {
        'value': round(sw * sh / 1e6 / dt, 3), 'unit': 'MP/s', 'cores': threads, 'kind': 'port',
        'sample': f'{reps} x one {sw}x{sh} RGGB frame through the strict-fp32 C oracle of the same chain (OpenMP, {threads} threads); '
                  'fp32 storage (the oracle has no fp16 mode)',
    }


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus or world == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)  # before the process group: RCCL binds the communicator to the current device
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # RCCL printf()s its version banner to STDOUT when the communicator is created (NCCL_DEBUG=VERSION
        # is exported on these boxes); stdout must carry exactly one JSON line, so fd 1 points at stderr
        # until the communicator exists.
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend='nccl', rank=rank, world_size=world, device_id=dev)
            dist.barrier(device_ids=[local_rank])
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    import __graft_entry__

    __graft_entry__.ensure_built()
    import torch_darktable as td
    from torch_darktable import _native
    from torch_darktable.synthetic import synthetic_bayer

    storage = args.storage or ('f16' if args.workload == 'isp' else 'f32')
    frames = args.frames or (8 if args.workload == 'isp' else 1)
    w, h = args.width, args.height
    dtype, process = build_pipeline(td, dev, w, h, storage, args.workload)

    # device-resident synthetic inputs, per-frame seeds 1234 + i (distinct per rank)
    inputs = [synthetic_bayer(h, w, seed=1234 + rank * frames + i, device=dev).to(dtype) for i in range(frames)]
    torch.cuda.synchronize()

    def step():
        out = None
        for b in inputs:
            out = process(b)
        return out

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    use_timer = not args.no_kernel_timer
    # Untimed profiling pass: every launch of one step bracketed by events -> the per-kernel table
    # and the dominant kernel.  In the timed region only THAT kernel keeps its events (measured live,
    # on its launch stream), so the throughput number is not taxed by ~100 event pairs per step.
    table, dom = {}, None
    if use_timer:
        _native.profile_enable(True)
        step()
        torch.cuda.synchronize()
        table = _native.profile_report()
        _native.profile_enable(False)
        dom = max(table.items(), key=lambda kv: kv[1][1])[0]
    if use_dist:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()
    if use_timer:
        _native.profile_enable(True, only=dom)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    report = _native.profile_report() if use_timer else {}
    if use_timer:
        _native.profile_enable(False)
    if use_dist:
        dist.barrier(device_ids=[local_rank])
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    mp_per_frame = w * h / 1e6
    total_frames = frames * args.steps * world
    value = total_frames * mp_per_frame / elapsed
    sbytes = 2 if storage == 'f16' else 4

    roofline = None
    stage_ms = {}
    if report:
        launches_total = sum(c for c, _ in table.values()) * args.steps
        stage_ms = {k: round(ms / frames, 4) for k, (c, ms) in sorted(table.items(), key=lambda kv: -kv[1][1])}  # from the untimed pass
        cnt, ms = report[dom]  # the dominant kernel, timed live in the timed region
        avg_s = ms / cnt / 1e3
        bpp = algorithmic_bytes_per_px(dom, sbytes)
        achieved = (bpp * w * h / avg_s / 1e9) if bpp else None
        traffic, valu = None, None
        tfile = ROOT / 'profiles' / 'traffic.json'  # written by profiles/collect_traffic.py from rocprofv3 --pmc passes
        if tfile.exists():
            tj = json.loads(tfile.read_text())
            traffic = tj.get(dom)
            insts = tj.get('_valu', {}).get(dom)  # SQ_INSTS_VALU: wave-instructions per launch (PMC pass of the same build)
            if insts:
                # a SIMD issues one wave64 VALU instruction per 4 cycles: peak = CUs * 4 SIMDs * clock / 4
                peak_ginst = VALU_CUS * 4 * VALU_CLOCK_GHZ / 4
                valu = {'wave_insts_per_launch': insts, 'achieved_Ginst_per_s': round(insts / avg_s / 1e9, 1), 'peak_Ginst_per_s': round(peak_ginst, 1),
                        'issue_frac': round(insts / avg_s / 1e9 / peak_ginst, 4),
                        'note': 'the kernel is FP32-vector bound; bytes/s against HBM is reported above because the contract asks for it'}
        roofline = {
            'kernel': dom, 'bound': 'hbm', 'achieved': round(achieved, 2) if achieved else None, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(achieved / HBM_PEAK_GBS, 5) if achieved else None, 'traffic': traffic,
            'avg_launch_us': round(avg_s * 1e6, 2), 'launches': cnt, 'algorithmic_bytes_per_launch': int(bpp * w * h) if bpp else None,
            'kernel_launches_in_timed_region': launches_total, 'valu': valu,
        }
    # whole-pipeline roofline at the Python-wrapper stage boundaries (41 B/px f16, 79 B/px f32; RCD only: 4*s B/px)
    pipe_bpp = (41 if storage == 'f16' else 79) if args.workload == 'isp' else 4 * sbytes
    pipe_gbs = pipe_bpp * w * h * total_frames / world / elapsed / 1e9

    out = {
        'metric': 'megapixels/sec full ISP (debayer->denoise->tonemap) 12MP RGGB' if args.workload == 'isp' else 'megapixels/sec RCD demosaic 12MP RGGB',
        'value': round(value, 2), 'unit': 'MP/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {
            'workload': ('12 MP full pipeline (RCD -> Wiener log-L sigma=0.075 K=32 ov=4 -> bilateral sigma_s=2 sigma_r=0.2 detail=0.4 -> '
                         'metrics -> Reinhard gamma=0.75 intensity=2 light_adapt=1 -> u8), batch 8 per GPU') if args.workload == 'isp'
                        else '12 MP RCD demosaic, single frame',
            'width': w, 'height': h, 'frames_per_gpu_per_step': frames, 'storage': storage, 'arithmetic': 'f32',
            'denoiser': 'Wiener (the reference has no nlmeans)', 'sharding': 'independent frames per GPU, no collective',
        },
        'pipeline_roofline': {'algorithmic_bytes_per_px': pipe_bpp, 'achieved_GBps_per_gpu': round(pipe_gbs, 2), 'frac_of_8TBps': round(pipe_gbs / HBM_PEAK_GBS, 5)},
        'roofline': roofline,
        'kernel_ms_per_frame': stage_ms,
        'kernel_ms_per_frame_source': 'one untimed step with every launch bracketed by events; the roofline kernel is timed live in the timed region',
    }
    if world == 1 and not args.no_cpu_baseline:
        try:
            out['cpu_baseline'] = cpu_baseline(args.workload, os.cpu_count() or 1)
        except Exception as e:  # noqa: BLE001
            out['cpu_baseline'] = {'value': None, 'unit': 'MP/s', 'cores': 0, 'kind': 'port', 'sample': f'failed: {e}'}
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Headline benchmark: megapixels/s of the full RAW ISP on synthetic 12 MP RGGB frames.

  python bench.py --gpus N --steps K --warmup W [--workload isp|rcd|ppg_wiener50] [--storage f16|f32]

One step = one pass of the hot path over one batch of device-resident synthetic frames per GPU:
  isp (BASELINE.json configs[2], the configuration the metric is quoted on):
      8 x 4096x3072 RGGB, fp16 storage / fp32 arithmetic,
      RCD -> Wiener.process_log_luminance(0.075; K=32, ov=4) -> Bilateral.process_rgb(sigma_s=2,
      sigma_r=0.2, detail=0.4) -> compute_image_metrics(stride 8) -> reinhard_tonemap(gamma .75,
      intensity 2.0, light_adapt 1.0) -> uint8    (ImageProcessingSettings defaults,
      reference torch_darktable/pipeline/config.py:114-146; "nlmeans" in BASELINE.json has no
      counterpart in the reference, its denoiser is the tiled-FFT Wiener filter)
  rcd (configs[1]): one 4096x3072 fp32 frame through RCD.process.
  ppg_wiener50 (configs[4]): 4 x 8192x6144 RGGB per GPU, fp16 storage, PPG.process -> Wiener.process C=3 (K=32, ov=4,
      sigma=0.05 as reference scripts/run_benchmark.py:96; the reference has no wavelet denoiser).
The frames of a batch are independent; they are issued round-robin on --streams HIP streams (default 3), each with
its own op workspaces, so one frame's kernel tails overlap the next frame's kernels.

Multi-GPU (BASELINE.json configs[3]): one process per GPU.  Frames are independent, so every rank
processes its own batch on its own stream -- weak scaling, NO collective on the data path and no
RCCL communicator at all: the barrier and the max-over-ranks of the timing go through a `gloo`
group on the host.  `python bench.py --gpus N` with N > 1 and no launcher starts the N ranks
itself (the parent never touches the GPU, it only relays rank 0's line); under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the launcher's ranks are
used as they are.  The number of ranks that ran must equal --gpus or the run fails: a line with
`n_gpus` different from `--gpus` cannot be printed.

Prints ONE JSON line on rank 0 (contract in the task description): metric/value/unit...,
"roofline" for the dominant kernel = its HBM fraction: the algorithmic bytes of the stage it implements
divided by its average launch time (measured live with HIP events on the launch stream by the library's tdk_profile_*
timer) against 8 TB/s; the vector-ALU issue floors (which are what actually bound these kernels) are reported beside it
under "valu" / "composite", never as `frac`; "cpu_baseline" (the strict-fp32 CPU oracle, i.e. a port -- the reference has
no CPU path -- timed on a bounded sample) and "parity_check": the uint8 output of GPU frame 0 in the last timed step
against the oracle's output for the same input.  Exit code 3 if that check fails.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / 'torch-darktable_amd'))
sys.path.insert(0, str(ROOT))

# MI355X constants (MI355X_MICROARCH.md): 256 CUs x 4 SIMD-32, 2.4 GHz peak engine clock.  A wave64
# VALU instruction occupies its SIMD for 2 cycles (4 is what ONE wave alone sustains), a transcendental
# (v_exp/v_log/v_rcp/v_rsq/v_sqrt) for 8 -- so the chip issues at most 256*4*2.4/2 = 1228.8 G plain
# wave-instructions/s (= the 157.3 TFLOP/s fp32 vector peak with FMAs).  Measured on the box
# (tests/hip_unit/valu_issue_bench.hip, profiles/r02/microbench.txt): 2.4 cycles plain, 8.5 transcendental,
# and twice the plain cost for any VALU instruction with an SGPR / VCC source operand -- the floor below
# uses the nominal 2 / 8 and is therefore optimistic.
VALU_CUS, VALU_SIMDS, VALU_CLOCK_GHZ = 256, 4, 2.4
VALU_CYCLES_PLAIN, VALU_CYCLES_TRANS, VALU_CYCLES_PACKED = 2, 8, 4
# what the same instruction classes cost on the box (tests/hip_unit/valu_issue_bench.hip, pk_issue_bench.hip: DESIGN.md 3.0)
VALU_MEASURED_PLAIN, VALU_MEASURED_TRANS, VALU_MEASURED_PACKED = 2.4, 8.5, 4.3
VALU_PEAK_GINST = VALU_CUS * VALU_SIMDS * VALU_CLOCK_GHZ / VALU_CYCLES_PLAIN
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured float4 copy

W12, H12 = 4096, 3072
W50, H50 = 8192, 6144


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', choices=['isp', 'rcd', 'ppg_wiener50'], default='isp')
    ap.add_argument('--storage', choices=['f16', 'f32'], default=None, help='image storage type (default: f32 for rcd, f16 otherwise)')
    ap.add_argument('--frames', type=int, default=None, help='frames per GPU per step (default 8 for isp, 1 for rcd, 4 for ppg_wiener50)')
    ap.add_argument('--streams', type=int, default=3, help='HIP streams the frames of a batch are spread over (each with its own workspaces)')
    ap.add_argument('--chain', choices=['lab', 'rgb'], default='lab',
                    help='isp: how the pixel travels from the denoiser to the local-contrast stage: lab = lightness + chroma planes (one colour round trip), rgb = the intermediate RGB image')
    ap.add_argument('--graph', action='store_true', help='replay each step as one captured HIP graph (measurement option; no live per-kernel timing)')
    ap.add_argument('--width', type=int, default=None)
    ap.add_argument('--height', type=int, default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip the CPU oracle run (and with it the parity check)')
    ap.add_argument('--pin-cpus', action='store_true', help='self-launched ranks: give each rank its own slice of the host cores (os.sched_setaffinity)')
    ap.add_argument('--no-kernel-timer', action='store_true', help='leave the per-kernel event timer off in the timed region')
    ap.add_argument('--stub-cpu', action='store_true',
                    help='REHEARSAL ONLY (tests/test_bench_launch.py): run the rank/launch/timing protocol on the CPU with a stand-in '
                         'stage instead of the HIP pipeline; the line is labelled "stub" and is not a measurement')
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------
# rank launch: `bench.py --gpus N` without a launcher starts its own N ranks
# ----------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def cpu_slices(n: int) -> list[list[int]]:
    """The host cores this process may run on, cut into n contiguous slices (rank r issues its ~1 000 launches per step from
    its own cores: no rank's launch thread migrates onto another's).  Contiguous core numbers share a socket / NUMA node on
    the usual enumeration; the GPU -> NUMA affinity itself is not visible without root tools, so none is assumed."""
    cores = sorted(os.sched_getaffinity(0))
    per = max(1, len(cores) // n)
    return [cores[(r * per) % len(cores):(r * per) % len(cores) + per] for r in range(n)]


def _parse_cpulist(text: str) -> list[int]:
    out = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_nodes(sysfs: str = '/sys') -> list[int | None]:
    """NUMA node of every GPU, in the order of the KFD topology (= the HIP device order unless *_VISIBLE_DEVICES re-maps it, which
    is applied here when it is a plain index list).  Read from sysfs only -- no HIP call, so a rank can pin itself before it touches
    the GPU: topology node -> PCI address (domain, location_id) -> /sys/bus/pci/devices/<bdf>/numa_node.  None where unknown."""
    base = Path(sysfs) / 'class' / 'kfd' / 'kfd' / 'topology' / 'nodes'
    nodes = []
    try:
        dirs = sorted((d for d in base.iterdir() if d.name.isdigit()), key=lambda d: int(d.name))
    except OSError:
        return []
    for d in dirs:
        try:
            props = dict(ln.split()[:2] for ln in (d / 'properties').read_text().splitlines() if len(ln.split()) >= 2)
        except OSError:
            continue
        if int(props.get('simd_count', '0')) == 0:
            continue  # a CPU node of the topology
        dom, loc = int(props.get('domain', '0')), int(props.get('location_id', '0'))
        bdf = f'{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}'
        try:
            node = int((Path(sysfs) / 'bus' / 'pci' / 'devices' / bdf / 'numa_node').read_text())
        except (OSError, ValueError):
            node = -1
        nodes.append(node if node >= 0 else None)
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        spec = os.environ.get(var, '')
        if spec and all(t.strip().isdigit() for t in spec.split(',')):
            nodes = [nodes[int(t)] if int(t) < len(nodes) else None for t in spec.split(',')]
    return nodes


def numa_cpu_slices(n: int, sysfs: str = '/sys', allowed: list[int] | None = None) -> list[list[int]]:
    """Cores for each of n local ranks (rank r drives GPU r): the cores of ITS GPU's NUMA node, shared evenly among the ranks
    whose GPUs sit on that node; ranks whose node is unknown (no sysfs entry, a container that hides it) get the plain index
    split of cpu_slices().  Launch latency is host work on the path to the GPU's PCIe root: issuing from the remote socket adds
    an inter-socket hop to every doorbell and every event query."""
    cores = sorted(os.sched_getaffinity(0)) if allowed is None else sorted(allowed)
    per = max(1, len(cores) // n)
    fallback = [cores[(r * per) % len(cores):(r * per) % len(cores) + per] for r in range(n)]
    gpu_nodes = gpu_numa_nodes(sysfs)
    out: list[list[int] | None] = [None] * n
    by_node: dict[int, list[int]] = {}
    for r in range(n):
        node = gpu_nodes[r] if r < len(gpu_nodes) else None
        if node is not None:
            by_node.setdefault(node, []).append(r)
    for node, ranks_here in by_node.items():
        try:
            node_cpus = [c for c in _parse_cpulist((Path(sysfs) / 'devices' / 'system' / 'node' / f'node{node}' / 'cpulist').read_text()) if c in set(cores)]
        except OSError:
            node_cpus = []
        share = len(node_cpus) // len(ranks_here)
        if share < 1:
            continue
        for i, r in enumerate(ranks_here):
            out[r] = node_cpus[i * share:(i + 1) * share]
    return [out[r] if out[r] else fallback[r] for r in range(n)]


def pin_self() -> list[int] | None:
    """Pin this rank to its cores before it starts a thread or touches the GPU: TDK_BENCH_CPUS from a self-launching parent
    (--pin-cpus), else -- under an external launcher (torchrun) with more than one local rank -- the NUMA-aware slice of this
    rank computed here from LOCAL_RANK / LOCAL_WORLD_SIZE (TDK_BENCH_NO_PIN=1 switches that off)."""
    spec = os.environ.get('TDK_BENCH_CPUS')
    try:
        if spec:
            cpus = [int(c) for c in spec.split(',')]
        else:
            lws = int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1')))
            if lws <= 1 or os.environ.get('TDK_BENCH_NO_PIN') or os.environ.get('TDK_BENCH_SELF_LAUNCHED'):
                return None
            cpus = numa_cpu_slices(lws, os.environ.get('TDK_BENCH_SYSFS', '/sys'))[int(os.environ.get('LOCAL_RANK', '0')) % lws]
        os.sched_setaffinity(0, cpus)
        return cpus
    except (OSError, ValueError):
        return None


def launch_ranks(args, argv) -> int:
    """Parent of a self-launched multi-GPU run.  Starts args.gpus children of this script (one per
    GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their environment), waits for all of them, relays
    rank 0's single JSON line and returns non-zero if any rank failed.  The parent makes no GPU
    call (so nothing is exec'ed or forked from a process that has initialised HIP)."""
    n = args.gpus
    if not args.stub_cpu:
        import __graft_entry__

        __graft_entry__.ensure_built()  # build once here, not N times under a lock (hipcc needs no GPU)
    env = dict(os.environ)
    env.update({'WORLD_SIZE': str(n), 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(_free_port()), 'TDK_BENCH_SELF_LAUNCHED': '1'})
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    procs = []
    slices = numa_cpu_slices(n, os.environ.get('TDK_BENCH_SYSFS', '/sys')) if args.pin_cpus else [None] * n
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        if slices[r]:
            e['TDK_BENCH_CPUS'] = ','.join(map(str, slices[r]))  # the child pins itself before it starts any thread (pin_self)
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *argv], env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode]
    deadline = time.time() + 120.0
    for p in procs[1:]:
        try:
            codes.append(p.wait(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:
            p.kill()  # the exact child we started, by handle
            codes.append(-9)
    if any(c != 0 for c in codes):
        for p in procs:
            if p.poll() is None:
                p.kill()
        print(f'bench.py: rank exit codes {codes} -- no result line', file=sys.stderr)
        return 1
    line = None
    for ln in (out0 or '').splitlines():
        ln = ln.strip()
        if ln.startswith('{'):
            try:
                if json.loads(ln).get('n_gpus') == n:
                    line = ln
            except ValueError:
                pass
    if line is None:
        print(f'bench.py: rank 0 printed no JSON line with n_gpus == {n}', file=sys.stderr)
        return 1
    print(line)
    return 0


class Ranks:
    """The host-side rendezvous of a run: gloo barrier, max / gather of a float.  World 1 = no group."""

    def __init__(self, args):
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        if self.world != args.gpus:
            raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={self.world} ranks; refusing to report a '
                             'line whose n_gpus differs from --gpus')
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
            dist.init_process_group(backend='gloo', rank=self.rank, world_size=self.world)  # host-side only: no RCCL communicator exists
            self.dist = dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def gather(self, x: float):
        """Every rank's value, on every rank (rank order)."""
        if not self.dist:
            return [x]
        import torch

        t = [torch.zeros(1, dtype=torch.float64) for _ in range(self.world)]
        self.dist.all_gather(t, torch.tensor([x], dtype=torch.float64))
        return [float(v) for v in t]

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


# algorithmic bytes per pixel of each NAMED STAGE at the Python-wrapper boundary (s = bytes per stored sample:
# 2 for f16, 4 for f32) and the kernels that implement the stage.  roofline.achieved prices the dominant kernel on the bytes of
# ITS stage (what the stage has to move, whatever the kernel split inside it).
def stages(workload: str, s: int):
    if workload == 'isp':
        return {
            'debayer': (1 * s + 3 * s, ['tdk_rcd', 'tdk_rcd(concurrent)', 'tdk_rcd(border)']),
            'denoise': (3 * s + 3 * s, ['tdk_compute_luminance', 'tdk_compute_luminance(lab)', 'tdk_wiener(tiles)', 'tdk_wiener(finish+modify)', 'tdk_wiener(finish+lab)']),
            'local_contrast': (3 * s + 3 * s, ['tdk_bilateral(tiles)', 'tdk_bilateral(tables)', 'tdk_bilateral(slice+modify)', 'tdk_bilateral(slice+lab)', 'tdk_bilateral(splat)',
                                                'tdk_bilateral(blur_xy)', 'tdk_bilateral(blur_z)']),
            'tonemap': (3 * s + 3, ['tdk_image_metrics', 'tdk_image_metrics_accumulate', 'tdk_image_metrics_finish', 'tdk_tonemap']),
        }
    if workload == 'rcd':
        return {'debayer': (1 * s + 3 * s, ['tdk_rcd', 'tdk_rcd(border)'])}
    return {'debayer': (1 * s + 3 * s, ['tdk_ppg', 'tdk_ppg(pre_median)']),
            'denoise': (3 * s + 3 * s, ['tdk_wiener(tiles)', 'tdk_wiener(finish)'])}


def stage_of(kernel: str, workload: str, s: int):
    for name, (bpp, kernels) in stages(workload, s).items():
        if kernel in kernels:
            return name, bpp
    return None, None


# Bytes ONE LAUNCH of a kernel has to move per pixel (its own compulsory reads + writes): what the composite floor prices a
# kernel's HBM time on.
def launch_bytes_per_px(kernel: str, s: int, chain: str = 'lab') -> float | None:
    table = {
        'tdk_rcd': 1 * s + 3 * s,
        'tdk_rcd(concurrent)': 1 * s + 3 * s,                  # the register-blocked strips (frames in flight on other streams)
        'tdk_ppg': 1 * s + 3 * s,
        'tdk_compute_luminance': 3 * s + 4,                    # RGB in, fp32 (log-)lightness plane out
        'tdk_wiener(tiles)': 4 + 4,                            # fp32 plane in, the denoised plane's sums out (fp32 slabs)
        'tdk_wiener(finish+modify)': 4 + 3 * s + 3 * s + 4,    # sums in, RGB in, RGB out, fp32 lightness of the result out
        'tdk_wiener(finish)': 4 + 1 * s,
        'tdk_compute_luminance(lab)': 3 * s + 4 + 8,           # Lab hand-over chain: RGB in, fp32 log-lightness and fp32 (a, b) planes out
        'tdk_wiener(finish+lab)': 4 + 8 + 4,                   # sums in, (a, b) in, fp32 lightness of the result out
        'tdk_bilateral(tiles)': (4 + 8 + 3 * s) if chain == 'lab' else (4 + 3 * s + 3 * s),  # fp32 lightness in, (a, b) or RGB in, RGB out
        'tdk_tonemap': 3 * s + 3,
        'tdk_image_metrics': 3 * s / 64.0,                     # stride-8 sample grid (one launch: sums + finish by the last workgroup)
        'tdk_image_metrics_accumulate': 3 * s / 64.0,
        'tdk_image_metrics_finish': 0.0,
        'tdk_rcd(border)': 0.0,                                # the 7-px ring: its bytes are in tdk_rcd's boundary count
        'tdk_bilateral(tables)': 0.0,                          # axis tables, a few hundred KB, once per workspace
    }
    return table.get(kernel)


def csrc_sha() -> str:
    """Content hash of the kernel sources: profiles/collect_traffic.py stores it with the counters it captures, and the counters
    are only used for a floor when they were captured from these very sources."""
    import hashlib

    h = hashlib.sha256()
    for f in sorted((ROOT / 'torch-darktable_amd' / 'csrc').glob('*')):
        if f.suffix in ('.hip', '.h'):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def build_pipeline(td, dev, w, h, storage, workload, chain='lab'):
    import torch

    dtype = torch.float16 if storage == 'f16' else torch.float32
    if workload == 'ppg_wiener50':
        ppg = td.PPG(dev, (w, h), td.BayerPattern.RGGB)
        wiener3 = td.Wiener(dev, (w, h), overlap_factor=4, tile_size=32)
        return dtype, lambda bayer: wiener3.process(ppg.process(bayer), 0.05)
    rcd = td.RCD(dev, (w, h), td.BayerPattern.RGGB)
    if workload == 'rcd':
        return dtype, lambda bayer: rcd.process(bayer)
    if os.environ.get('TDK_BENCH_RCD_STANDALONE'):  # A/B knob (profiles/): the stand-alone strips (rs::rcd_stream) also with frames in flight
        from torch_darktable.torch_darktable_extension import concurrent_frames

        class _Plain:
            def process(self, x):
                with concurrent_frames(False):
                    return rcd_.process(x)

        rcd_, rcd = rcd, _Plain()
    wiener = td.Wiener(dev, (w, h), overlap_factor=4, tile_size=32)
    bilateral = td.Bilateral(dev, (w, h), sigma_s=2.0, sigma_r=0.2)
    params = td.TonemapParameters(gamma=0.75, intensity=2.0, light_adapt=1.0, vibrance=0.0)

    # Stage hand-over of the pipeline package (pipeline/image_processor.py process_rgb; results identical to the
    # separate wrapper calls, tests/test_gpu_fusion.py): the denoiser also writes the lightness plane of its result,
    # which Bilateral.process_rgb would extract first.  Per-frame statistics (moving_average = 1) through a
    # MetricsAccumulator == compute_image_metrics([rgb], stride=8).
    lum = torch.empty((h, w), dtype=torch.float32, device=dev)
    acc = td.tonemap.MetricsAccumulator(dev, stride=8)

    def frame_two_stage(bayer):  # --chain rgb: the intermediate RGB image between denoiser and local contrast is materialised
        rgb = rcd.process(bayer)
        rgb = wiener.process_log_luminance(rgb, 0.075, luminance_out=lum)
        rgb = bilateral.process_rgb(rgb, 0.4, luminance=lum, metrics=acc)
        return td.reinhard_tonemap(rgb, acc.finish(), params)

    # Lab hand-over (include/tdk_hip.h: tdk_wiener_log_luminance_lab / tdk_bilateral_lab): the two stages are two Lab round
    # trips of the same pixel in the reference (denoise.py:54-58, local_contrast.py:109-114); between them the pixel travels as
    # fp32 lightness plane + fp32 (a, b) plane and is converted back to RGB once.  Same results within the colour operators'
    # tolerance (tests/test_gpu_lab_chain.py), 34 transcendentals per pixel fewer.
    ab = torch.empty((h, w, 2), dtype=torch.float32, device=dev)

    def frame(bayer):
        rgb = rcd.process(bayer)
        wiener.process_log_luminance_lab(rgb, 0.075, luminance_out=lum, chroma_out=ab)
        rgb = bilateral.process_lab(lum, ab, 0.4, out_dtype=dtype, metrics=acc)
        return td.reinhard_tonemap(rgb, acc.finish(), params)

    return dtype, (frame_two_stage if chain == 'rgb' else frame)


def cpu_baseline(workload, threads, frame0, budget_s=25.0):
    """The oracle chain (a CPU port: the reference has no CPU path) on a bounded sample: full
    4096x3072 frames, one warm-up on a quarter-size frame, then timed
    full-size runs until three are done or the time budget is spent (at least one).  frame0 = the very samples GPU frame 0
    of rank 0 holds (copied back after the timed region, as float32; the device generator's noise differs from the CPU
    generator's, so the frame cannot be re-made on the host), or None: a host-made frame of the same kind.  The last
    result is returned for the parity check."""
    import numpy as np

    sys.path.insert(0, str(ROOT / 'oracle'))
    os.environ['OMP_NUM_THREADS'] = str(threads)
    import tdk_oracle as O
    from torch_darktable.synthetic import synthetic_bayer

    O.build()

    def run(bayer):
        if workload == 'ppg_wiener50':
            return O.wiener(O.ppg(bayer[:, :, 0] if bayer.ndim == 3 else bayer, O.RGGB), [0.05, 0.05, 0.05], 32, 4)
        rgb = O.rcd(bayer, O.RGGB)
        if workload == 'rcd':
            return rgb
        ll = O.compute_luminance(rgb, log=True, eps=1e-4)
        rgb = O.modify_luminance(rgb, O.wiener(ll[:, :, None], 0.075, 32, 4)[:, :, 0], log=True)
        rgb = O.modify_luminance(rgb, O.bilateral(O.compute_luminance(rgb), 2.0, 0.2, 0.4))
        m = O.image_metrics([rgb], 8)
        return O.tonemap('reinhard', rgb, m, 0.75, 2.0, 1.0, 0.0)

    # the CPU sample is a 12 MP frame for every workload (50 MP through the oracle takes minutes)
    bayer = frame0 if frame0 is not None and frame0.shape[:2] == (H12, W12) else synthetic_bayer(H12, W12, seed=1234, device='cpu').numpy()
    run(np.ascontiguousarray(bayer[:H12 // 2, :W12 // 2]))  # warm-up (page-in, OpenMP pool)
    times, t_start, result = [], time.perf_counter(), None
    while len(times) < 3 and (not times or time.perf_counter() - t_start < budget_s):
        t0 = time.perf_counter()
        result = run(bayer)
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    return {
        'value': round(W12 * H12 / 1e6 / dt, 3), 'unit': 'MP/s', 'cores': threads, 'kind': 'port',
        'sample': f'median of {len(times)} x one {W12}x{H12} RGGB frame through the strict-fp32 C oracle of the same chain (OpenMP, {threads} threads), '
                  'after one warm-up on a 2048x1536 crop; fp32 arithmetic on the samples GPU frame 0 read (the oracle has no fp16 storage mode)',
    }, result


def parity_check(workload, gpu_out, oracle_out):
    """GPU frame 0 of the last timed step against the oracle's result for the same input.  isp: uint8 output, bound = the
    one of tests/test_gpu_fullsize.py::test_full_pipeline_12mp_fp16_vs_fp32_oracle (+-2 LSB, > 1 LSB on <= 1e-5 of the values);
    rcd (fp32 storage): bit-exact."""
    import numpy as np

    got = gpu_out.cpu().numpy()
    if got.shape != oracle_out.shape:
        return {'ok': None, 'skipped': f'shapes differ: GPU {got.shape}, oracle sample {oracle_out.shape} (the CPU sample of this workload is not the GPU frame)'}
    if workload == 'isp':
        d = np.abs(got.astype(np.int32) - oracle_out.astype(np.int32))
        res = {'max_lsb': int(d.max()), 'frac_gt_1lsb': float((d > 1).mean()), 'frac_ne': float((d > 0).mean()),
               'bound': 'max_lsb <= 2 and frac_gt_1lsb <= 1e-5 (fp16 storage of three intermediates against the fp32 oracle; measured 8e-8)'}
        res['ok'] = bool(res['max_lsb'] <= 2 and res['frac_gt_1lsb'] <= 1e-5)
        return res
    if workload == 'rcd':
        bad = int((got != oracle_out.astype(got.dtype)).sum())
        return {'mismatches': bad, 'bound': 'bit-exact', 'ok': bad == 0}
    return {'ok': None, 'skipped': 'no full-frame oracle sample for this workload'}


def _git_head() -> str | None:
    """Short hash of the sources being measured: from git, or (GPU-box snapshots carry no .git) from the
    .git_head file a post-commit hook keeps next to the sources."""
    try:
        head = subprocess.run(['git', '-C', str(ROOT), 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True, timeout=5).stdout.strip()
    except Exception:  # noqa: BLE001
        head = ''
    if not head and (ROOT / '.git_head').exists():
        head = (ROOT / '.git_head').read_text().strip()
    return head or None


def run_stub(args, ranks: Ranks, pinned=None):
    """CPU rehearsal of the rank protocol (no HIP): same warm-up / barrier / timed region / max-over-ranks
    / one JSON line, with a pure-torch stand-in stage.  Used by tests/test_bench_launch.py only."""
    import torch

    from torch_darktable.synthetic import synthetic_bayer

    frames = args.frames or 2
    w, h = min(args.width or 64, 64), min(args.height or 64, 64)
    inputs = [synthetic_bayer(h, w, seed=1234 + ranks.rank * frames + i, device='cpu') for i in range(frames)]

    def step():
        return [torch.nn.functional.avg_pool2d(b.permute(2, 0, 1)[None], 3, 1, 1) for b in inputs]

    for _ in range(args.warmup):
        step()
    ranks.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    elapsed = time.perf_counter() - t0
    ranks.barrier()
    per_rank = ranks.gather(elapsed)
    ncpus = ranks.gather(float(len(os.sched_getaffinity(0))))
    if ranks.rank == 0:
        assert len(per_rank) == args.gpus
        worst = max(per_rank)
        print(json.dumps({
            'metric': 'STUB rehearsal of the bench protocol on the CPU (not a measurement)', 'value': round(frames * args.steps * ranks.world * w * h / 1e6 / worst, 3),
            'unit': 'MP/s', 'n_gpus': ranks.world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(worst / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'stub',
            'config': {'workload': 'stub', 'frames_per_gpu_per_step': frames}, 'ranks_ran': len(per_rank),
            'per_rank_MPps': [round(frames * args.steps * w * h / 1e6 / t, 3) for t in per_rank],
            'cpus_per_rank': [int(c) for c in ncpus], 'pinned': bool(pinned),
        }))
    ranks.close()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit('bench.py: --gpus must be >= 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # no launcher: start the N ranks ourselves, BEFORE anything in this process touches the GPU
        return launch_ranks(args, argv)

    pinned = pin_self()
    ranks = Ranks(args)
    if args.stub_cpu:
        run_stub(args, ranks, pinned)
        return 0

    import torch

    world, rank, local_rank = ranks.world, ranks.rank, ranks.local_rank
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    shared = os.environ.get('TDK_BENCH_SHARE_GPU') == '1'   # rehearsal of the rank protocol on a box with fewer GPUs than ranks
    if torch.cuda.device_count() <= local_rank and not shared:
        raise SystemExit(f'bench.py: rank {rank} needs cuda:{local_rank} but only {torch.cuda.device_count()} device(s) are visible')
    dev = torch.device('cuda', local_rank % torch.cuda.device_count() if shared else local_rank)
    torch.cuda.set_device(dev)

    import __graft_entry__

    __graft_entry__.ensure_built()
    import torch_darktable as td
    from torch_darktable import _native
    from torch_darktable.synthetic import synthetic_bayer

    storage = args.storage or ('f32' if args.workload == 'rcd' else 'f16')
    frames = args.frames or {'isp': 8, 'rcd': 1, 'ppg_wiener50': 4}[args.workload]
    w, h = (args.width or (W50 if args.workload == 'ppg_wiener50' else W12)), (args.height or (H50 if args.workload == 'ppg_wiener50' else H12))
    # The frames of a batch are independent: they are issued round-robin on `nstreams` HIP streams, each with its own
    # op workspaces, so that the tail of one frame's kernels (partially filled last rounds, 1-workgroup finish kernels)
    # overlaps the next frames' -- measured +13 % at 2 streams (profiles/streams_exp.py, round 2), +1.5 % more at 3 with round 4's
    # kernels (profiles/r04/experiments/streams_sweep.txt); 4 streams lose 3 %.
    nstreams = max(1, min(args.streams, frames))
    from torch_darktable.sharding import FrameStreams

    dtype, process = build_pipeline(td, dev, w, h, storage, args.workload, args.chain)
    runner = FrameStreams(dev, lambda: build_pipeline(td, dev, w, h, storage, args.workload, args.chain)[1], streams=nstreams)

    # device-resident synthetic inputs, per-frame seeds 1234 + i (distinct per rank)
    inputs = [synthetic_bayer(h, w, seed=1234 + rank * frames + i, device=dev).to(dtype) for i in range(frames)]
    torch.cuda.synchronize()

    def step_serial():  # every frame on the current stream: the per-kernel table comes from this (the kernels the timed region runs)
        import contextlib
        from torch_darktable.torch_darktable_extension import concurrent_frames
        out = None
        with (concurrent_frames() if nstreams > 1 else contextlib.nullcontext()):
            for b in inputs:
                out = process(b)
        return out

    last = [None]

    def step():  # the timed region synchronises the device after its K steps, so the streams are not joined per step
        last[0] = runner.issue(inputs)

    use_timer = not args.no_kernel_timer
    if args.graph:
        # one step = one replay of the batch captured as a HIP graph (FrameStreams.capture): the host launches one graph instead of
        # ~57 kernels.  Events cannot bracket a kernel inside a graph, so this mode carries no live roofline kernel time.
        use_timer = False
        captured = runner.capture(inputs)

        def step():  # noqa: F811
            last[0] = captured.replay()
    # Untimed profiling pass, BEFORE the warm-up (so that the W warm-up steps run directly into the timed region): every launch of
    # one step bracketed by events -> the per-kernel table and the dominant kernel.  In the timed region only THAT kernel keeps its
    # events (measured live, on its launch stream), so the throughput number is not taxed by ~100 event pairs per step.
    table, dom = {}, None
    if use_timer:
        step_serial()  # first touch: code objects, workspaces, the allocator's pools
        torch.cuda.synchronize()
        _native.profile_enable(True)
        step_serial()
        torch.cuda.synchronize()
        table = _native.profile_report()
        _native.profile_enable(False)
        dom_alone = max(table.items(), key=lambda kv: kv[1][1])[0]
        # the same with the frames on their streams, as the timed region issues them: a kernel's time now includes the company it
        # keeps on the CUs, and the kernel with the largest total here is the one whose live time the timed region measures
        table_live = table
        if nstreams > 1 and not args.graph:
            _native.profile_enable(True)
            step()
            torch.cuda.synchronize()
            table_live = _native.profile_report()
            _native.profile_enable(False)
        dom = max(table_live.items(), key=lambda kv: kv[1][1])[0]

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ranks.barrier()
    torch.cuda.synchronize()
    if use_timer:
        _native.profile_enable(True, only=dom)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t0
    ranks.barrier()
    report = _native.profile_report() if use_timer else {}
    if use_timer:
        _native.profile_enable(False)
    per_rank = ranks.gather(elapsed_local)
    elapsed = max(per_rank)

    if rank != 0:
        ranks.close()
        return 0
    if len(per_rank) != args.gpus:
        raise SystemExit(f'bench.py: {len(per_rank)} ranks ran but --gpus is {args.gpus}')

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
        alone_s = table[dom][1] / table[dom][0] / 1e3 if dom in table else avg_s  # the same kernel with the GPU to itself (untimed serial pass)
        stage, stage_bpp = stage_of(dom, args.workload, sbytes)
        planes = 3 if (args.workload == 'ppg_wiener50' and dom.startswith('tdk_wiener')) else 1  # C = 3: one launch covers the three planes
        alg_bytes = stage_bpp * w * h if stage_bpp else None
        achieved = alg_bytes / avg_s / 1e9 if alg_bytes else None
        lbpp = launch_bytes_per_px(dom, sbytes, args.chain)
        stage_kernels = stages(args.workload, sbytes)[stage][1] if stage else []
        stage_us = sum(table[k][1] / frames * 1e3 for k in stage_kernels if k in table)  # per frame, GPU to itself
        captured, valu, composite, why_not = None, None, None, None
        tfile = ROOT / 'profiles' / 'traffic.json'  # written by profiles/collect_traffic.py from rocprofv3 --pmc passes
        if not tfile.exists():
            why_not = 'profiles/traffic.json missing'
        elif (w, h, storage, args.workload) != (W12, H12, 'f16', 'isp'):
            why_not = 'the counters in profiles/traffic.json were captured on the isp workload at 12 MP fp16'
        else:
            tj = json.loads(tfile.read_text())
            if tj.get('_csrc_sha') != csrc_sha():
                why_not = (f"profiles/traffic.json was captured from other kernel sources (its csrc hash {tj.get('_csrc_sha')}, git {tj.get('_git')}; "
                           f'these sources {csrc_sha()}): counters not used')
                tj = None
        if why_not is None:
            def alu_floor_s(kernel, plain=VALU_CYCLES_PLAIN, trans=VALU_CYCLES_TRANS, packed=VALU_CYCLES_PACKED):
                """Issue floor of one launch from the captured counters: cycles per wave64 instruction by class."""
                insts = tj.get('_valu', {}).get(kernel)
                if not insts:
                    return None
                n_trans, n_packed = tj.get('_trans', {}).get(kernel) or 0, tj.get('_packed', {}).get(kernel) or 0
                cyc = (insts - n_trans - n_packed) * plain + n_trans * trans + n_packed * packed
                return cyc / (VALU_CUS * VALU_SIMDS * VALU_CLOCK_GHZ * 1e9)

            measured = dict(plain=VALU_MEASURED_PLAIN, trans=VALU_MEASURED_TRANS, packed=VALU_MEASURED_PACKED)
            captured = {'hbm_bytes_per_launch': tj.get(dom), 'git': tj.get('_git'), 'csrc_sha': tj.get('_csrc_sha'),
                        'source': 'profiles/traffic.json (rocprofv3 --pmc passes of these kernel sources, not this run)'}
            floor_s = alu_floor_s(dom)
            if floor_s:
                valu = {'wave_insts_per_launch': tj['_valu'][dom], 'transcendental_insts_per_launch': tj.get('_trans', {}).get(dom),
                        'packed_fp32_insts_per_launch': tj.get('_packed', {}).get(dom),
                        'achieved_Ginst_per_s': round(tj['_valu'][dom] / avg_s / 1e9, 1), 'peak_Ginst_per_s': round(VALU_PEAK_GINST, 1),
                        'alu_floor_us': round(floor_s * 1e6, 2), 'alu_floor_measured_us': round(alu_floor_s(dom, **measured) * 1e6, 2),
                        'issue_frac': round(floor_s / avg_s, 4), 'issue_frac_measured': round(alu_floor_s(dom, **measured) / avg_s, 4),
                        'prices': {'nominal_cycles': [VALU_CYCLES_PLAIN, VALU_CYCLES_PACKED, VALU_CYCLES_TRANS],
                                   'measured_cycles': [VALU_MEASURED_PLAIN, VALU_MEASURED_PACKED, VALU_MEASURED_TRANS], 'order': 'plain, packed fp32, transcendental'}}
            # composite floor of the whole frame: every kernel at the larger of its HBM time and its issue floor
            comp_us, comp_m_us, parts = 0.0, 0.0, {}
            for k, (c, ms_k) in table.items():
                kb = launch_bytes_per_px(k, sbytes, args.chain)
                t_hbm = (kb * w * h / (HBM_PEAK_GBS * 1e9)) if kb else 0.0
                t_alu, t_alu_m = alu_floor_s(k) or 0.0, alu_floor_s(k, **measured) or 0.0
                parts[k] = {'hbm_us': round(t_hbm * 1e6, 1), 'alu_floor_us': round(t_alu * 1e6, 1), 'alu_floor_measured_us': round(t_alu_m * 1e6, 1),
                            'measured_us': round(ms_k / c * 1e3, 1), 'launches_per_frame': c / frames}
                comp_us += max(t_hbm, t_alu) * c / frames * 1e6
                comp_m_us += max(t_hbm, t_alu_m) * c / frames * 1e6
            frame_us = elapsed / (frames * args.steps) * 1e6
            composite = {'composite_floor_us_per_frame': round(comp_us, 1), 'composite_floor_measured_prices_us_per_frame': round(comp_m_us, 1),
                         'measured_us_per_frame': round(frame_us, 1), 'frac_of_composite': round(comp_us / frame_us, 4),
                         'frac_of_composite_measured': round(comp_m_us / frame_us, 4), 'kernels': parts,
                         'note': 'sum over the kernels of a frame of max(HBM time of the launch bytes at 8 TB/s, issue floor from the captured '
                                 'counters); the second pair prices the instructions at what they cost on the box'}
        roofline = {
            'kernel': dom, 'bound': 'hbm', 'frac': round(achieved / HBM_PEAK_GBS, 5) if achieved else None,
            'frac_means': f"SURVEY.md 8(d) bytes of the '{stage}' stage ({stage_bpp} B/px x {w}x{h}) / this kernel's average launch time / 8 TB/s",
            'achieved': round(achieved, 2) if achieved else None, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'algorithmic_bytes_per_launch': int(alg_bytes) if alg_bytes else None, 'planes_per_launch': planes,
            'traffic': captured['hbm_bytes_per_launch'] if captured else None, 'traffic_captured': captured, 'traffic_unavailable': why_not,
            'avg_launch_us': round(avg_s * 1e6, 2), 'launches': cnt,
            # the same kernel with the GPU to itself: in the timed region frames on the other stream(s) share the CUs with it
            'avg_launch_us_alone': round(alone_s * 1e6, 2),
            'frac_alone': round(alg_bytes / alone_s / 1e9 / HBM_PEAK_GBS, 5) if alg_bytes else None,
            # the kernel's own compulsory bytes (it is one of several kernels of its stage)
            'launch_bytes': int(lbpp * w * h) if lbpp else None,
            'launch_bytes_frac': round(lbpp * w * h / avg_s / 1e9 / HBM_PEAK_GBS, 5) if lbpp else None,
            # the whole stage the kernel belongs to, per frame, kernels back to back
            'stage': {'name': stage, 'kernels': [k for k in stage_kernels if k in table], 'us_per_frame_alone': round(stage_us, 1),
                      'frac': round(alg_bytes / (stage_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5) if alg_bytes and stage_us else None},
            'kernel_launches_in_timed_region': launches_total, 'launches_per_frame': round(launches_total / (args.steps * frames), 2),
            'valu': valu, 'composite': composite,
            # which schedule each choice belongs to: `kernel` is the kernel with the largest total device time in an untimed pass
            # with the frames on their streams (what the timed region runs); `dominant_alone` the one of the serial one-stream pass
            'kernel_chosen_from': f'untimed pass with the frames on {nstreams} stream(s)',
            'dominant_alone': dom_alone,
            'kernel_ms_per_frame_streams': {k: round(ms / frames, 4) for k, (c, ms) in sorted(table_live.items(), key=lambda kv: -kv[1][1])},
        }
    # whole-pipeline roofline at the Python-wrapper stage boundaries (isp 41 B/px f16, 79 B/px f32; RCD 4 s; config 5 10 s)
    pipe_bpp = sum(b for b, _ in stages(args.workload, sbytes).values())
    pipe_gbs = pipe_bpp * w * h * total_frames / world / elapsed / 1e9

    metric = {'isp': 'megapixels/sec full ISP (debayer->denoise->tonemap) 12MP RGGB', 'rcd': 'megapixels/sec RCD demosaic 12MP RGGB',
              'ppg_wiener50': 'megapixels/sec PPG demosaic + Wiener C=3 denoise 50MP RGGB'}[args.workload]
    workload_text = {
        'isp': ('12 MP full pipeline (RCD -> Wiener log-L sigma=0.075 K=32 ov=4 -> bilateral sigma_s=2 sigma_r=0.2 detail=0.4 -> '
                'metrics -> Reinhard gamma=0.75 intensity=2 light_adapt=1 -> u8), batch 8 per GPU'),
        'rcd': '12 MP RCD demosaic, single frame',
        'ppg_wiener50': '50 MP PPG demosaic -> Wiener.process C=3 sigma=0.05 K=32 ov=4 (BASELINE config 5; the reference has no wavelet denoiser), batch 4 per GPU',
    }[args.workload]
    out = {
        'metric': metric,
        'value': round(value, 2), 'unit': 'MP/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {
            'workload': workload_text,
            'width': w, 'height': h, 'frames_per_gpu_per_step': frames, 'streams_per_gpu': nstreams, 'storage': storage, 'arithmetic': 'f32',
            'chain': args.chain if args.workload == 'isp' else None, 'issue': 'hip graph replay per step' if args.graph else 'eager launches',
            'denoiser': 'Wiener (the reference has no nlmeans / wavelet denoiser)', 'sharding': 'independent frames per GPU, no collective (gloo barrier + max of the timing only)'
            + (' -- REHEARSAL: ranks share GPUs (TDK_BENCH_SHARE_GPU=1), not a scaling measurement' if shared else ''),
            'cpus_pinned': pinned,
        },
        'ranks_ran': len(per_rank),
        'per_rank_MPps': [round(frames * args.steps * mp_per_frame / t, 1) for t in per_rank],
        'pipeline_roofline': {'algorithmic_bytes_per_px': pipe_bpp, 'achieved_GBps_per_gpu': round(pipe_gbs, 2), 'frac_of_8TBps': round(pipe_gbs / HBM_PEAK_GBS, 5)},
        'roofline': roofline,
        'kernel_ms_per_frame': stage_ms,
        'kernel_ms_per_frame_source': 'one untimed step, frames back to back on one stream, every launch bracketed by events; the roofline kernel is '
                                      'timed live in the timed region (where frames on other streams share the GPU with it)',
        'git': _git_head(),
    }
    rc = 0
    if world == 1 and not args.no_cpu_baseline:
        try:
            frame0 = inputs[0].float().cpu().numpy()  # what GPU frame 0 read (binary16 values for fp16 storage), after the timed region
            out['cpu_baseline'], oracle_out = cpu_baseline(args.workload, os.cpu_count() or 1, frame0)
            out['parity_check'] = parity_check(args.workload, last[0][0], oracle_out) if (w, h) == (W12, H12) else {
                'ok': None, 'skipped': 'the CPU sample is a 12 MP frame; this run used another size'}
            out['parity_check']['what'] = 'GPU frame 0 of the last timed step (seed 1234) against the CPU oracle on the same input'
            if out['parity_check'].get('ok') is False:
                print(f"bench.py: PARITY CHECK FAILED: {out['parity_check']}", file=sys.stderr)
                rc = 3
        except Exception as e:  # noqa: BLE001
            out['cpu_baseline'] = {'value': None, 'unit': 'MP/s', 'cores': 0, 'kind': 'port', 'sample': f'failed: {e}'}
            out['parity_check'] = {'ok': None, 'skipped': f'oracle run failed: {e}'}
    print(json.dumps(out))
    ranks.close()
    return rc


if __name__ == '__main__':
    sys.exit(main())

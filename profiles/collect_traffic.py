"""Turns rocprofv3 PMC passes into profiles/traffic.json: per library kernel, HBM-side bytes per launch
(FETCH_SIZE / WRITE_SIZE -- they do not fit one pass on gfx950), the vector-ALU instruction count
(wave-instructions per launch) under "_valu", its transcendental part under "_trans", and every SQ
counter of the passes under "_sq" (means per launch).

On the GPU box (profiles/capture.sh does all of it):
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 profiles/run_op.py isp --iters 3
  rocprofv3 --kernel-trace --pmc WRITE_SIZE ...   /   --pmc SQ_INSTS_VALU SQ_WAIT_ANY ...
  python3 profiles/collect_traffic.py $OUT/fetch $OUT/write $OUT/traffic.json [$OUT/valu [GIT_HASH [$OUT/sq2 ...]]]

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE count in units of 1 KiB; the
counters derive from TCC_EA0_RDREQ x 64 B, and on gfx950 a wide coalesced streaming read (16 B/lane) is
tallied at HALF its bytes, so the read side is doubled for kernels whose loads are 16-B (or, calibrated
below, 8-B) vector streams (marked WIDE); other access widths are reported uncorrected (uncalibrated).
WRITE_SIZE needs no correction.  Values are means over the launches seen.

Transcendentals: SQ_INSTS_VALU_TRANS_F32 when one of the passes collected it; otherwise the kernel's STATIC share
of v_exp/v_log/v_rcp/v_rsq/v_sqrt/v_sin/v_cos among its VALU instructions (from the code object's
disassembly, profiles/isa_mix.py) times the measured SQ_INSTS_VALU -- exact for straight-line streaming
kernels, an estimate for kernels with data-dependent branches.
"""
import csv
import glob
import json
import subprocess
import sys
from collections import defaultdict
from pathlib import Path

# kernel function name fragment -> (TDK_LAUNCH name used by bench.py, loads are wide vector streams)
KERNELS = {
    # tile kernel: every lane streams its own image row (16 B per load, 64 different rows per wave request) -> the
    # requests are line-sized, not wide; the finish kernels read slabs and image as coalesced 16-B / 8-B streams
    'wiener_stream': ('tdk_wiener(tiles)', False),
    # y-streaming tile kernel (K = 32, ov = 4): 16-B staging loads of in-frame groups, 64 consecutive bytes per 4 lanes
    'wiener_ystream': ('tdk_wiener(tiles)', True),
    'wiener_finish_modify': ('tdk_wiener(finish+modify)', True),
    'wiener_finish_lab': ('tdk_wiener(finish+lab)', True),     # Lab hand-over chain: slabs + (a, b) in, lightness out
    'lum_lab_extract': ('tdk_compute_luminance(lab)', True),   # RGB in, log-lightness + (a, b) planes out
    'slice_lab_kernel': ('tdk_bilateral(slice+lab)', False),
    'wiener_finish<': ('tdk_wiener(finish)', True),
    'rcd_interior': ('tdk_rcd', False),
    # column strips: one 4-B (fp16) / 8-B (fp32) sample pair per lane and step, 12-B / 24-B pixel pairs out
    'rcd_stream': ('tdk_rcd', False),
    'rcd_quad': ('tdk_rcd(concurrent)', False),  # register-blocked strips (TDK_RCD_CONCURRENT): same loads and stores
    'rcd_border': ('tdk_rcd(border)', False),
    'bilateral_tile_kernel': ('tdk_bilateral(tiles)', True),
    'bilateral_axis_tables_kernel': ('tdk_bilateral(tables)', False),
    'metrics_kernel': ('tdk_image_metrics', False),
    'metrics_finish_reset_kernel': ('tdk_image_metrics_finish', False),
    'splat_gather_kernel': ('tdk_bilateral(splat)', False),
    'blur_xy_kernel': ('tdk_bilateral(blur_xy)', False),
    'blur_z_kernel': ('tdk_bilateral(blur_z)', False),
    'slice_modify_kernel': ('tdk_bilateral(slice+modify)', False),
    'slice_kernel': ('tdk_bilateral(slice)', False),
    # 8-B / 16-B per-lane streaming loads: calibrated on these kernels' known byte counts (12 MP fp16:
    # tonemap reads 75.5 MB, FETCH_SIZE reported 37.8 MB) -> the factor 2 applies to them as well
    'tonemap_vec4': ('tdk_tonemap', True),
    # JPEG (run_op.py isp_jpeg): dword / byte image reads and one 128-byte coefficient line per lane -- no wide streams
    'jpeg_fdct_kernel': ('tdk_jpeg(fdct)', False),
    'jpeg_code_kernel<0>': ('tdk_jpeg(histogram)', False),
    'jpeg_code_kernel<1>': ('tdk_jpeg(lengths)', False),
    'jpeg_code_kernel<2>': ('tdk_jpeg(write)', False),
    'jpeg_stuff_kernel<false>': ('tdk_jpeg(count ff)', True),
    'jpeg_stuff_kernel<true>': ('tdk_jpeg(stuff)', True),
    'lum_modify_vec4': ('tdk_modify_luminance', True),
    'lum_extract_vec4': ('tdk_compute_luminance', True),
}


def mean_counters(directory):
    """{counter: {kernel fragment: mean value per launch}} for every counter found under `directory`."""
    files = glob.glob(f'{directory}/**/*_counter_collection.csv', recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for row in csv.DictReader(open(f)):
            for frag in KERNELS:
                if frag in row['Kernel_Name']:
                    acc[row['Counter_Name']][frag].append(float(row['Counter_Value']))
                    break
    return {c: {k: sum(v) / len(v) for k, v in per.items()} for c, per in acc.items()}


def static_trans_share(key='trans'):
    """{kernel fragment: share of its static VALU instructions that are transcendental (or `key`)}, via profiles/isa_mix.py."""
    try:
        out = subprocess.run([sys.executable, str(Path(__file__).with_name('isa_mix.py')), '--json'], capture_output=True, text=True, timeout=300)
        mix = json.loads(out.stdout)
    except Exception:  # noqa: BLE001
        return {}
    share = {}
    for frag in KERNELS:
        v = t = 0
        for name, m in mix.items():
            if frag in name:
                v += m['valu']
                t += m.get(key, 0)
        if v:
            share[frag] = t / v
    return share


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    sq_dirs = [d for i, d in enumerate(sys.argv[4:]) if i != 1]
    git = sys.argv[5] if len(sys.argv) > 5 else None
    fetch = mean_counters(fetch_dir).get('FETCH_SIZE', {})
    write = mean_counters(write_dir).get('WRITE_SIZE', {})
    sq = {}
    for d in sq_dirs:
        for c, per in mean_counters(d).items():
            sq.setdefault(c, {}).update(per)
    valu = sq.get('SQ_INSTS_VALU', {})
    share = static_trans_share()
    result, detail = {}, {}
    for frag, (name, wide) in KERNELS.items():
        if frag not in fetch and frag not in write:
            continue
        rd = fetch.get(frag, 0.0) * 1024.0 * (2.0 if wide else 1.0)
        wr = write.get(frag, 0.0) * 1024.0
        result[name] = int(rd + wr)
        detail[name] = {'FETCH_SIZE_raw': fetch.get(frag), 'WRITE_SIZE_raw': write.get(frag), 'read_bytes': int(rd), 'write_bytes': int(wr),
                        'read_doubled': wide}
    valu_named = {KERNELS[k][0]: int(v) for k, v in valu.items()}
    measured_trans = sq.get('SQ_INSTS_VALU_TRANS_F32', {})
    if measured_trans:   # the counter exists on gfx950: prefer it over the static share
        trans_named = {KERNELS[k][0]: int(v) for k, v in measured_trans.items()}
    else:
        trans_named = {KERNELS[k][0]: int(v * share[k]) for k, v in valu.items() if k in share}
    pshare = static_trans_share('packed')   # packed fp32 (two operations per lane, 4 SIMD cycles): static share x SQ_INSTS_VALU
    packed_named = {KERNELS[k][0]: int(v * pshare[k]) for k, v in valu.items() if pshare.get(k)}
    sq_named = {c: {KERNELS[k][0]: round(v, 1) for k, v in per.items()} for c, per in sq.items()}
    import hashlib
    h = hashlib.sha256()   # content hash of the kernel sources the counters belong to (bench.py csrc_sha): a capture is only
    for f in sorted((Path(__file__).resolve().parent.parent / 'torch-darktable_amd' / 'csrc').glob('*')):  # used for these sources
        if f.suffix in ('.hip', '.h'):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    json.dump({**result, '_valu': valu_named, '_trans': trans_named, '_packed': packed_named, '_sq': sq_named, '_detail': detail, '_git': git,
               '_csrc_sha': h.hexdigest()[:16],
               '_note': 'HBM-side bytes per launch; see profiles/collect_traffic.py for the corrections'}, open(out, 'w'), indent=1)
    print(json.dumps(result, indent=1))


if __name__ == '__main__':
    main()

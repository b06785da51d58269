"""Turns rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass on gfx950 -- and
optionally SQ_INSTS_VALU) into profiles/traffic.json: HBM-side bytes per launch for each library
kernel, plus its vector-ALU instruction count (wave-instructions per launch) under "_valu".

On the GPU box:
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/profiles/run_op.py isp --iters 3
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/profiles/run_op.py isp --iters 3
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d $OUT/valu -- python3 $REPO/profiles/run_op.py isp --iters 3
  python3 $REPO/profiles/collect_traffic.py $OUT/fetch $OUT/write $REPO/gpurun_out/traffic.json [$OUT/valu]

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB-like units of
1024 B... the counters derive from TCC_EA0_RDREQ x 64 B, and on gfx950 a wide coalesced streaming
read (16 B/lane) is tallied at HALF its bytes, so the read side is doubled for kernels whose loads
are 16-B vector loads (marked WIDE below); other access widths are reported uncorrected
(uncalibrated).  WRITE_SIZE needs no correction.  Values are means over the launches seen.
"""
import csv
import glob
import json
import sys
from collections import defaultdict

# kernel function name fragment -> (TDK_LAUNCH name used by bench.py, loads are wide 16-B streams)
KERNELS = {
    'wiener_tiles': ('tdk_wiener(tiles)', True),
    'wiener_finish_modify': ('tdk_wiener(finish+modify)', False),
    'wiener_finish<': ('tdk_wiener(finish)', False),
    'rcd_interior': ('tdk_rcd', False),
    'rcd_border': ('tdk_rcd(border)', False),
    'bilateral_tile_kernel': ('tdk_bilateral(tiles)', True),
    'metrics_kernel': ('tdk_image_metrics', False),
    'splat_gather_kernel': ('tdk_bilateral(splat)', False),
    'blur_xy_kernel': ('tdk_bilateral(blur_xy)', False),
    'blur_z_kernel': ('tdk_bilateral(blur_z)', False),
    'slice_modify_kernel': ('tdk_bilateral(slice+modify)', False),
    'slice_kernel': ('tdk_bilateral(slice)', False),
    # 8-B / 16-B per-lane streaming loads: calibrated on these kernels' known byte counts (12 MP fp16:
    # tonemap reads 75.5 MB, FETCH_SIZE reported 37.8 MB) -> the factor 2 applies to them as well
    'tonemap_vec4': ('tdk_tonemap', True),
    'lum_modify_vec4': ('tdk_modify_luminance', True),
    'lum_extract_vec4': ('tdk_compute_luminance', True),
}


def mean_counter(directory, counter):
    files = glob.glob(f'{directory}/**/*_counter_collection.csv', recursive=True)
    acc = defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] != counter:
                continue
            for frag in KERNELS:
                if frag in row['Kernel_Name']:
                    acc[frag].append(float(row['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    valu = mean_counter(sys.argv[4], 'SQ_INSTS_VALU') if len(sys.argv) > 4 else {}
    fetch = mean_counter(fetch_dir, 'FETCH_SIZE')
    write = mean_counter(write_dir, 'WRITE_SIZE')
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
    json.dump({**result, '_valu': valu_named, '_detail': detail, '_note': 'HBM-side bytes per launch; see profiles/collect_traffic.py for the corrections'},
              open(out, 'w'), indent=1)
    print(json.dumps(result, indent=1))


if __name__ == '__main__':
    main()

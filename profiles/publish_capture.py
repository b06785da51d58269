"""Copy the summaries of a capture (gpurun_out/<tag>/, written by profiles/capture.sh + the op benches) into profiles/<tag>/
and refresh profiles/traffic.json; condenses each PMC pass to one row per library kernel (means per launch).

  python profiles/publish_capture.py r02
"""
import collections
import csv
import glob
import shutil
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).parent))
from collect_traffic import KERNELS  # noqa: E402

tag = sys.argv[1]
src, dst = Path('gpurun_out') / tag, Path('profiles') / tag
dst.mkdir(exist_ok=True)
for name in ['bench_isp_plain.json', 'bench_rcd_plain.json', 'bench_ppg_wiener50_plain.json', 'bench_isp_streams1_plain.json', 'bench_isp_streams2_plain.json', 'bench_isp_under_rocprof.json', 'op_bench_f16.json', 'op_bench_f32.json', 'op_bench_50mp_f16.json',
             'laplacian_kernels.txt', 'traffic.json', 'bench_isp_repeats.txt', 'jpeg_bench.txt']:
    if (src / name).exists():
        shutil.copy(src / name, dst / name)
shutil.copy(src / 'stats' / 'bench_kernel_stats.csv', dst / 'bench_isp_kernel_stats.csv')
if (src / 'stats1' / 'bench_kernel_stats.csv').exists():
    shutil.copy(src / 'stats1' / 'bench_kernel_stats.csv', dst / 'bench_isp_streams1_kernel_stats.csv')
    shutil.copy(src / 'bench_isp_streams1_under_rocprof.json', dst / 'bench_isp_streams1_under_rocprof.json')
if (src / 'stats_jpeg' / 'jpeg_kernel_stats.csv').exists():
    shutil.copy(src / 'stats_jpeg' / 'jpeg_kernel_stats.csv', dst / 'jpeg_kernel_stats.csv')
shutil.copy(src / 'traffic.json', Path('profiles') / 'traffic.json')
for d in ['fetch', 'write', 'valu', 'sq2', 'sq3']:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f'{src}/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            for frag in KERNELS:
                if frag in r['Kernel_Name']:
                    acc[frag][r['Counter_Name']].append(float(r['Counter_Value']))
                    break
    counters = sorted({c for k in acc.values() for c in k})
    if not counters:
        continue
    with open(dst / f'pmc_{d}.csv', 'w') as out:
        out.write('kernel,launches,' + ','.join(counters) + '\n')
        for frag, cs in acc.items():
            n = len(next(iter(cs.values())))
            out.write(f'{KERNELS[frag][0]},{n},' + ','.join(f'{sum(cs[c]) / len(cs[c]):.6g}' if c in cs else '' for c in counters) + '\n')
print('published', tag)

"""Which kernels of the multi-stream bench run at the same time: from a rocprofv3 --kernel-trace CSV.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ov -o t -- python3 bench.py --no-cpu-baseline --steps 5
    python profiles/overlap_trace.py gpurun_out/ov
Prints per kernel: launches, mean live duration, and the share of its live time during which a kernel of each other name was
also running; then the share of the traced span during which 0 / 1 / 2 / 3+ kernels ran."""
import collections
import csv
import glob
import sys


def short(n):
    for k in ('rcd_quad', 'rcd_stream', 'wiener_ystream', 'wiener_finish_modify', 'bilateral_tile', 'bilateral_axis', 'tonemap', 'lum_extract', 'metrics_kernel'):
        if k in n:
            return k
    return n[:40]


def main():
    rows = []
    for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
    rows.sort()
    # keep the steady part: drop the first and last 15 % of the launches
    n = len(rows)
    rows = rows[int(n * 0.15):int(n * 0.85)]
    names = sorted({r[2] for r in rows})
    dur = collections.defaultdict(list)
    ov = collections.defaultdict(lambda: collections.defaultdict(int))
    for i, (s, e, k) in enumerate(rows):
        dur[k].append(e - s)
        for j in range(max(0, i - 40), min(len(rows), i + 40)):
            if j == i:
                continue
            s2, e2, k2 = rows[j]
            o = min(e, e2) - max(s, s2)
            if o > 0:
                ov[k][k2] += o
    for k in names:
        tot = sum(dur[k])
        print(f'{k:22s} n={len(dur[k]):4d} live {tot / len(dur[k]) / 1e3:7.1f} us | with: ' + ' '.join(f'{k2}={ov[k][k2] / tot:.2f}' for k2 in names if ov[k][k2] > 0.005 * tot))
    ev = []
    for s, e, _ in rows:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    level, last, hist = 0, ev[0][0], collections.Counter()
    for t, d in ev:
        hist[min(level, 3)] += t - last
        last = t
        level += d
    span = sum(hist.values())
    print('kernels running at once: ' + ' '.join(f'{k}{"+" if k == 3 else ""}: {v / span:.3f}' for k, v in sorted(hist.items())))
    print(f'span {span / 1e6:.2f} ms, kernel time summed {sum(sum(v) for v in dur.values()) / 1e6:.2f} ms')


if __name__ == '__main__':
    main()

"""Per-op table of the `uir_gemm` launches of a train step from a bench.py --breakdown-file CSV.
usage: python tools/gemm_shapes.py <op_breakdown.csv> > profiles/rNN/gemm1x1_shapes.md
One row per distinct op (block index wild-carded): launches per step, serial us per launch, algorithmic MB per launch, TB/s, lane; the
rows x K -> N shape where the op name determines it (B = 16, 256 x 256)."""
import collections
import re
import sys


def shape(n):
    M0 = 16 * 256 * 256
    lv = {'encoders.0': (M0, 16), 'encoders.1': (M0 // 4, 32), 'encoders.2': (M0 // 16, 64), 'encoders.3': (M0 // 64, 128), 'decoders.0': (M0 // 16, 64),
          'decoders.1': (M0 // 4, 32), 'decoders.2': (M0, 32), 'refinement': (M0, 32)}
    for k, (M, C) in lv.items():
        if n.startswith(k):
            h = int(C * 2.66)
            hp = (h + 15) // 16 * 16
            t = {'dqk': (M, 2 * C, 2 * C), 'dt1': (M, 3 * C, C), 'dt2': (M, 2 * hp, C), 'dt1+ln1_bwd': (M, 3 * C, C), 'dt2+ln2_bwd': (M, 2 * hp, C), 'dvf': (M, C, C),
                 'dv': (M, C, C), 'dgate': (M, C, hp), 'P': (C, C, C), 'dWkvf': (C, C, C)}
            return t.get(n.split('.')[-1])
    return None


def main():
    rows = [l.strip().split(',') for l in open(sys.argv[1])]
    start = [i for i, r in enumerate(rows) if r and r[0] == 'op'][0]
    agg = collections.OrderedDict()
    for r in rows[start + 1:]:
        if len(r) < 6 or r[1] != 'uir_gemm' or '+dW' in r[0]:
            continue
        name, ms, mb, lane = r[0], float(r[2]), float(r[3]), r[4]
        key = (re.sub(r'^(encoders|decoders|refinement)\.(\d)(\.\d)?\.', r'\1.\2.*.', name), shape(name))
        a = agg.setdefault(key, [0, 0.0, 0.0, lane])
        a[0] += 1
        a[1] += ms
        a[2] += mb
    tot = sum(v[1] for v in agg.values())
    print(f"# The plain `uir_gemm` launches of one train step, per distinct op ({sum(v[0] for v in agg.values())} launches, {tot:.2f} ms serial)\n")
    print("From a `bench.py --breakdown-file` CSV (`tools/gemm_shapes.py`): serial HIP-event time per launch; GEMMs with a fused Gram (`+dW`) are listed in the breakdown "
          "itself.  Kernels behind the entry point: `uir_gemm1x1_kernel` (panel), `uir_ts_split_kernel` (4 096 - 65 536 rows, K, N >= 64), `uir_conv3x3_*`.\n")
    print("| op | rows x K -> N | launches | us / launch | MB / launch | TB/s | lane |\n|---|---|---|---|---|---|---|")
    for (name, sh), (n, ms, mb, lane) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        us, m = ms / n * 1e3, mb / n
        print(f"| {name} | {'%d x %d -> %d' % sh if sh else ''} | {n} | {us:.1f} | {m:.1f} | {m / us if us else 0:.2f} | {lane} |")


if __name__ == "__main__":
    main()

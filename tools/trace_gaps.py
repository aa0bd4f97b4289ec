"""Idle time on the main queue of a train step, from a rocprofv3 --kernel-trace CSV.
usage: python tools/trace_gaps.py <kernel_trace.csv> [steps=3] [who_us=0]
The last `steps` steps are delimited by the optimizer kernel (uir_adam*); per step: busy and idle time of the queue that runs the most
kernels (the chain), the idle time by preceding kernel, and how much of the wall the other queues were busy.
who_us > 0: for every chain gap longer than that in the LAST step, the chain kernels on both sides and every kernel of another queue
that overlaps the gap (start / end relative to the gap's start / end) - the producer the chain waited for ends at the gap's end."""
import csv
import collections
import re
import sys


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    rows.sort()
    byq = collections.Counter(q for _, _, _, q in rows)
    mainq = byq.most_common(1)[0][0]
    print("queues (kernels):", dict(byq), "-> chain queue", mainq)
    ends = [i for i, r in enumerate(rows) if "adam" in r[2].lower() or "optim" in r[2].lower()]
    if len(ends) < steps + 1:
        print("not enough optimizer launches found", len(ends))
        return
    lo, hi = ends[-steps - 1] + 1, ends[-1]
    sel = rows[lo:hi + 1]
    wall = sel[-1][1] - sel[0][0]
    chain = [r for r in sel if r[3] == mainq]
    busy = sum(e - s for s, e, _, _ in chain)
    gaps = collections.defaultdict(lambda: [0, 0])
    tot_gap = 0
    hist = collections.Counter()
    for a, b in zip(chain, chain[1:]):
        g = b[0] - a[1]
        if g > 0:
            tot_gap += g
            k = re.sub(r"<.*", "", a[2].replace("(anonymous namespace)::", "").replace("void ", ""))
            gaps[k][0] += g
            gaps[k][1] += 1
            hist[min(int(g / 1000), 20)] += 1
    other = [r for r in sel if r[3] != mainq]
    print(f"{steps} steps: wall {wall / steps / 1e6:.3f} ms/step; chain queue busy {busy / steps / 1e6:.3f} ms, idle {tot_gap / steps / 1e6:.3f} ms "
          f"({len(chain) // steps} launches/step); other queues busy {sum(e - s for s, e, _, _ in other) / steps / 1e6:.3f} ms ({len(other) // steps} launches)")
    print("idle after kernel (ms/step, launches/step, us per gap):")
    for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"  {k[:60]:60s} {g / steps / 1e6:7.3f} {n // steps:5d} {g / n / 1e3:7.2f}")
    print("gap histogram (us: count/step):", {k: v // steps for k, v in sorted(hist.items())})
    who = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    if who > 0:
        def nm(s):
            return re.sub(r"\(.*", "", s.replace("(anonymous namespace)::", "").replace("void ", ""))[:56]
        last = rows[ends[-2] + 1:ends[-1] + 1]
        lchain = [r for r in last if r[3] == mainq]
        lother = [r for r in last if r[3] != mainq]
        t0 = last[0][0]
        print(f"gaps > {who} us of the last step (time since its first kernel):")
        for a, b in zip(lchain, lchain[1:]):
            g = (b[0] - a[1]) / 1e3
            if g <= who:
                continue
            print(f"  t={(a[1] - t0) / 1e6:7.3f} ms  gap {g:6.1f} us  after {nm(a[2])} -> next {nm(b[2])}")
            for o in lother:
                if o[0] < b[0] and o[1] > a[1]:
                    print(f"        q{o[3]} {nm(o[2])}  start {(o[0] - a[1]) / 1e3:+8.1f}  end {(o[1] - b[0]) / 1e3:+8.1f}  ({(o[1] - o[0]) / 1e3:.1f} us)")


if __name__ == "__main__":
    main()

"""Idle time of the chain (main stream) of the headline step WITHOUT the profiler: HIP event pairs around every launch on the stream it runs on
(Recorder.run(timed=...): the real three-lane execution), gaps = time between the end event of a chain launch and the start event of the next
one.  rocprofv3 --kernel-trace slows the host down enough (the step's 743 launches take 3.1 ms to enqueue untraced) that the deep levels'
short kernels outrun it: gaps in a traced run can be the host's, not a dependency's.  Here the host stays tens of ms ahead.
usage: python tools/chain_gaps_events.py [min_us=12] [passes=3]
Prints per pass the chain's busy / idle time inside the forward and backward lists and every gap above min_us with the launches on both
sides and the side / bulk launches that END inside the gap (a producer the chain waited for ends at the gap's end)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    min_us = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from underwater_image_restoration_amd import init_model
    from underwater_image_restoration_amd.trainer import TrainStep
    torch.manual_seed(1234)
    model = init_model("SpectralTransformer").to(dev)
    step = TrainStep(model, "L1withColor", lr=3e-4)
    g = torch.Generator(device="cpu").manual_seed(1000)
    raw = (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev)
    ref = (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev)
    for _ in range(3):
        step(raw, ref)
    torch.cuda.synchronize()
    plan = model._plan_for(raw)
    for p in range(passes):
        for tag, ops in (("fwd", plan.fwd), ("bwd", plan.bwd)):
            timed = []
            origin = torch.cuda.Event(enable_timing=True)
            origin.record()
            plan.run(ops, timed=timed)
            torch.cuda.synchronize()
            lanes = plan.lanes[id(ops)]
            rows = sorted((origin.elapsed_time(e0) * 1e3, origin.elapsed_time(e1) * 1e3, i) for i, e0, e1 in timed)   # us
            chain = [r for r in rows if lanes[r[2]] == 0]
            other = [r for r in rows if lanes[r[2]] != 0]
            busy = sum(e - s for s, e, _ in chain)
            gaps = [(b[0] - a[1], a, b) for a, b in zip(chain, chain[1:])]
            idle = sum(max(0.0, g_) for g_, _, _ in gaps)
            small = sorted(g_ for g_, _, _ in gaps if g_ <= min_us)
            med = small[len(small) // 2] if small else 0.0
            print(f"pass {p} {tag}: chain {len(chain)} launches, busy {busy / 1e3:.3f} ms, between launches {idle / 1e3:.3f} ms "
                  f"(median gap {med:.1f} us = the event pair's own cost; {sum(1 for g_, _, _ in gaps if g_ > min_us)} gaps > {min_us} us "
                  f"= {sum(g_ for g_, _, _ in gaps if g_ > min_us) / 1e3:.3f} ms); other lanes {len(other)} launches {sum(e - s for s, e, _ in other) / 1e3:.3f} ms")
            if other and chain:
                # what the list's closing join waits for: the other lanes' last end against the chain's (the weight-gradient lane's tail)
                last_c, last_o = max(e for _, e, _ in chain), max(other, key=lambda r: r[1])
                n_after = sum(1 for s, e, _ in other if e > last_c)
                print(f"         tail: the other lanes end {last_o[1] - last_c:+.1f} us after the chain's last launch ({n_after} launches end behind it; last: {ops[last_o[2]][2]})")
            if p == passes - 1:
                for g_, a, b in gaps:
                    if g_ > min_us:
                        print(f"   gap {g_:7.1f} us  after {ops[a[2]][2]} -> next {ops[b[2]][2]}")
                        for s, e, i in other:
                            if a[1] - 2 <= e <= b[0] + 2:
                                print(f"         lane {lanes[i]} {ops[i][2]:40s} start {s - a[1]:+8.1f}  end {e - b[0]:+8.1f} us rel. to the gap's start / end")


if __name__ == "__main__":
    main()

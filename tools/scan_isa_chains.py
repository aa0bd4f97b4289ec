"""Scan the gfx950 code objects of csrc/_build/*.o for SERIALISED round trips the source did not ask for (CPU only).

Three shapes, each counted as its longest run of consecutive repeats per kernel:
  vm   [<= 3 global loads / stores] s_waitcnt vmcnt(0)                - an exposed HBM / L2 round trip per repeat: loads the compiler
       sank next to their consumers or wrapped in exec branches, and STORES under divergent branches (vmcnt retires in issue order
       and the compiler cannot count operations issued under a branch, so every later wait becomes vmcnt(0) and waits for the
       previous store's acknowledgement too)
  lds  [1-3 ds_read] s_waitcnt lgkmcnt(0) [1-2 MFMA]                  - an exposed LDS round trip per MFMA (pair): operand fetches
       the compiler moved from the batch the source issued them in to just before their consumer
  st   s_waitcnt vmcnt(0) directly in front of a global_store, repeated
Found this way in round 4: uir_dwln_bwd's pixel-contraction phase (32 LDS round trips per chunk), its LayerNorm-backward
epilogue (16 store round trips per tile) and its xhat prologue (4 load round trips per tile).
    python tools/scan_isa_chains.py [--min 4] [--show KERNEL_SUBSTRING]
"""
import argparse
import glob
import os
import re
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from underwater_image_restoration_amd.build import OBJ  # noqa: E402
from scan_isa_waits import disassemble, kernels  # noqa: E402


def classify(l):
    m = re.match(r"\s+([a-z_0-9]+)", l)
    if not m:
        return None
    o = m.group(1)
    if "mfma" in o:
        return "M"
    if o.startswith(("global_load", "buffer_load")):
        return "g"
    if o.startswith(("global_store", "buffer_store")):
        return "S"
    if o.startswith("ds_read") or o.startswith("ds_bpermute"):
        return "r"
    if o == "s_waitcnt":
        vm = re.search(r"vmcnt\((\d+)\)", l)
        lg = re.search(r"lgkmcnt\((\d+)\)", l)
        if vm and int(vm.group(1)) == 0:
            return "V"
        if lg and int(lg.group(1)) == 0:
            return "L"
        return None
    if o == "s_barrier":
        return "B"
    return "."


def runs(body):
    seq = [c for c in (classify(l) for l in body) if c]
    s = "".join(seq)
    s_nodot = s.replace(".", "")
    def longest(pat, text):
        best = 0
        for m in re.finditer("(?:%s)+" % pat, text):
            n = len(re.findall(pat, m.group(0)))
            best = max(best, n)
        return best
    vm = longest(r"[gS]{1,3}V", s_nodot.replace("r", "").replace("L", "").replace("M", "").replace("B", ""))
    # one LDS wait per one or two MFMAs, operands fetched one to three reads at a time (a wait-free unit may sit between two waited ones)
    t = s_nodot.replace("g", "").replace("S", "").replace("V", "")
    lds = 0
    for m in re.finditer(r"(?:r{1,3}L?M{1,2})+", t):
        lds = max(lds, m.group(0).count("L"))
    st = longest(r"V\.{0,8}S", s.replace("r", "").replace("L", ""))
    return vm, lds, st, s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min", type=int, default=4)
    ap.add_argument("--show", default="")
    a = ap.parse_args()
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(OBJ, "*.o"))):
            for name, body in kernels(disassemble(obj, tmp)):
                vm, lds, st, s = runs(body)
                rows.append((max(vm, lds, st), vm, lds, st, os.path.basename(obj), name))
                if a.show and a.show in name:
                    print(name)
                    print(re.sub(r"\.{3,}", lambda m: ".%d " % len(m.group(0)), s))
    print(f"{'vm':>4} {'lds':>4} {'st':>4}  object        kernel")
    for worst, vm, lds, st, obj, name in sorted(rows, reverse=True):
        if worst >= a.min:
            print(f"{vm:4d} {lds:4d} {st:4d}  {obj:12s}  {name[:150]}")


if __name__ == "__main__":
    main()

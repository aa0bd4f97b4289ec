"""Scan the gfx950 code objects of csrc/_build/*.o for memory loads the compiler serialised.

For every kernel: the `s_waitcnt vmcnt(N)` instructions in program order, each with the number of global / buffer loads issued since
the previous one.  A healthy software pipeline reads  (16 loads, vmcnt 15) (0, 13) (0, 11) ...: a batch in flight, drained while
the consumers run.  A chain of  (1 load, vmcnt 0) (1, 0) (1, 0) ...  is one exposed memory round trip per load - what hipcc
produces when it sinks each load of an operand array next to the MFMA that consumes it, or wraps a `cond ? load : 0` select of a
conditionally executed loop body in its own exec branch (the split-bf16 contraction, the window-attention
forward, the DFT kernel).  Prints the kernels whose longest such chain is >= --min (default 4), longest first.

    python tools/scan_isa_waits.py [--min 4] [--show KERNEL_SUBSTRING]      (CPU only: disassembles, runs nothing)
"""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from underwater_image_restoration_amd.build import LLVM_BIN, OBJ  # noqa: E402


def disassemble(obj, tmp):
    fat, co = os.path.join(tmp, "x.fat"), os.path.join(tmp, "x.co")
    heads = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-h", obj], check=True, capture_output=True, text=True).stdout
    if ".hip_fatbin" not in heads:
        return ""
    subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", obj], check=True)
    subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={fat}", f"--output={co}"], check=True)
    return subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout


def kernels(dis):
    lines = dis.split("\n")
    idx = [i for i, l in enumerate(lines) if l.endswith(">:")] + [len(lines)]
    for a, b in zip(idx[:-1], idx[1:]):
        yield lines[a].split("<")[-1][:-2], [l for l in lines[a:b] if l.startswith("\t")]


def wait_profile(body):
    loads = [i for i, l in enumerate(body) if "global_load" in l or "buffer_load" in l]
    prof, prev = [], 0
    for i, l in enumerate(body):
        m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)
        if m:
            prof.append((sum(1 for x in loads if prev <= x < i), int(m.group(1))))
            prev = i
    return len(loads), prof


def longest_chain(prof):
    best = run = 0
    for n, v in prof:
        if 1 <= n <= 2 and v == 0:
            run += 1
            best = max(best, run)
        elif n != 0:
            run = 0
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min", type=int, default=4)
    ap.add_argument("--show", default="", help="print the whole wait profile of the kernels whose (mangled) name contains this")
    args = ap.parse_args()
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(OBJ, "*.o"))):
            for name, body in kernels(disassemble(obj, tmp)):
                nloads, prof = wait_profile(body)
                if args.show and args.show in name:
                    print(name, f"loads {nloads}", prof)
                c = longest_chain(prof)
                if c >= args.min:
                    rows.append((c, nloads, sum('v_mfma' in l for l in body), os.path.basename(obj), name))
    print("longest chain of (load, vmcnt(0)) | loads | MFMAs | object | kernel")
    for r in sorted(rows, reverse=True):
        print(*r)


if __name__ == "__main__":
    main()

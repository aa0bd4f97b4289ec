"""Build the four code objects of the interference probe from old_dwconv_wgrad.hip (hipcc cross-compiles without a GPU):

  A  the kernel as hipcc emits it: 72 packed FMAs per row, two of them `v_pk_fma_f32 ... op_sel:[0,1,0] op_sel_hi:[0,0,1]`
  B  those two instructions as two scalar `v_fma_f32` each (same arithmetic, everything else byte-identical)
  C  those two instructions kept, `s_nop 7` before and after each (wait-state theory)
  D  the halves of their src1 pre-swapped by two `v_mov_b32` into v[248:249], then the packed FMA in the `op_sel_hi:[0,1,1]`
     form the other 70 use (packed form kept, half-swap removed)

Outputs variant_{A,B,C,D}.hsaco next to this file (git-ignored; they travel to the GPU box with the tree)."""
import os, re, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "old_dwconv_wgrad_kernel"
SWAP = re.compile(r"v_pk_fma_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[\1:\2\] op_sel:\[0,1,0\] op_sel_hi:\[0,0,1\]")


def main():
    s_path = os.path.join(HERE, "_old.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                    os.path.join(HERE, "old_dwconv_wgrad.hip"), "-o", s_path], check=True)
    src = open(s_path).read()
    n_swapped = len(SWAP.findall(src))
    n_packed = len(re.findall(r"v_pk_fma_f32", src))
    m = re.search(r"\.amdhsa_next_free_vgpr (\d+)", src)
    nv = int(m.group(1))
    print(f"packed FMAs {n_packed}, half-swapped {n_swapped}, VGPRs {nv}")
    assert n_swapped == 2, "the compiler no longer emits the two half-swapped packed FMAs: the probe would test nothing"

    def scalar(mm):      # D.lo = S0.lo * S1.hi + S2.lo ; D.hi = S0.lo * S1.lo + S2.hi
        d0, d1, a0, _a1, b0, b1 = mm.groups()
        return f"v_fma_f32 v{d0}, v{a0}, v{b1}, v{d0}\n\tv_fma_f32 v{d1}, v{a0}, v{b0}, v{d1}"

    def padded(mm):
        return "s_nop 7\n\t" + mm.group(0) + "\n\ts_nop 7"

    t0, t1 = nv, nv + 1
    assert nv % 2 == 0 and nv + 2 <= 256

    def preswap(mm):
        d0, d1, a0, a1, b0, b1 = mm.groups()
        return (f"v_mov_b32_e32 v{t0}, v{b1}\n\tv_mov_b32_e32 v{t1}, v{b0}\n\t"
                f"v_pk_fma_f32 v[{d0}:{d1}], v[{a0}:{a1}], v[{t0}:{t1}], v[{d0}:{d1}] op_sel_hi:[0,1,1]")

    variants = {"A": src, "B": SWAP.sub(scalar, src), "C": SWAP.sub(padded, src), "D": SWAP.sub(preswap, src)}
    d = variants["D"]
    for old, new in ((f".amdhsa_next_free_vgpr {nv}", f".amdhsa_next_free_vgpr {nv + 2}"), (f".amdhsa_accum_offset {nv}", f".amdhsa_accum_offset {(nv + 2 + 3) // 4 * 4}"),
                     (f"{KERNEL}.num_vgpr, {nv}", f"{KERNEL}.num_vgpr, {nv + 2}"), (f".vgpr_count:     {nv}", f".vgpr_count:     {nv + 2}")):
        assert old in d, old
        d = d.replace(old, new)
    variants["D"] = d
    for k, text in variants.items():
        sp = os.path.join(HERE, f"_variant_{k}.s")
        open(sp, "w").write(text)
        obj = os.path.join(HERE, f"_variant_{k}.o")
        subprocess.run([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", sp, "-o", obj], check=True)
        subprocess.run([f"{LLVM}/ld.lld", "-shared", obj, "-o", os.path.join(HERE, f"variant_{k}.hsaco")], check=True)
        os.remove(obj)
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", os.path.join(HERE, f"variant_{k}.hsaco")], capture_output=True, text=True).stdout
        left = len(re.findall(r"v_pk_fma_f32.*op_sel:\[0,1,0\]", dis))
        print(f"variant {k}: half-swapped packed FMAs in the code object: {left}")
    if "--keep-asm" not in sys.argv:
        for k in variants:
            os.remove(os.path.join(HERE, f"_variant_{k}.s"))
        os.remove(s_path)


if __name__ == "__main__":
    main()

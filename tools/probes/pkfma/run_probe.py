"""Interference probe: which instruction of the old depthwise weight-gradient kernel loses an FMA next to the
split-bf16 weight-gradient contraction?  Four code objects of the SAME kernel (make_variants.py) that differ only in the two
half-swapped packed FMAs run, one after the other, beside the same aggressor on another stream; every run is compared bit for bit
with the kernel's result alone.

    python tools/probes/pkfma/make_variants.py            (build container)
    python tools/probes/pkfma/run_probe.py [iters=200] [aggressor=split|f32]

The aggressor is the product's own uir_gram on csrc/tiled.hip (split-bf16 with UIR_TILED_SPLIT=1, set here)."""
import ctypes as C, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
AGG = sys.argv[2] if len(sys.argv) > 2 else "split"
if AGG == "split":
    os.environ["UIR_TILED_SPLIT"] = "1"
import torch
from underwater_image_restoration_amd.engine import Recorder
from underwater_image_restoration_amd import _lib as L

hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


def chk(rc, what):
    if rc:
        raise RuntimeError(f"{what}: hip error {rc}")


dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
B, H, W, Cc = 1, 32, 32, 1024
M = B * H * W
r1 = Recorder(dev, scratch_floats=4 << 20)
dY, X, dW = rnd(M, Cc), rnd(M, Cc), r1.alloc(Cc, 9)
r1.dwconv_wgrad("wg", dY, X, dW, C_=Cc, creal=Cc, dims=(B, H, W))     # builds the argument struct + the slab reduction
(fn0, args0, _), red_ops = r1.fwd[0], r1.fwd[1:]
a = args0[0]._obj                                                        # L.DwWgradArgs
HS = 2                                                                   # uir_dw_strip(1, 32, 32, 1024)
c4n, ny = Cc // 4, 1
lds = ny * c4n * 36 * 4
hs = C.c_int(HS)
params = (C.c_void_p * 2)(C.cast(C.pointer(a), C.c_void_p), C.cast(C.pointer(hs), C.c_void_p))


def load(variant):
    mod, fn = C.c_void_p(), C.c_void_p()
    chk(hip.hipModuleLoad(C.byref(mod), os.path.join(HERE, f"variant_{variant}.hsaco").encode()), "hipModuleLoad")
    chk(hip.hipModuleGetFunction(C.byref(fn), mod, b"old_dwconv_wgrad_kernel"), "hipModuleGetFunction")
    return fn


def victim(fn):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    chk(hip.hipModuleLaunchKernel(fn, a.nblocks, 1, 1, c4n, ny, 1, lds, st, params, None), "hipModuleLaunchKernel")
    for rfn, rargs, rname in red_ops:          # the slab reduction of the recorder, on the same stream
        L.check(rfn(*rargs, st), rname)


r2 = Recorder(dev, scratch_floats=4 << 20)
Mg, K, N = 16384, 512, 2048
xa, ya, dwa = rnd(Mg, K), rnd(Mg, N), r2.alloc(N, K)
r2.gram("g", ya, xa, dwa, N=N, K=K, o_sn=K, o_sk=1, own_slab=True)
ops2 = r2.fwd
s2 = torch.cuda.Stream()
print(f"aggressor: uir_gram {Mg} x {N} x {K} ({'split-bf16 v_mfma_f32_32x32x16_bf16' if AGG == 'split' else 'f32 v_mfma_f32_32x32x2_f32'}), "
      f"victim grid {a.nblocks} x ({c4n},{ny}), {ITERS} runs per variant", flush=True)
for v in "ABCD":
    fn = load(v)
    victim(fn); torch.cuda.synchronize()
    ref = dW.clone()
    alone = 0
    for it in range(20):                       # the variant alone: must be bitwise stable
        dW.zero_(); victim(fn); torch.cuda.synchronize()
        alone += int(not torch.equal(dW, ref))
    bad, lanes, comps, taps = 0, set(), set(), set()
    for it in range(ITERS):
        dW.zero_()
        with torch.cuda.stream(s2):
            r2.run(ops2)
        victim(fn)
        torch.cuda.synchronize()
        if not torch.equal(dW, ref):
            bad += 1
            d = (dW - ref).abs()
            idx = d.nonzero()
            for c, t in idx.tolist():
                lanes.add((c // 4) % 64 // 16); comps.add("xyzw"[c % 4]); taps.add(t)
    print(f"variant {v}: alone {alone}/20 differ; beside the aggressor {bad}/{ITERS} differ"
          + (f"  [16-lane rows {sorted(lanes)}, components {sorted(comps)}, taps {sorted(taps)}]" if bad else ""), flush=True)

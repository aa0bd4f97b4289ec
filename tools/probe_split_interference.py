"""Reproducer of the interference recorded in DESIGN.md section 17: does a victim kernel (default: uir_dwconv_wgrad - set UIR_DW_WGRAD_OLD=1 for
its first version, the one that is disturbed: 248 VGPRs, 36 KB LDS, VALU-bound, ~100 v_mov per row) give different results when a tiled
contraction runs beside it on another stream?
    UIR_DW_WGRAD_OLD=1 UIR_TILED_SPLIT=1 python tools/probe_split_interference.py dw     # split-bf16 weight-gradient contraction: ~185 of 200 differ
    UIR_TILED_SPLIT=1 python tools/probe_split_interference.py dw          # the current uir_dwconv_wgrad beside it: 0 of 200
    UIR_TILED_SPLIT=1 python tools/probe_split_interference.py fwd | dx    # split-bf16 forward / input-gradient: 0 of 200
    python tools/probe_split_interference.py dw                            # f32-MFMA contraction (the default): 0 of 200
    ... dw colsum | gram | dwfwd | dwbwd                                   # other victims: 0 of 200
In every differing run exactly one FMA (one pixel x one tap x one float4 component) of one 16-lane slice of one workgroup is missing."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd.engine import Recorder
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
B, H, W, C = 1, 32, 32, 1024
M = B * H * W
r1 = Recorder(dev, scratch_floats=4 << 20)
dY, X, dW = rnd(M, C), rnd(M, C), r1.alloc(C, 9)
victim = sys.argv[2] if len(sys.argv) > 2 else "wgrad"
if victim == "wgrad":
    r1.dwconv_wgrad("wg", dY, X, dW, C_=C, creal=C, dims=(B, H, W))
elif victim == "colsum":
    dW = r1.alloc(256)
    r1.colsum("cs", dY, 0, 256, dW)
elif victim == "gram":
    dW = r1.alloc(64, 64)
    r1.gram("gr", dY, X, dW, N=64, K=64, o_sn=64, o_sk=1)
elif victim == "dwbwd":          # the fused depthwise backward (input gradient + weight-gradient partials, ~240 VGPRs)
    dW = r1.alloc(C, 9)
    Wt, dXo = rnd(C, 9), r1.alloc(M, C)
    os.environ["UIR_FUSED_DW_MINROWS"] = "1"
    r1.dwconv_bwd("dwb", dY, X, Wt, dXo, dW, C_=C, creal=C, dims=(B, H, W))
elif victim == "dwfwd":
    dW = r1.alloc(M, C)
    Wt = rnd(C, 9)
    r1.dwconv("dw", dY, Wt, dW, C_=C, dims=(B, H, W))
ops1 = r1.fwd
r2 = Recorder(dev, scratch_floats=4 << 20)
Mg, K, N = 16384, 512, 2048
xa, wa, ya = rnd(Mg, K), rnd(N, K) * .05, r2.alloc(Mg, N)
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
if which == "fwd":
    r2.gemm("g", xa, wa, ya, N=N, K=K, w_sn=K, w_sk=1)
elif which == "dx":
    dxa = r2.alloc(Mg, K)
    r2.gemm("g", ya, wa, dxa, N=K, K=N, w_sn=1, w_sk=K)
else:
    dwa = r2.alloc(N, K)
    r2.gram("g", ya, xa, dwa, N=N, K=K, o_sn=K, o_sk=1, own_slab=True)
ops2 = r2.fwd
r1.run(ops1); torch.cuda.synchronize()
ref = dW.clone()
s2 = torch.cuda.Stream()
bad = 0
for it in range(200):
    dW.zero_()
    with torch.cuda.stream(s2):
        r2.run(ops2)
    r1.run(ops1)
    torch.cuda.synchronize()
    if not torch.equal(dW, ref):
        bad += 1
        d = (dW - ref).abs()
        chs = d.amax(dim=1).nonzero().flatten().tolist()
        print("iteration", it, "max diff", float(d.max()), "channels", chs[:20], flush=True)
        if bad <= 3 and victim == "wgrad":
            c = chs[0]
            print("   ch", c, "got", [round(v, 4) for v in dW[c].flatten().tolist()], "\n          ref", [round(v, 4) for v in ref[c].flatten().tolist()], flush=True)
            # per-pixel decomposition of the centre tap for this channel: which pixel's product is missing?
            diff = float(dW[c].flatten()[4] - ref[c].flatten()[4])
            prod = (dY[:, c] * X[:, c]).cpu()
            near = (prod - (-diff)).abs().argmin() if diff else None
            print("   centre-tap diff", diff, "closest single-pixel product (negated):", None if near is None else (int(near), float(prod[near])), flush=True)
print("mismatching iterations:", bad, "of 200")

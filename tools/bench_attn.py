import sys, os, time, ctypes, torch
sys.path.insert(0, os.getcwd())
from underwater_image_restoration_amd import _lib as L
lib = L.lib
dev = "cuda"
for C, heads in ((32, 1), (128, 8), (64, 4), (16, 1)):
    B = 16; d = C // heads
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
    G = rnd(B, C, C); sq = rnd(B, C).abs() + 1; sk = rnd(B, C).abs() + 1; T = rnd(heads).abs() + .5
    P = rnd(C, C); Wof = rnd(C, C); dWA = rnd(B, C, C); dWB = rnd(B, C, C)
    attn = torch.empty(B, heads, d, d, device=dev); WA = torch.empty(B, C, C, device=dev); WB = torch.empty(B, C, C, device=dev)
    Mqk = torch.empty(B, 2 * C, 2 * C, device=dev); dP = torch.empty(B, C, C, device=dev); dWof = torch.empty(B, C, C, device=dev); dT = torch.empty(B, heads, device=dev)
    fa = L.AttnFwdArgs(); fa.G, fa.sq_q, fa.sq_k, fa.temperature, fa.P, fa.Wof = G.data_ptr(), sq.data_ptr(), sk.data_ptr(), T.data_ptr(), P.data_ptr(), Wof.data_ptr()
    fa.attn, fa.WA, fa.WB = attn.data_ptr(), WA.data_ptr(), WB.data_ptr(); fa.B, fa.C, fa.heads = B, C, heads
    ba = L.AttnBwdArgs(); ba.dWA, ba.dWB, ba.G, ba.sq_q, ba.sq_k, ba.temperature = dWA.data_ptr(), dWB.data_ptr(), G.data_ptr(), sq.data_ptr(), sk.data_ptr(), T.data_ptr()
    ba.attn, ba.P, ba.Wof = attn.data_ptr(), P.data_ptr(), Wof.data_ptr(); ba.Mqk, ba.dP, ba.dWof, ba.dT = Mqk.data_ptr(), dP.data_ptr(), dWof.data_ptr(), dT.data_ptr()
    ba.B, ba.C, ba.heads = B, C, heads
    big = torch.empty(64 << 20, device=dev)
    for name, fn, arg in (("fwd", lib.uir_attn_fwd, fa), ("bwd", lib.uir_attn_bwd, ba)):
        for cold in (0, 1):
            ts = []
            for _ in range(20):
                if cold: big.add_(1.0)          # evict caches (incl. instruction cache lines in L2)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(ctypes.byref(arg), None); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            print(f"C={C} heads={heads} {name} cold={cold}: median {ts[10]:.1f} us  min {ts[0]:.1f}")

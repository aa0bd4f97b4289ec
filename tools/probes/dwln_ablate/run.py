"""Time the ablation variants built by make.py (B = 16, 256 x 256, C = 32, GDFN layout)."""
import ctypes as C, glob, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
import torch
from underwater_image_restoration_amd.engine import Recorder, hidden_of, pad16
from underwater_image_restoration_amd import _lib as L
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
B, H, W, Cc = 16, 256, 256, 32
M = B * H * W
h, hp = hidden_of(Cc), pad16(hidden_of(Cc))
cols, chreal = 2 * hp, 2 * h
r = Recorder(dev, scratch_floats=1 << 20)
G, XP = rnd(M, cols), rnd(M, cols)
X, dRes, stats = rnd(M, Cc), rnd(M, Cc), torch.stack([rnd(M) * 0.1, 1 + 0.1 * rnd(M)], 1).contiguous()
Wdw, W1, gam, bet = rnd(chreal, 9), rnd(chreal, Cc), rnd(Cc), rnd(Cc)
dX, dWdw, dW1, dgb = r.alloc(M, Cc), r.alloc(chreal, 9), r.alloc(chreal, Cc), r.alloc(2 * Cc)
r.dwln_bwd("k", G, XP, Wdw, W1, X, stats, gam, bet, dRes, dX, dWdw, dW1, dgb[:Cc], dgb[Cc:], C_=Cc, dims=(B, H, W), h=h, hp=hp)
_fn, args, _ = r.fwd[0]
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for path in sorted(glob.glob(os.path.join(HERE, "libdwln_*.so"))):
    lib = C.CDLL(path)
    fn = lib.uir_dwln_bwd
    fn.argtypes = [C.POINTER(L.DwLnBwdArgs), C.c_void_p]
    for _ in range(3):
        assert fn(*args, sp) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn(*args, sp)
    e1.record()
    torch.cuda.synchronize()
    print(f"{os.path.basename(path):24s} {e0.elapsed_time(e1) * 50:8.1f} us", flush=True)

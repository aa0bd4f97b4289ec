"""Level-0 hot kernels in isolation (B=16, 256x256 -> M = 1,048,576 HWC rows), for rocprofv3.
usage: python tools/microbench.py [reps] [only-substring]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from underwater_image_restoration_amd import _lib as L          # noqa: E402
from underwater_image_restoration_amd.engine import Recorder    # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
only = sys.argv[2] if len(sys.argv) > 2 else ""
dev = "cuda"
B, H, W, C = 16, 256, 256, 32
M = B * H * W
h, hp = 85, 96
r = Recorder(dev, scratch_floats=64 << 20)
g = torch.Generator(device=dev).manual_seed(0)


def rnd(*s):
    return torch.rand(*s, device=dev, generator=g) * 2 - 1


x, x1 = rnd(M, C), rnd(M, C)
qkv, qkv_pre = rnd(M, 3 * C), rnd(M, 3 * C)
gbuf, gpre = rnd(M, 2 * hp), rnd(M, 2 * hp)
for t in (gbuf, gpre):
    t[:, h:hp] = 0
    t[:, hp + h:] = 0
y = r.alloc(M, C)
stats = r.alloc(M, 2)
Wqkv, Win, Wout = rnd(3 * C, C) * 0.2, rnd(2 * h, C) * 0.2, rnd(C, h) * 0.1
Wdw = rnd(2 * h, 9) * 0.3
WB = rnd(B, C * C) * 0.2
g1, b1 = rnd(C), rnd(C)
cases = {}


def case(name):
    def deco(fn):
        if only in name:
            r.fwd = []
            r.cur = r.fwd
            r.meta[id(r.fwd)] = []
            r.lanes[id(r.fwd)] = []
            fn()
            cases[name] = (r.fwd, r.meta[id(r.fwd)])
        return fn
    return deco


@case("gemm_qkv_ln_k32_n96")
def _():
    r.gemm("qkv", x, Wqkv, qkv_pre, N=3 * C, K=C, w_sn=C, w_sk=1, pro=L.PRO_LN, ln=(g1, b1), stats=stats)


@case("gemm_ffn_in_ln_k32_n176")
def _():
    r.gemm("ffn_in", x1, Win, gpre, N=2 * hp, K=C, w_sn=C, w_sk=1, n_grp=(h, hp), pro=L.PRO_LN, ln=(g1, b1), stats=stats)


@case("gemm_ffn_out_gate_k85_n32")
def _():
    r.gemm("ffn_out", gbuf, Wout, y, N=C, K=h, w_sn=h, w_sk=1, pro=L.PRO_GATE, gate_off=hp, R=x1)


@case("gemm_dgate_k32_n88")
def _():
    r.gemm("dgate", x, Wout, gpre, N=hp, K=C, w_sn=1, w_sk=h, n_grp=(h, hp), epi=L.EPI_GATE_BWD, G=gbuf, g_off=hp)


@case("gemm_dt2_k176_n32")
def _():
    r.gemm("dt2", gpre, Win, y, N=C, K=2 * hp, w_sn=1, w_sk=C, k_grp=(h, hp))


@case("gemm_WB_perbatch_k32_n32_res")
def _():
    r.gemm("WB", x, WB, y, N=C, K=C, w_sn=C, w_sk=1, w_bs=C * C, batches=B, R=x1)


@case("gemm_plain_k32_n32")
def _():
    r.gemm("plain", x, WB, y, N=C, K=C, w_sn=C, w_sk=1)


@case("gram_dWin_ln_n176_k32")
def _():
    r.gram("dWin", gpre, x1, r.alloc(2 * h, C), N=2 * hp, K=C, o_sn=C, o_sk=1, n_grp=(h, hp), pro_b=L.PRO_LN, ln=(g1, b1), stats=stats)


@case("gram_dWout_gate_n32_k85")
def _():
    r.gram("dWout", x, gbuf, r.alloc(C, h), N=C, K=h, o_sn=h, o_sk=1, pro_b=L.PRO_GATE, gate_off=hp)


@case("gram_dWqkv_ln_n96_k32")
def _():
    r.gram("dWqkv", qkv_pre, x, r.alloc(3 * C, C), N=3 * C, K=C, o_sn=C, o_sk=1, pro_b=L.PRO_LN, ln=(g1, b1), stats=stats)


@case("gram_qk_perbatch_n32_k32")
def _():
    gb = r.alloc(B * C * C + 2 * B * C)
    r.gram("qk", qkv[:, :C], qkv[:, C:2 * C], gb[:B * C * C], N=C, K=C, o_sn=C, o_sk=1, o_bs=C * C, batches=B,
           out_elems=B * C * C, extras={"sq_a": gb[B * C * C:B * C * C + B * C], "sq_b": gb[B * C * C + B * C:]})


@case("gram_conv3x3_outputl_n8_k32")
def _():
    r.gram("outputl.dW", rnd(M, 8), x, r.alloc(8, C, 3, 3), N=8, K=C, o_sn=C * 9, o_sk=9, o_st=1, taps=9, hw=(H, W))


@case("dwconv_c176")
def _():
    r.dwconv("ffn_conv", gpre, Wdw, gbuf, C_=2 * hp, dims=(B, H, W), grp=(h, hp))


@case("dwconv_c96")
def _():
    r.dwconv("qkv_conv", qkv_pre, rnd(3 * C, 9), qkv, C_=3 * C, dims=(B, H, W))


@case("dwconv_wgrad_c176")
def _():
    r.dwconv_wgrad("dWffdw", gbuf, gpre, r.alloc(2 * h, 9), C_=2 * hp, creal=2 * h, dims=(B, H, W), grp=(h, hp))


@case("ln_bwd_c32")
def _():
    r.ln_bwd("ln", x, x1, stats, g1, y, r.alloc(M, C), r.alloc(C), r.alloc(C), C_=C)


# stats must be valid for the LN Gram prologue
stats[:, 0] = 0.0
stats[:, 1] = 1.0
torch.cuda.synchronize()
for name, (ops, meta) in cases.items():
    r.run(ops)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r.run(ops)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    nb = sum(m[2] for m in meta)
    fl = sum(m[3] for m in meta)
    print(f"{name:34s} {ms:8.3f} ms  {nb / 1e6:9.1f} MB  {nb / ms / 1e6:8.1f} GB/s  {fl / ms / 1e9:7.2f} TFLOP/s", flush=True)

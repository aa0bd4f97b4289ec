"""Microbench of MDTA's value-path backward at the headline's sizes: the three-launch path (dvf GEMM + fused Gram, depthwise backward, dv GEMM + fused Gram)
against csrc/attnvb.hip's one launch.   usage: python tools/bench_value_bwd.py"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import _lib as L          # noqa: E402
from underwater_image_restoration_amd import engine as E        # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    for name, B, S, Cc in (("L0 C=32", 16, 256, 32), ("L0 C=16", 16, 256, 16), ("L1 C=32", 16, 128, 32), ("L2 C=64", 16, 64, 64)):
        M, HW = B * S * S, S * S
        dims = (B, S, S)
        r = E.Recorder(dev)
        dx1, vf, u = torch.randn(M, Cc, device=dev), torch.randn(M, Cc, device=dev), torch.randn(M, Cc, device=dev)
        qkv = torch.randn(M, 3 * Cc, device=dev)
        v_ = qkv[:, 2 * Cc:]
        WA, WB = torch.randn(B, Cc * Cc, device=dev) * Cc ** -0.5, torch.randn(B, Cc * Cc, device=dev) * Cc ** -0.5
        wdw = torch.randn(Cc, 9, device=dev)
        nab = B * Cc * Cc
        dvf, du, dqkv, dwd = r.alloc(M, Cc), r.alloc(M, Cc), r.alloc(M, 3 * Cc), r.alloc(Cc, 9)
        if Cc <= 32:
            nch = max(1, min(768 // B, HW // 128))
            slab = r.alloc(nch, 2 * nab)
            r.gemm(name + " dvf+dWB", dx1, WB, dvf, N=Cc, K=Cc, w_sn=1, w_sk=Cc, w_bs=Cc * Cc, batches=B, epi=L.EPI_PLAIN_DW, G=vf, dw_slab=slab, dw_cols=Cc, dw_rows=nch)
            r.dwconv_bwd(name + " du+dWkvdw", dvf, u, wdw, du, dwd, C_=Cc, creal=Cc, dims=dims)
            r.gemm(name + " dv+dWA", du, WA, dqkv[:, 2 * Cc:], N=Cc, K=Cc, w_sn=1, w_sk=Cc, w_bs=Cc * Cc, batches=B, epi=L.EPI_PLAIN_DW, G=v_, dw_slab=slab[:, nab:], dw_cols=Cc,
                   dw_rows=nch)
        else:       # the deep levels' five launches (engine.py _block: shared slab, split-bf16 Grams)
            tn_ = (Cc + 31) // 32
            nch, _ = r._chunks(HW, ((tn_ + 1) // 2) ** 2, B, split_nk=r.gram_split_case(HW, Cc, Cc))
            slab = r.alloc(nch * 2 * nab)
            dWBA = r.alloc(2 * nab)
            r.gram(name + " dWB", dx1, vf, dWBA[:nab].view(B, Cc * Cc), N=Cc, K=Cc, o_sn=Cc, o_sk=1, o_bs=Cc * Cc, batches=B, out_elems=nab, shared_slab=(slab, 0, 2 * nab))
            r.gemm(name + " dvf", dx1, WB, dvf, N=Cc, K=Cc, w_sn=1, w_sk=Cc, w_bs=Cc * Cc, batches=B)
            r.dwconv_bwd(name + " du+dWkvdw", dvf, u, wdw, du, dwd, C_=Cc, creal=Cc, dims=dims)
            r.gram(name + " dWA", du, v_, dWBA[nab:].view(B, Cc * Cc), N=Cc, K=Cc, o_sn=Cc, o_sk=1, o_bs=Cc * Cc, batches=B, out_elems=nab, shared_slab=(slab, nab, 2 * nab))
            r.gemm(name + " dv", du, WA, dqkv[:, 2 * Cc:], N=Cc, K=Cc, w_sn=1, w_sk=Cc, w_bs=Cc * Cc, batches=B)
        rows = L.lib.uir_attnv_bwd_rows_c(S, S, Cc)
        slab2, dqkv2, dwd2 = r.alloc(rows, 2 * nab), r.alloc(M, 3 * Cc), r.alloc(Cc, 9)
        r.attnv_bwd(name + " fused", dx1, WA, WB, wdw, u, vf, v_, dqkv2[:, 2 * Cc:], slab2, dwd2, C_=Cc, dims=dims)
        r.flush_reductions("t")
        sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for fn, args, label in r.fwd:
            if fn is None:
                continue
            for _ in range(3):
                assert fn(*args, sp) == 0, label
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn(*args, sp)
            e1.record()
            torch.cuda.synchronize()
            print(f"{label:28s} {e0.elapsed_time(e1) * 1000 / 20:8.1f} us", flush=True)
        torch.cuda.synchronize()
        err = float((dqkv2[:, 2 * Cc:] - dqkv[:, 2 * Cc:]).abs().max() / dqkv[:, 2 * Cc:].abs().max())
        print(f"   fused vs three launches: dv {err:.2e}", flush=True)


if __name__ == "__main__":
    main()

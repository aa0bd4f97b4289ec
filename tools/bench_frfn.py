"""Measure the FRFN feed-forward (reference src/model/block.py:249-282) at BASELINE configs[3]'s first
level: dim 32, hidden 128, 256 x 256 tokens, batch 8 per GPU, forward + backward (input and all parameter gradients).
ONE JSON line:

  value        images/s (forward + backward of one FRFN over a resident batch)
  roofline     both ceilings, as the path is composed of HBM-bound passes today:
               bound "mfma": algorithmic FLOPs (3 x the forward contractions) / time vs the dense fp32 MFMA peak 157.3 TFLOP/s
               hbm_frac:     the launch plan's algorithmic bytes (every operand of every launch once) / time vs 8 TB/s
  cpu_baseline the oracle (torch CPU restatement pinned to the reference's own outputs) on the host cores, bounded sample
"""
import argparse
import glob
import json
import os
import sys
import time

import torch

for _i, _a in enumerate(sys.argv):          # the library reads UIR_MIXED once per process: set it before anything launches
    if _a == "--mixed" and _i + 1 < len(sys.argv):
        os.environ["UIR_MIXED"] = sys.argv[_i + 1]
    elif _a.startswith("--mixed="):
        os.environ["UIR_MIXED"] = _a.split("=", 1)[1]

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd.model.block import FRFN  # noqa: E402
from underwater_image_restoration_amd.model.model import DecoderBlock, EncoderBlock  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MFMA_F32_PEAK = 157.3e12
HBM_PEAK = 8.0e12


def bench_net(args):
    """BASELINE configs[3] on one GPU: --arch NewBigFRFNModel, 256 x 256, bs 8, Charbonnier loss, full train step (forward, loss,
    backward, clip 1.0, Adam) through trainer.ModuleTrainStep.  value = images/s; roofline: the plans' algorithmic bytes / flops.
    --what ast: BASELINE configs[4]'s network on one GPU: --arch AST, 512 x 512, bs 4, fp32 (the reference has no AMP path:
    no autocast / GradScaler anywhere in src/), Charbonnier (fflMix needs a VGG16 download and two un-vendored packages)."""
    import contextlib
    from underwater_image_restoration_amd.Models import init_model
    from underwater_image_restoration_amd.trainer import make_train_step
    arch = "AST" if args.what == "ast" else "NewBigFRFNModel"
    mixed = os.environ.get("UIR_MIXED", "0")
    mixed = "" if mixed in ("", "0") else mixed
    cfg = ("configs[4], " + (f"MIXED {mixed}: wide contractions with {mixed} operands, f32 accumulation and master weights" if mixed else "fp32")) if arch == "AST" \
        else ("configs[3]" + (f", MIXED {mixed}" if mixed else ""))
    if arch == "AST" and (args.batch, args.size) == (8, 256):
        args.batch, args.size = 4, 512
    B, S = args.batch, args.size
    torch.manual_seed(1234)
    with contextlib.redirect_stdout(sys.stderr):
        model = init_model(arch).cuda()
    model.train()
    vgg_note = None
    if args.loss in ("perceptual", "mix", "bigMix", "fflMix", "fflMixTV"):
        from underwater_image_restoration_amd.Losses import perceptual
        try:
            perceptual._find_weights()
            vgg_note = "VGG16 weights: the user's torchvision file"
        except NotImplementedError:            # timing only: the arithmetic does not depend on the values
            from oracle import perceptual_ref
            perceptual.set_vgg16_weights(perceptual_ref.random_vgg16_state(0))
            vgg_note = "VGG16 weights: RANDOM values in torchvision's layout (no pretrained file on this box) - a timing of the loss's arithmetic, not a training result"
    step = make_train_step(model, args.loss, 3e-4)
    g = torch.Generator().manual_seed(1000)
    raw = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).cuda()
    ref = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).cuda()
    for _ in range(args.warmup):
        loss, norm = step(raw, ref)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, norm = step(raw, ref)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    plans = [p for m in model.plan_modules() for p in m._plans.values()]
    nbytes, flops, launches = sum(p.alg_bytes() for p in plans), sum(p.flops() for p in plans), sum(len(p.fwd) + len(p.bwd) for p in plans)
    lname = "Charbonnier" if args.loss == "charbonnier" else args.loss
    line = {"metric": f"train images/sec, {arch} {S}x{S} bs={B}/GPU, {lname} (BASELINE {cfg}, one GPU)",
            "value": B / (ms * 1e-3), "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "dtype": f"{mixed}-mixed" if mixed else "f32", "data": "synthetic",
            "config": {"workload": f"{arch} {S}x{S} bs={B}, {args.loss}, clip 1.0 + Adam 3e-4, DropPath as the reference sets it" + (f"; {vgg_note}" if vgg_note else ""),
                       "params": sum(p.numel() for p in model.parameters()), "launches_per_step": launches,
                       "final_loss": float(loss), "grad_norm": float(norm),
                       "plan_memory_GB": sum(p.bytes for p in plans) / 1e9},
            "roofline": {"bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": nbytes / (ms * 1e-3) / HBM_PEAK, "traffic": None, "plan_algorithmic_bytes": nbytes,
                         "plan_flops": flops, "mfma_frac_fp32": flops / (ms * 1e-3) / MFMA_F32_PEAK,
                         "note": "bytes / flops as the launch plans count them (every launch's operands once); the network is a chain of "
                                 "separate HBM-bound launches this round"}}
    # HBM bytes per step from PMC passes of THIS command (tools/round_end.sh -> tools/pmc_traffic.py), quoted only while the kernels
    # and launch plans still hash to what was measured
    tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"*{args.what}_hbm_counters.json")))
    if tfiles and not mixed and (B, S) == ((4, 512) if arch == "AST" else (8, 256)):
        from underwater_image_restoration_amd import build as _build
        with open(tfiles[-1]) as tf:
            tj = json.load(tf)
        rel = os.path.relpath(tfiles[-1], ROOT)
        if tj.get("source_sha") == _build.source_sha():
            line["roofline"]["traffic"] = tj["hbm_bytes_per_step"]
            line["roofline"]["traffic_source"] = rel + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2)"
        else:
            line["roofline"]["traffic_note"] = f"{rel} was measured on other sources ({tj.get('source_sha')}): stale, not quoted"
    if args.breakdown_file:
        # every module's launch lists timed one launch at a time (HIP event pairs, serial): where a step's kernel time is,
        # module by module; the overlapped step is shorter than the sum (weight-gradient lane)
        named = {id(m): n for n, m in model.named_modules()}
        with open(args.breakdown_file, "w") as f:
            f.write(f"# {arch} {S}x{S} bs={B}: step {ms:.2f} ms; serial per-launch times, one pass\n")
            f.write("module,fwd_ms,bwd_ms,launches\n")
            allrows = []
            for m in model.plan_modules():
                for key, p in m._plans.items():
                    rf, rb = p.run_timed(p.fwd), p.run_timed(p.bwd)
                    lf = list(p.lanes[id(p.fwd)]) + list(p.lanes[id(p.bwd)])
                    mn = named.get(id(m), type(m).__name__) or type(m).__name__
                    f.write(f"{mn}{list(key[:3])},{sum(r[4] for r in rf):.3f},{sum(r[4] for r in rb):.3f},{len(rf) + len(rb)}\n")
                    allrows += [(mn, key[1], r, ln) for r, ln in zip(rf + rb, lf)]
            agg = {}
            for mn, side, (name, sym, nb, fl, t), ln in allrows:
                a = agg.setdefault((sym, side, ln), [0, 0.0, 0, 0])
                a[0] += 1; a[1] += t; a[2] += nb; a[3] += fl
            f.write("\nsymbol,token grid side,lane,launches,ms,alg_GB,GB/s,TFLOP/s\n")
            for (sym, side, ln), (c, t, nb, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                f.write(f"{sym},{side},{ln},{c},{t:.3f},{nb / 1e9:.3f},{nb / max(t, 1e-9) / 1e6:.0f},{fl / max(t, 1e-9) / 1e9:.1f}\n")
            f.write("\nmodule,op,symbol,lane,ms,alg_MB,GFLOP\n")
            for mn, side, (name, sym, nb, fl, t), ln in sorted(allrows, key=lambda r: -r[2][4])[:150]:
                f.write(f"{mn},{name},{sym},{ln},{t:.4f},{nb / 1e6:.1f},{fl / 1e9:.2f}\n")
    if not args.no_cpu:
        from oracle import decoder_ref as D, ast_ref as OA
        net_fn = OA.ast_net if arch == "AST" else D.frfn_net
        cores = min(16, len(os.sched_getaffinity(0)))
        torch.set_num_threads(cores)
        model.eval()
        P = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        xb, pb = raw[:1].cpu(), ref[:1].cpu()
        t0 = time.perf_counter()
        D.grads(lambda t, Pp: net_fn(t, Pp), [xb], P, pb)
        dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": 1.0 / dt, "unit": "images/s", "cores": cores, "kind": "port",
                                "sample": f"1 forward+backward of one {S}x{S} image through oracle/{net_fn.__module__.split('.')[-1]}.{net_fn.__name__} (torch CPU fp32, {cores} threads)"}
    print(json.dumps(line))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=32)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--breakdown-file", default="", help="--what net / ast: per-module and per-launch serial times (csv)")
    ap.add_argument("--mixed", default="", choices=("", "bf16", "fp16"),
                    help="--what net / ast: UIR_MIXED for this process - wide contractions with 16-bit operands, f32 accumulation (its own metric, never the f32 one)")
    ap.add_argument("--loss", default="charbonnier", help="loss name of the network benches (--what net / ast), e.g. fflMix = BASELINE configs[4]'s")
    ap.add_argument("--what", default="frfn", choices=("frfn", "encoder", "decoder", "net", "ast"),
                    help="frfn: one FRFN (block.py:249-282); encoder: one EncoderBlock = LN + 2 FRFN + 2 (L,C) transforms (model.py:16-93); "
                         "decoder: one DecoderBlock(dim, enc_out=True) = 2 LN + MDASSA (2 window attentions + FDFP) + FRFN(2 dim) + proj (model.py:96-160)")
    args = ap.parse_args()
    if args.what in ("net", "ast"):
        return bench_net(args)
    dim, hid, B, S = args.dim, 4 * args.dim, args.batch, args.size
    torch.manual_seed(0)
    enc, dec = args.what == "encoder", args.what == "decoder"
    if dec:
        m = DecoderBlock(dim, (S, S), 4, token_mlp="frfn", enc_out=True).cuda()
    else:
        m = (EncoderBlock(dim, (S, S), 4, token_mlp="frfn", freq_mlp="frfn") if enc else FRFN(dim, hid)).cuda()
    win = dim * 2 if dec else dim                     # the decoder block works on cat([x, enc_out]) (model.py:146)
    x = torch.randn(B, S * S, win, device="cuda")
    dy = torch.randn(B, S * S, dim, device="cuda")
    plan = m._plan_for(x)
    xr, dyr = x.view(-1, win), dy.view(-1, dim)

    def step():
        plan.forward(xr)
        plan.backward(dyr)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    M = B * S * S
    c4 = dim // 4
    fwd_flops = M * (2 * dim * 2 * hid + 2 * hid * dim + 2 * 9 * c4 * c4 + 18 * hid)
    flops = 3 * fwd_flops
    if enc:
        # 2 FRFNs + LayerNorm (8/elem) + 4 applications of the (L, C) transform per step (2 forward, 2 backward), each counted
        # as an FFT would be: 5 N log2 N per length-N complex transform (N = L per channel, N = C per token)
        import math
        Ltok = S * S
        fft_flops = B * (dim * 5 * Ltok * math.log2(Ltok) + Ltok * 5 * dim * math.log2(dim))
        flops = 2 * 3 * fwd_flops + 3 * 8 * M * dim + 4 * fft_flops
    rows = plan.run_timed(plan.fwd) + plan.run_timed(plan.bwd)
    top = sorted(rows, key=lambda r: -r[4])[:6]
    what = "EncoderBlock (Fourier mode, FRFN mlps)" if enc else ("DecoderBlock (enc_out, MDASSA + FRFN)" if dec else "FRFN")
    if dec:
        flops = plan.flops()          # the plan's own count: GEMMs, window attention, depthwise convs, dense / two-stage DFT stages
    line = {"metric": f"{what} forward+backward images/s (dim {dim}, hidden {hid}, {S}x{S} tokens, bs {B})", "value": B / (ms * 1e-3),
            "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{what}(dim={dim}, hidden={hid}) fwd+bwd on {B} x {S}x{S} tokens"},
            "roofline": {"bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s",
                         "frac": flops / (ms * 1e-3) / MFMA_F32_PEAK, "traffic": None,
                         "hbm_achieved_GBps": plan.alg_bytes() / (ms * 1e-3) / 1e9, "hbm_frac": plan.alg_bytes() / (ms * 1e-3) / HBM_PEAK,
                         "plan_algorithmic_bytes": plan.alg_bytes(), "algorithmic_flops": flops,
                         "note": "composed of separate HBM-bound launches this round: the HBM fraction is the binding one",
                         "top_launches_serial_ms": [[r[0], r[1], round(r[4], 4)] for r in top]}}
    if enc:
        line["roofline"]["executed_flops"] = plan.flops()
        line["roofline"]["note"] = ("algorithmic flops count the (L,C) transforms as FFTs (5 N log2 N); the plan executes the length-L "
                                    "transform as two dense 256-point MFMA contractions: executed_flops is what the launches do")
    if not args.no_cpu and dec:
        from oracle import decoder_ref as D
        cores = min(16, len(os.sched_getaffinity(0)))
        torch.set_num_threads(cores)
        P = {n: v.detach().cpu() for n, v in m.state_dict().items()}
        xb, pb = x[:1].cpu(), dy[:1].cpu()
        fn = lambda t, Pp: D.decoder_block(t[:, :, :dim], t[:, :, dim:], Pp, 4)          # noqa: E731
        D.grads(fn, [xb], P, pb)
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < 10.0:
            D.grads(fn, [xb], P, pb)
            k += 1
        dt = (time.perf_counter() - t0) / k
        line["cpu_baseline"] = {"value": 1.0 / dt, "unit": "images/s", "cores": cores, "kind": "port",
                                "sample": f"{k} forward+backward passes of one {S}x{S} image through oracle/decoder_ref.py (torch CPU, {cores} threads)"}
    elif not args.no_cpu and enc:
        from oracle import encoder_ref as E
        cores = min(16, len(os.sched_getaffinity(0)))
        torch.set_num_threads(cores)
        P = {n: p.detach().cpu() for n, p in m.named_parameters()}
        xb, pb = x[:1].cpu(), dy[:1].cpu()
        E.out_and_grads(xb, P, pb)
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < 10.0:
            E.out_and_grads(xb, P, pb)
            k += 1
        dt = (time.perf_counter() - t0) / k
        line["cpu_baseline"] = {"value": 1.0 / dt, "unit": "images/s", "cores": cores, "kind": "port",
                                "sample": f"{k} forward+backward passes of one {S}x{S} image through oracle/encoder_ref.py (torch CPU, {cores} threads)"}
    elif not args.no_cpu:
        from oracle import frfn_ref as R
        cores = min(16, len(os.sched_getaffinity(0)))
        torch.set_num_threads(cores)
        P = {n: p.detach().cpu() for n, p in m.named_parameters()}
        xb, pb = x[:1].cpu(), dy[:1].cpu()
        R.out_and_grads(xb, P, pb)
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < 10.0:
            R.out_and_grads(xb, P, pb)
            k += 1
        dt = (time.perf_counter() - t0) / k
        line["cpu_baseline"] = {"value": 1.0 / dt, "unit": "images/s", "cores": cores, "kind": "port",
                                "sample": f"{k} forward+backward passes of one {S}x{S} image through oracle/frfn_ref.py (torch CPU, {cores} threads)"}
    print(json.dumps(line))


if __name__ == "__main__":
    main()

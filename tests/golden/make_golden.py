"""Generate the golden vectors under tests/golden/ FROM THE REFERENCE ITSELF.

Run once in the build container (where /root/reference is mounted):

    python tests/golden/make_golden.py

It imports the reference's ``src/Models/SpectralTransformer.py`` and
``src/Losses/luminanceLoss.py`` *by file path* (``import src.Models`` would
pull timm/torchvision/pywt, which are not installed), loads the closed-form
weights of ``oracle.cpu_ref.closed_form_state`` into the reference modules and
records what the reference computes.  Only data (inputs are regenerated from
the integer recipe; outputs are stored) is written - no reference source.

Files written (all small):
  st_manifest.npz      state_dict key order + shapes of the reference model
  st_modules.npz       per-module outputs + input-grads at 2xCx16x16
  st_full64.npz        full model 2x3x64x64: output, stage taps (strided
                       samples), losses, per-parameter grad L2 norms, a few
                       complete gradient tensors, list of grad-is-None names
  st_train_cfg1.npz    BASELINE config 1: 4 train / 2 test 64x64 pairs, B=2,
                       2 epochs Adam(3e-4)+clip 1.0+MultiStepLR: loss + norm per
                       step, val loss / PSNR per epoch, final parameter norms
  fft_zero_sign.npz    sign of Im at the self-conjugate bins of torch.fft.fft2
"""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import cpu_ref as O  # noqa: E402

REF = "/root/reference"


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    torch.set_num_threads(8)
    torch.manual_seed(0)
    ST = _load(f"{REF}/src/Models/SpectralTransformer.py", "ref_spectral_transformer")
    LUM = _load(f"{REF}/src/Losses/luminanceLoss.py", "ref_luminance")

    net = ST.SpectralTransformer()
    sd = net.state_dict()
    shapes = O.param_shapes()
    assert list(sd.keys()) == list(shapes.keys()), "oracle parameter inventory != reference state_dict"
    for k, v in sd.items():
        assert tuple(v.shape) == shapes[k], (k, v.shape, shapes[k])
    np.savez(os.path.join(HERE, "st_manifest.npz"),
             names=np.array(list(sd.keys())), shapes=np.array([str(tuple(v.shape)) for v in sd.values()]),
             n_params=np.int64(sum(v.numel() for v in sd.values())))

    P = O.closed_form_state()
    net.load_state_dict(P)
    net.train()

    # ---------------- per-module fixtures (2 x C x 16 x 16) -----------------
    mods = {}

    def run_module(tag, module, cin, hw=16, salt=11):
        x = O.module_input(cin, hw, salt).requires_grad_(True)
        y = module(x)
        # fixed pseudo-random cotangent
        g = torch.from_numpy(O._hash_uniform(y.numel(), 777 + salt).reshape(y.shape)).float()
        (gx,) = torch.autograd.grad((y * g).sum(), x)
        mods[f"{tag}.y"] = y.detach().numpy()
        mods[f"{tag}.gx"] = gx.numpy()

    run_module("encoders.0.0", net.encoders[0][0], 16, salt=11)
    run_module("encoders.1.0", net.encoders[1][0], 32, salt=12)
    run_module("refinement.0", net.refinement[0], 32, salt=13)
    run_module("encoders.3.0", net.encoders[3][0], 128, hw=8, salt=14)
    run_module("encoders.0.0.attn", net.encoders[0][0].attn, 16, salt=15)
    run_module("encoders.1.0.ffn", net.encoders[1][0].ffn, 32, salt=16)
    run_module("down1", net.down1, 16, salt=17)
    run_module("ups_3.Fups", net.ups_3.Fups, 32, salt=18)
    run_module("ups_3.Sups", net.ups_3.Sups, 32, salt=19)
    run_module("ups_3", net.ups_3, 32, salt=20)
    run_module("ups_1", net.ups_1, 128, hw=8, salt=21)
    run_module("ups_2", net.ups_2, 64, hw=12, salt=33)       # non-power-of-two spectrum
    np.savez_compressed(os.path.join(HERE, "st_modules.npz"), **mods)

    # ---------------- full model 2x3x64x64 ----------------------------------
    raw = O.closed_form_images(2, 64, 101)
    ref = O.closed_form_images(2, 64, 202)
    full = {}
    out = net(raw)
    full["out"] = out.detach().numpy()
    lum = LUM.LuminanceLoss()
    l1 = torch.nn.L1Loss()
    b, c = ref.shape[:2]
    # LossFunction("L1") = L1Loss / (B*C)  (losses.py:55-57); Charbonnier (:190-192);
    # ColorLoss (:209-212); LuminanceLoss (luminanceLoss.py:16-19) - restated inline
    # because losses.py cannot be imported (torchvision/pytorch_msssim/... missing).
    d = out - ref
    full["loss_L1"] = (l1(out, ref) / (b * c)).detach().numpy()
    full["loss_charbonnier"] = torch.mean(torch.sqrt(d * d + 1e-3 * 1e-3)).detach().numpy()
    full["loss_color"] = torch.mean(torch.mean(d ** 2, dim=(2, 3))).detach().numpy()
    full["loss_luminance"] = lum(out, ref).detach().numpy()
    imd = torch.clamp(out, 0, 1) - torch.clamp(ref, 0, 1)
    full["psnr"] = (20 * torch.log10(1 / (imd ** 2).mean().sqrt())).detach().numpy()

    net.zero_grad()
    loss = l1(out, ref) / (b * c)
    loss.backward()
    none_names, gnames, gnorms = [], [], []
    for n, p in net.named_parameters():
        if p.grad is None:
            none_names.append(n)
        else:
            gnames.append(n)
            gnorms.append(float(p.grad.double().norm()))
    full["grad_none"] = np.array(none_names)
    full["grad_names"] = np.array(gnames)
    full["grad_norms"] = np.array(gnorms)
    keep = ["embed_conv_rgb.weight", "encoders.0.0.attn.temperature", "encoders.0.0.attn.qkv.weight",
            "encoders.0.0.attn.kv.weight", "encoders.0.0.attn.kv_conv.weight", "encoders.0.0.norm1.weight",
            "encoders.3.3.ffn.project_out.weight", "ups_1.Fups.pha_fuse.0.weight", "ups_1.Fups.amp_fuse.2.bias",
            "ups_3.Sups.body.0.weight", "refinement.3.ffn.conv.weight", "refinement.3.attn.project_outf.weight",
            "down2.body.0.weight", "reduces1.weight", "outputl.weight", "output.weight"]
    named = dict(net.named_parameters())
    for k in keep:
        full[f"grad/{k}"] = named[k].grad.numpy()
    np.savez_compressed(os.path.join(HERE, "st_full64.npz"), **full)
    print("full64: loss_L1", float(full["loss_L1"]), "psnr", float(full["psnr"]),
          "n_none", len(none_names), "n_live", len(gnames))

    # ---------------- cfg1 training trajectory -------------------------------
    net.load_state_dict(O.closed_form_state())
    train = [(O.closed_form_images(1, 64, 1000 + i)[0], O.closed_form_images(1, 64, 2000 + i)[0]) for i in range(4)]
    test = [(O.closed_form_images(1, 64, 3000 + i)[0], O.closed_form_images(1, 64, 4000 + i)[0]) for i in range(2)]
    opt = torch.optim.Adam(net.parameters(), lr=3e-4)                                  # ModelTrainer.py:199
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1, 100, 250], gamma=0.25)  # :55
    rec = {"step_loss": [], "step_norm": [], "val_loss": [], "val_psnr": [], "lr": []}
    for epoch in range(2):
        net.train()
        for bi in range(2):   # shuffle off: batches are (0,1), (2,3)
            r = torch.stack([train[2 * bi][0], train[2 * bi + 1][0]])
            t = torch.stack([train[2 * bi][1], train[2 * bi + 1][1]])
            opt.zero_grad()                                                            # :78
            o = net(r)                                                                 # :79
            ls = l1(o, t) / (t.shape[0] * t.shape[1])                                  # :83
            ls.backward()                                                              # :86
            nrm = torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)                # :87
            opt.step()                                                                 # :88
            rec["step_loss"].append(ls.item())
            rec["step_norm"].append(float(nrm))
            rec["lr"].append(opt.param_groups[0]["lr"])
        sched.step()                                                                   # :129
        net.eval()
        with torch.no_grad():                                                          # :134-158
            r = torch.stack([test[0][0], test[1][0]])
            t = torch.stack([test[0][1], test[1][1]])
            o = net(r)
            rec["val_loss"].append((l1(o, t) / (t.shape[0] * t.shape[1])).item())
            imd = torch.clamp(o, 0, 1) - torch.clamp(t, 0, 1)
            rec["val_psnr"].append((20 * torch.log10(1 / (imd ** 2).mean().sqrt())).item())
    fin = {n: float(p.detach().double().norm()) for n, p in net.named_parameters()}
    np.savez_compressed(os.path.join(HERE, "st_train_cfg1.npz"),
                        **{k: np.array(v) for k, v in rec.items()},
                        final_names=np.array(list(fin.keys())), final_norms=np.array(list(fin.values())),
                        final_out=o.numpy())
    print("cfg1:", rec)

    # ---------------- fft zero-sign probe -----------------------------------
    signs = {}
    for n in (8, 16, 32, 64, 128):
        x = O.closed_form_images(2, n, 900 + n)
        f = torch.fft.fft2(x)
        bins = [(0, 0), (n // 2, 0), (0, n // 2), (n // 2, n // 2)]
        im = np.stack([f[..., a, b_].imag.numpy() for a, b_ in bins])
        signs[f"n{n}_imag_is_poszero"] = np.array(bool(np.all(im == 0) and not np.any(np.signbit(im))))
    np.savez(os.path.join(HERE, "fft_zero_sign.npz"), **signs)
    print("fft zero signs:", {k: bool(v) for k, v in signs.items()})


def init_checksums():
    """st_init_seed1234.npz: per-tensor sums of the REFERENCE module's default initialisation under
    torch.manual_seed(1234) - pins that the HIP module consumes the RNG stream identically."""
    ST = _load(f"{REF}/src/Models/SpectralTransformer.py", "ref_spectral_transformer_init")
    torch.manual_seed(1234)
    sd = ST.SpectralTransformer().state_dict()
    np.savez(os.path.join(HERE, "st_init_seed1234.npz"), names=np.array(list(sd.keys())),
             sums=np.array([float(v.double().sum()) for v in sd.values()]))


if __name__ == "__main__":
    main()
    init_checksums()

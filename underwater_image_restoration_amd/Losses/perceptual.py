"""VGGPerceptualLoss (src/Losses/losses.py:215-255) and the loss mixes that contain it (:88-117), on the HIP kernels.

    inp, target -> (x - mean) / std -> F.interpolate(bilinear, 224 x 224, align_corners=False)
    -> torchvision VGG16 features[:4] | [4:9] | [9:16] | [16:23] (frozen), loss = sum of the four F.l1_loss(x_i, y_i)

The reference builds the network with ``torchvision.models.vgg16(pretrained=True)``: a weight DOWNLOAD.  This build cannot download,
does not ship and does not fake those weights.  It takes them from the file the reference's call would have cached
(``$TORCH_HOME/hub/checkpoints/vgg16-397923af.pth``, default ``~/.cache/torch/hub/checkpoints/``), from ``UIR_VGG16_WEIGHTS=<path>``
(a torchvision ``vgg16().state_dict()`` file) or from ``set_vgg16_weights(state_dict)``; without any of them every loss that
contains the term raises with that explanation.  The arithmetic - preprocessing, 10 convolutions + ReLU, 3 max-poolings, the L1
feature losses and the whole backward to the input image - is checked against a CPU restatement with random weights
(oracle/perceptual_ref.py); with the pretrained file it is the reference's loss, with no pretrained file here to hold it to.

Convolutions are uir_gemm launches (taps = 9, bias, ReLU) on HWC rows, their input gradients uir_leaky_bwd + uir_gemm with flipped
taps; the weights are frozen, so no weight gradient is computed."""
from __future__ import annotations

import ctypes as C
import os

import torch

from .. import _lib as L
from .. import knobs

lib = L.lib
S = 224
_BLOCKS = (((0, 3, 64), (2, 64, 64)), ((5, 64, 128), (7, 128, 128)), ((10, 128, 256), (12, 256, 256), (14, 256, 256)),
           ((17, 256, 512), (19, 512, 512), (21, 512, 512)))            # (torchvision features index, Cin, Cout) per block
_weights_sd: "dict | None" = None


def set_vgg16_weights(state_dict) -> None:
    """Use this torchvision ``vgg16().state_dict()`` (keys ``features.<i>.weight`` / ``.bias``) for the perceptual loss."""
    global _weights_sd
    need = [f"features.{i}.{k}" for blk in _BLOCKS for (i, _, _) in blk for k in ("weight", "bias")]
    missing = [k for k in need if k not in state_dict]
    if missing:
        raise KeyError(f"not a torchvision vgg16 state dict: missing {missing[:3]} ...")
    _weights_sd = {k: state_dict[k].detach().to(torch.float32).cpu().contiguous() for k in need}
    _VggPlan._cache.clear()


def _find_weights():
    if _weights_sd is not None:
        return _weights_sd
    cand = []
    p = knobs.path("UIR_VGG16_WEIGHTS")
    if p:
        cand.append(p)
    home = knobs.path("TORCH_HOME") or os.path.join(os.path.expanduser("~"), ".cache", "torch")
    cand.append(os.path.join(home, "hub", "checkpoints", "vgg16-397923af.pth"))
    for f in cand:
        if os.path.isfile(f):
            set_vgg16_weights(torch.load(f, map_location="cpu", weights_only=True))     # a state_dict of tensors: nothing else is unpickled
            return _weights_sd
    raise NotImplementedError(
        "the perceptual loss needs torchvision's pretrained VGG16 weights, which the reference downloads and this build can neither "
        f"download nor ship: put the torchvision file at {cand[-1]} (where vgg16(pretrained=True) caches it), point UIR_VGG16_WEIGHTS "
        "at a vgg16().state_dict() file, or call Losses.perceptual.set_vgg16_weights(state_dict)")


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _VggPlan:
    """Launch lists of the perceptual loss for one (B, H, W): rec.fwd = both feature stacks + the four L1 means -> self.val,
    rec.bwd = the unit gradient with respect to the preprocessed prediction -> self.dx0 (scaled where it joins dpred)."""

    _cache: dict = {}

    @classmethod
    def get(cls, B, H, W, device):
        key = (B, H, W, str(device))
        if key not in cls._cache:
            if len(cls._cache) >= 2:
                cls._cache.pop(next(iter(cls._cache)))
            cls._cache[key] = cls(B, H, W, device)
        return cls._cache[key]

    def __init__(self, B, H, W, device):
        from ..engine import Recorder, _ptr
        sd = _find_weights()
        self.B, self.H, self.W = B, H, W
        r = self.rec = Recorder(device, scratch_floats=1 << 16)
        dev = r.dev
        self.Wt = {i: (sd[f"features.{i}.weight"].to(dev), sd[f"features.{i}.bias"].to(dev)) for blk in _BLOCKS for (i, _, _) in blk}
        r.keep.append(self.Wt)
        M0 = B * S * S
        self.x0 = [r.alloc(M0, 4), r.alloc(M0, 4)]               # preprocessed prediction / target
        self.val = r.alloc(1)
        self._token = None

        def stack(x, tag, keep):
            """-> block outputs; keep: [(layer output y, layer input x, idx, Cin, Cout, side)] per block for the backward"""
            feats, side = [], S
            for bi, blk in enumerate(_BLOCKS):
                if bi > 0:
                    Cc = x.shape[1]
                    y = r.alloc(B * (side // 2) ** 2, Cc)
                    r._hold(x, y)
                    r._op(f"{tag}.pool{bi}", lib.uir_maxpool2_hwc, _ptr(x), _ptr(y), B, side, side, Cc, nbytes=5 * x.numel())
                    side //= 2
                    x = y
                layers = []
                for (i, cin, cout) in blk:
                    Wc, bc = self.Wt[i]
                    y = r.alloc(B * side * side, cout)
                    kw = dict(k_grp=(3, 4)) if cin == 3 else {}
                    # bias in the convolution, ReLU as its own in-place pass: the launch paths of the wide 3x3 convolutions (tap groups,
                    # LDS-tiled kernel) carry a bias but no activation epilogue
                    if r.conv_as_patches(B * side * side, cin, cout):          # conv3_2 ... conv4_3 on a small batch: one K = 9 Cin GEMM
                        r.gemm(f"{tag}.conv{i}", r.patches3x3(f"{tag}.conv{i}.patches", x, cin, (B, side, side)), Wc, y, N=cout, K=9 * cin,
                               w_sn=9 * cin, w_sk=1, bias=bc)
                    else:
                        r.gemm(f"{tag}.conv{i}", x, Wc, y, N=cout, K=4 if cin == 3 else cin, w_sn=cin * 9, w_sk=9, w_st=1, taps=9, hw=(side, side),
                               bias=bc, **kw)
                    r._op(f"{tag}.relu{i}", lib.uir_relu_inplace, _ptr(y), y.numel(), nbytes=8 * y.numel())
                    layers.append((y, x, i, cin, cout, side))
                    x = y
                keep.append(layers)
                feats.append(x)
            return feats

        self.layers: list = []
        fp = stack(self.x0[0], "p", self.layers)
        ft = stack(self.x0[1], "t", [])
        nblk = 256
        slab = r.alloc(nblk)
        for bi, (a, b) in enumerate(zip(fp, ft)):
            n = a.numel()
            r._op(f"l1.{bi}", lib.uir_l1_feat_partials, _ptr(a), _ptr(b), _ptr(slab), nblk, n, nbytes=8 * n)
            r._op(f"l1.{bi}.red", lib.uir_reduce_slab, _ptr(slab), nblk, 1, _ptr(self.val), 1, 1 if bi else 0, 1.0 / n)
        # ---- backward (unit gradient of the loss value)
        r.cur = r.bwd
        d = None
        for bi in range(3, -1, -1):
            a, b = fp[bi], ft[bi]
            if d is None:
                d = r.alloc(*a.shape)
                r._op(f"l1.{bi}.bwd", lib.uir_l1_feat_bwd, _ptr(a), _ptr(b), _ptr(d), 1.0 / a.numel(), a.numel(), 0, nbytes=12 * a.numel())
            else:                                             # d = gradient of this block's pooled copy: through the pooling, + its own L1 term
                side = self.layers[bi][-1][5]
                dd = r.alloc(*a.shape)
                r._hold(a, d, dd)
                r._op(f"pool{bi + 1}.bwd", lib.uir_maxpool2_hwc_bwd, _ptr(a), _ptr(d), _ptr(dd), B, side, side, a.shape[1], 0, nbytes=12 * a.numel())
                r._op(f"l1.{bi}.bwd", lib.uir_l1_feat_bwd, _ptr(a), _ptr(b), _ptr(dd), 1.0 / a.numel(), a.numel(), 1, nbytes=12 * a.numel())
                d = dd
            for (y, x, i, cin, cout, side) in reversed(self.layers[bi]):
                Wc, _ = self.Wt[i]
                dz = r.alloc(*y.shape)
                r._hold(y, d, dz)
                r._op(f"conv{i}.relu_bwd", lib.uir_leaky_bwd, _ptr(y), y.stride(0), _ptr(d), d.stride(0), _ptr(dz), dz.stride(0), y.shape[0], cout, 0.0,
                      nbytes=12 * y.numel())
                dx = r.alloc(y.shape[0], 4 if cin == 3 else cin)
                kw = dict(n_grp=(3, 4)) if cin == 3 else {}
                if r.conv_as_patches(y.shape[0], cin, cout):
                    r.conv3x3_dx_from_patches(f"conv{i}.dX", dz, Wc, dx, cin, cout, (B, side, side))
                else:
                    r.gemm(f"conv{i}.dX", dz, Wc, dx, N=4 if cin == 3 else cin, K=cout, w_sn=9, w_sk=cin * 9, w_st=1, taps=9, flip=1, hw=(side, side), **kw)
                d = dx
        self.dx0 = d
        r.cur = r.fwd

    def forward(self, pred, truth, token=None) -> torch.Tensor:
        st = _stream()
        B, H, W = self.B, self.H, self.W
        L.check(lib.uir_vgg_prep(pred.data_ptr(), self.x0[0].data_ptr(), B, H, W, S, st), "uir_vgg_prep")
        L.check(lib.uir_vgg_prep(truth.data_ptr(), self.x0[1].data_ptr(), B, H, W, S, st), "uir_vgg_prep")
        self.rec.run(self.rec.fwd)
        self._token = token
        return self.val

    def backward(self, pred, truth, gscale_ptr: int, coef: float, dpred, token=None):
        """dpred += coef * gscale * d perceptual / d pred"""
        if token is None or token != self._token:           # the plan's features belong to another (pred, truth): recompute them
            self.forward(pred, truth)
        self.rec.run(self.rec.bwd)
        L.check(lib.uir_vgg_prep_bwd(self.dx0.data_ptr(), dpred.data_ptr(), gscale_ptr, coef, self.B, self.H, self.W, S, _stream()), "uir_vgg_prep_bwd")


# term coefficients of the reference's mixes (src/Losses/losses.py:82-117): charbonnier, perceptual, gradient, ffl, (1 - ms_ssim), tv
MIXES = {
    "perceptual": dict(perc=1.0),
    "mix": dict(ch=0.03, perc=0.025, grad=0.02, ssim=0.01),
    "fflMixTV": dict(ch=0.03, perc=0.025, grad=0.02, ssim=0.01, ffl=0.02, tv=1.0),
    "bigMix": dict(ch=0.4, perc=0.25, grad=0.25, ssim=0.1),
    "fflMix": dict(ch=0.03, perc=0.025, grad=0.01, ffl=0.005, ssim=0.1),
}
_KIND_NAME = {L.LOSS_KINDS[name]: name for name in MIXES}


def mix_sums(pred, truth, kind: int) -> torch.Tensor:
    """-> device tensor [10] = (loss, PSNR, 0, 0, charbonnier, perceptual, gradient, ffl, 1 - ms_ssim, tv): the first four like
    ops.loss_sums; pass it to mix_grad as `aux`."""
    from .. import ops
    c = MIXES[_KIND_NAME[kind]]
    B, _, H, W = pred.shape
    out = torch.zeros(10, dtype=torch.float32, device=pred.device)
    base = ops.loss_sums(pred, truth, L.LOSS_KINDS["charbonnier"])
    out[1] = base[1]
    out[4] = base[0]
    if "perc" in c:
        out[5] = _VggPlan.get(B, H, W, pred.device).forward(pred, truth, token=out.data_ptr())[0]
    if "grad" in c:
        out[6] = ops.loss_sums(pred, truth, L.LOSS_GRADIENT)[0]
    if "ffl" in c:
        ops._FflPlan.get(B, H, W, pred.device).forward(pred, truth, out.data_ptr() + 28)
    if "ssim" in c:
        ops._MsSsimPlan.get(B, H, W, pred.device).forward(pred, truth, out.data_ptr() + 32)
        out[8] = 1.0 - out[8]
    if "tv" in c:                                   # TVLoss(pred), weight 1 (:11-28)
        slab = torch.empty(512, dtype=torch.float32, device=pred.device)
        L.check(lib.uir_tv_partials(pred.data_ptr(), slab.data_ptr(), 256, B * 3, H, W, _stream()), "uir_tv_partials")
        s2 = slab.view(256, 2).sum(0)
        out[9] = 2.0 * (s2[0] / (3 * (H - 1) * W) + s2[1] / (3 * H * (W - 1))) / B
    out[0] = (c.get("ch", 0.0) * out[4] + c.get("perc", 0.0) * out[5] + c.get("grad", 0.0) * out[6] + c.get("ffl", 0.0) * out[7]
              + c.get("ssim", 0.0) * out[8] + c.get("tv", 0.0) * out[9])
    return out


def mix_grad(pred, truth, gscale, kind: int, aux=None) -> torch.Tensor:
    """gscale * d loss / d pred of a mix (every term's own backward, accumulated)."""
    from .. import ops
    c = MIXES[_KIND_NAME[kind]]
    B, _, H, W = pred.shape
    gs = gscale.reshape(1).to(torch.float32).contiguous()
    if "ch" in c:
        dpred = ops.loss_grad(pred, truth, gs * c["ch"], L.LOSS_KINDS["charbonnier"])
    else:
        dpred = torch.zeros_like(pred)
    if "grad" in c:
        dpred += ops.loss_grad(pred, truth, gs * c["grad"], L.LOSS_GRADIENT)
    if "ffl" in c:
        ops._FflPlan.get(B, H, W, pred.device).backward(pred, truth, gs.data_ptr(), None, dpred, accumulate=True, scale=c["ffl"])
    if "ssim" in c:
        ops._MsSsimPlan.get(B, H, W, pred.device).backward(pred, truth, gs.data_ptr(), -c["ssim"], dpred)
    if "perc" in c:
        _VggPlan.get(B, H, W, pred.device).backward(pred, truth, gs.data_ptr(), c["perc"], dpred, token=None if aux is None else aux.data_ptr())
    if "tv" in c:
        kh, kw = c["tv"] * 2.0 / (3 * (H - 1) * W * B), c["tv"] * 2.0 / (3 * H * (W - 1) * B)
        L.check(lib.uir_tv_bwd(pred.data_ptr(), dpred.data_ptr(), gs.data_ptr(), kh, kw, B * 3, H, W, _stream()), "uir_tv_bwd")
    return dpred

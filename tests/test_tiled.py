"""The LDS-tiled wide contraction (csrc/tiled.hip) behind uir_gemm / uir_gram: the Linear layers with 256 ... 2048 channels of
NewBigFRFNModel / AST (src/model/block.py:223-282, src/Models/AST.py:42-64,294-326) and their weight gradients, against fp64
matmuls; every operand layout (k-contiguous / index-contiguous), ragged M / N edges, split-k chunks with a ragged tail."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")]

from underwater_image_restoration_amd import _lib as L  # noqa: E402
from underwater_image_restoration_amd.engine import Recorder  # noqa: E402

DEV = "cuda"


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1


def dev(t):
    return t.float().to(DEV).contiguous()


@pytest.mark.parametrize("M,K,N", [(4096, 2048, 512), (1000, 1024, 256), (260, 256, 72), (16384, 512, 2048), (5000, 256, 64), (777, 512, 64)])
def test_wide_linear_forward_bias_gelu_preout(M, K, N):
    """linear1 of a LeFF: Y = GELU(X W^T + b), pre-activation stored too (weights [N][K], k-contiguous operands)."""
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2) * K ** -0.5, rnd(N, seed=3)
    r = Recorder(DEV, scratch_floats=1 << 20)
    Y, pre = r.alloc(M, N), r.alloc(M, N)
    r.gemm("l1", dev(x), dev(w), Y, N=N, K=K, w_sn=K, w_sk=1, bias=dev(b), act=L.ACT_GELU, pre_out=pre)
    r.run(r.fwd)
    want = x @ w.T + b
    assert rel(pre, want) < 2e-5 and rel(Y, F.gelu(want)) < 2e-5


@pytest.mark.parametrize("M,K,N", [(4096, 2048, 512), (520, 1024, 256), (2048, 512, 100), (3000, 256, 64)])
def test_wide_linear_input_gradient_and_residual(M, K, N):
    """dX = dY W (weights [K][N] read in place: the operand is contiguous along the OUTPUT index), + residual accumulate."""
    dy, w, res = rnd(M, K, seed=4), rnd(K, N, seed=5) * K ** -0.5, rnd(M, N, seed=6)
    r = Recorder(DEV, scratch_floats=1 << 20)
    dX, R = r.alloc(M, N), dev(res)
    if N % 4 == 0:
        r.gemm("dx", dev(dy), dev(w), dX, N=N, K=K, w_sn=1, w_sk=N, R=R)
        r.run(r.fwd)
        assert rel(dX, dy @ w + res) < 2e-5
    else:                                       # N not a multiple of 4: stays on the panel / fallback kernels, same answer
        r.gemm("dx", dev(dy), dev(w), dX, N=N, K=K, w_sn=1, w_sk=N)
        r.run(r.fwd)
        assert rel(dX, dy @ w) < 2e-5


@pytest.mark.parametrize("M,N,K", [(4096, 512, 2048), (16384, 2048, 512), (65536, 128, 128), (3000, 256, 136), (1048576, 128, 256),
                                   (262144, 256, 64), (50000, 64, 256), (4100, 192, 64), (2048, 64, 128)])   # the last four: 128 x 64 tiles
def test_wide_weight_gradient(M, N, K):
    """dW[n][k] = sum_m dY[m][n] X[m][k] over the token rows, partial slabs per chunk + uir_reduce_slab; also the transposed
    output layout of the ConvTranspose2d weight."""
    dy, x = rnd(M, N, seed=7), rnd(M, K, seed=8)
    r = Recorder(DEV, scratch_floats=8 << 20)
    out, outT = r.alloc(N, K), r.alloc(K, N)
    A, Bm = dev(dy), dev(x)
    r.gram("dw", A, Bm, out, N=N, K=K, o_sn=K, o_sk=1, own_slab=True)
    r.gram("dwT", A, Bm, outT, N=N, K=K, o_sn=1, o_sk=N, own_slab=True)
    r.run(r.fwd)
    want = dy.T @ x
    assert rel(out, want) < 3e-5 and rel(outT, want.T) < 3e-5
    again = out.clone()
    r.run(r.fwd)
    assert torch.equal(out, again)              # fixed summation order: bitwise run-to-run


@pytest.mark.parametrize("mode", ["1", "0", "3"])
def test_split_bf16_mode_is_f32_accurate(mode):
    """UIR_TILED_SPLIT: 1 = every wide contraction on the bf16 matrix pipe with every f32 operand split into three bf16 pieces and six
    products per pair (csrc/tiled.hip); 0 = f32 MFMA everywhere; 3 = the marked calls; (the default, 2, picks per shape - the other
    tests of this file run under it).  Against fp64 the error must stay at the f32 kernel's level in EVERY mode (<= 3e-6 of the
    largest output here; measured 0.6-1.9e-6 for both kernels).  The variable is read once per process: run in a child."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from underwater_image_restoration_amd.engine import Recorder\n"
        "g = torch.Generator().manual_seed(0)\n"
        "M, K, N = 4096, 1024, 512\n"
        "x = (torch.rand(M, K, generator=g, dtype=torch.float64) * 2 - 1) * torch.exp2(torch.randint(-6, 7, (M, 1), generator=g).double())\n"
        "w = (torch.rand(N, K, generator=g, dtype=torch.float64) * 2 - 1) * K ** -0.5\n"
        "dy = torch.rand(M, N, generator=g, dtype=torch.float64) * 2 - 1\n"
        "r = Recorder('cuda', scratch_floats=16 << 20)\n"
        "X, W, DY = x.float().cuda(), w.float().cuda(), dy.float().cuda()\n"
        "Y, DX, DW = r.alloc(M, N), r.alloc(M, K), r.alloc(N, K)\n"
        "r.gemm('f', X, W, Y, N=N, K=K, w_sn=K, w_sk=1)\n"
        "r.gemm('dx', DY, W, DX, N=K, K=N, w_sn=1, w_sk=K)\n"
        "r.gram('dw', DY, X, DW, N=N, K=K, o_sn=K, o_sk=1, own_slab=True)\n"
        "r.run(r.fwd); torch.cuda.synchronize()\n"
        "xd, wd, dyd = X.double().cpu(), W.double().cpu(), DY.double().cpu()\n"
        "for got, ref in ((Y, xd @ wd.T), (DX, dyd @ wd), (DW, dyd.T @ xd)):\n"
        "    e = float((got.double().cpu() - ref).abs().max() / ref.abs().max())\n"
        "    print('ERR %%.3e' %% e)\n"
        "    assert e < 3e-6, e\n" % root)
    env = dict(os.environ, UIR_TILED_SPLIT=mode)
    env.pop("UIR_MIXED", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ERR") == 3


@pytest.mark.parametrize("kind", ["bf16", "fp16"])
def test_mixed_mode_equals_the_rounded_operand_products(kind):
    """UIR_MIXED=bf16 | fp16 (opt-in mixed precision, BASELINE configs[4]; csrc/tiled.hip uir_tiled_lp_kernel): every wide contraction
    takes each f32 operand rounded ONCE, round-to-nearest-even, to the 16-bit type, multiplies exactly and accumulates in f32.  So the
    result must equal the fp64 product of the ROUNDED operands (torch's .to(bfloat16 / float16) is the same rounding) to f32 accumulation
    level - tolerance 5e-6 of the largest output, as for the f32 kernels - for the forward with bias + GELU + pre-activation copy, the
    input gradient with a residual and a per-sample output factor, and the split-k weight gradient with ragged edges; and it must
    differ from the unrounded product by the type's rounding level (the mode really is on).  Read once per process: run in a child."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "import torch.nn.functional as F\n"
        "from underwater_image_restoration_amd import _lib as L\n"
        "from underwater_image_restoration_amd.engine import Recorder\n"
        "lp = torch.bfloat16 if %r == 'bf16' else torch.float16\n"
        "q = lambda t: t.float().cpu().to(lp).double()\n"
        "g = torch.Generator().manual_seed(0)\n"
        "rnd = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64) * 2 - 1\n"
        "worst = 0.0\n"
        "def check(got, ref, exact, lo, hi):\n"
        "    e = float((got.double().cpu() - ref).abs().max() / ref.abs().max())\n"
        "    d = float((got.double().cpu() - exact).abs().max() / exact.abs().max())\n"
        "    print('ERR %%.3e vs rounded operands, %%.3e vs f32 operands' %% (e, d))\n"
        "    assert e < 5e-6, e\n"
        "    assert lo < d < hi, d\n"
        "for (M, K, N) in ((4096, 1024, 512), (1000, 512, 200), (300, 260, 136)):\n"
        "    x, w, b, dy, res = rnd(M, K), rnd(N, K) * K ** -0.5, rnd(N), rnd(M, N), rnd(M, K)\n"
        "    sc = rnd(4).abs() + 0.5\n"
        "    r = Recorder('cuda', scratch_floats=16 << 20)\n"
        "    X, W, Bv, DY, R, SC = (t.float().cuda().contiguous() for t in (x, w, b, dy, res, sc))\n"
        "    Y, PRE, DX, DW, DWT = r.alloc(M, N), r.alloc(M, N), r.alloc(M, K), r.alloc(N, K), r.alloc(K, N)\n"
        "    r.gemm('f', X, W, Y, N=N, K=K, w_sn=K, w_sk=1, bias=Bv, act=L.ACT_GELU, pre_out=PRE)\n"
        "    rows = M // 4\n"
        "    if N >= 256:\n"
        "        r.gemm('dx', DY, W, DX, N=K, K=N, w_sn=1, w_sk=K, R=R, out_scale=SC, scale_rows=rows)\n"
        "    r.gram('dw', DY, X, DW, N=N, K=K, o_sn=K, o_sk=1, own_slab=True)\n"
        "    r.gram('dwT', DY, X, DWT, N=N, K=K, o_sn=1, o_sk=N, own_slab=True)   # transposed output: the column-per-lane epilogue\n"
        "    r.run(r.fwd); torch.cuda.synchronize()\n"
        "    xd, wd, dyd = X.double().cpu(), W.double().cpu(), DY.double().cpu()\n"
        "    pre_q, pre_x = q(X) @ q(W).T + Bv.double().cpu(), xd @ wd.T + Bv.double().cpu()\n"
        "    check(PRE, pre_q, pre_x, 1e-5, 2e-2)\n"
        "    check(Y, F.gelu(pre_q), F.gelu(pre_x), 1e-5, 2e-2)\n"
        "    if N >= 256:\n"
        "        f = SC.double().cpu()[torch.arange(M) // rows][:, None]\n"
        "        check(DX, (q(DY) @ q(W)) * f + R.double().cpu(), (dyd @ wd) * f + R.double().cpu(), 1e-6, 2e-2)\n"
        "    check(DW, q(DY).T @ q(X), dyd.T @ xd, 1e-6, 2e-2)\n"
        "    check(DWT, (q(DY).T @ q(X)).T, (dyd.T @ xd).T, 1e-6, 2e-2)\n" % (root, kind))
    env = dict(os.environ, UIR_MIXED=kind)
    env.pop("UIR_TILED_SPLIT", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ERR") == 13, out.stdout


def test_mixed_mode_refuses_an_unknown_type():
    """UIR_MIXED=<anything but bf16 / fp16 / 0>: the first wide contraction fails with an error instead of silently running in f32."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, torch; sys.path.insert(0, %r)\n"
            "from underwater_image_restoration_amd.engine import Recorder\n"
            "r = Recorder('cuda', scratch_floats=1 << 20)\n"
            "X, W = torch.ones(512, 512, device='cuda'), torch.ones(256, 512, device='cuda')\n"
            "Y = r.alloc(512, 256)\n"
            "r.gemm('f', X, W, Y, N=256, K=512, w_sn=512, w_sk=1)\n"
            "try:\n"
            "    r.run(r.fwd); torch.cuda.synchronize()\n"
            "except Exception as e:\n"
            "    print('REFUSED', type(e).__name__)\n" % root)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, UIR_MIXED="fp8"), capture_output=True, text=True, timeout=300)
    assert "REFUSED" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(3, 10, 6, 512, 72), (1, 16, 16, 512, 200), (5, 60, 56, 256, 128)])
def test_wide_conv3x3_forward_and_input_gradient(B, H, W, Cin, Cout):
    """3 x 3 convolutions with >= 128 input channels (the sampler convolutions) as ONE launch of the tiled kernel: native weight
    layout [Cout][Cin][3][3] read at its own strides, neighbour rows zero outside the image, pixel counts that are no multiple of
    the 128-row tile, bias + residual; the input gradient through the same entry with mirrored taps."""
    x = rnd(B, Cin, H, W, seed=1).requires_grad_(True)
    w = rnd(Cout, Cin, 3, 3, seed=2) * (9 * Cin) ** -0.5
    bias, res = rnd(Cout, seed=4), rnd(B * H * W, Cout, seed=5)
    y = F.conv2d(x, w, bias, padding=1)
    dy = rnd(*y.shape, seed=3)
    (dx,) = torch.autograd.grad((y * dy).sum(), x)
    hwc = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()
    r = Recorder(DEV, scratch_floats=1 << 20)
    Wd = dev(w)
    Y, dX = r.alloc(B * H * W, Cout), r.alloc(B * H * W, Cin)
    r.gemm("c", dev(hwc(x.detach())), Wd, Y, N=Cout, K=Cin, w_sn=Cin * 9, w_sk=9, w_st=1, taps=9, hw=(H, W), bias=dev(bias), R=dev(res))
    r.gemm("c.dx", dev(hwc(dy)), Wd, dX, N=Cin, K=Cout, w_sn=9, w_sk=Cin * 9, w_st=1, taps=9, flip=1, hw=(H, W))
    r.run(r.fwd)
    assert rel(Y, hwc(y.detach()) + res) < 2e-5
    assert rel(dX, hwc(dx)) < 2e-5

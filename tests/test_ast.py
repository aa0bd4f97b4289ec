"""--arch AST (reference src/Models/AST.py) on the HIP path.
CPU: the factory builds the reference's 274 state_dict tensors with the same default-initialisation values under a seed
(tests/golden/ast_manifest.npz <- make_ast_vectors.py).  GPU: every building block and the whole network against outputs and
gradients recorded from the reference's own classes (tests/golden/ast_ref.npz); training-mode stochastic depth and a larger
token image against the fp64 oracle (oracle/ast_ref.py, itself pinned to the same fixtures by tests/test_ast_oracle.py)."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import ast_ref as O

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "ast_ref.npz"))
MAN = np.load(os.path.join(HERE, "golden", "ast_manifest.npz"))
S = 16
gpu = pytest.mark.gpu
needs_gpu = pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")


def _rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _params(tag, salt, dtype=torch.float32):
    shapes = {n: ast.literal_eval(s) for n, s in zip(G[tag + ".param_names"], G[tag + ".param_shapes"])}
    return O.closed_form_params(shapes, salt, dtype)


def test_factory_builds_the_references_ast():
    from underwater_image_restoration_amd.Models import init_model
    torch.manual_seed(1234)
    m = init_model("AST", use_dwt="Fourier")                    # the reference drops use_dwt for this arch (__init__.py:25-29)
    sd = m.state_dict()
    assert list(sd.keys()) == list(MAN["keys"]) and len(sd) == 274
    assert [str(tuple(v.shape)) for v in sd.values()] == list(MAN["shapes"])
    assert sum(p.numel() for p in m.parameters()) == int(MAN["n_params"]) == 19919507
    for (k, v), s, a in zip(sd.items(), MAN["sums"], MAN["abs_sums"]):      # same registration order and initialisers: same RNG stream
        assert abs(float(v.double().sum()) - s) <= 1e-7 * max(1.0, abs(s)), k
        assert abs(float(v.double().abs().sum()) - a) <= 1e-7 * max(1.0, a), k
    blocks = [b for l in O.LAYERS for b in m.get_submodule(l).blocks]
    assert np.allclose([b.drop_prob for b in blocks], MAN["drop_probs"], atol=1e-7)
    assert [b.shift_size for b in blocks] == [0, 4] * 9 and [b.att for b in blocks] == [False] * 8 + [True] * 10
    assert len(m.plan_modules()) == 28 and sum(len(list(t.parameters())) for t in m.plan_modules()) == len(list(m.parameters()))
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 100, 100))
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 3, 128, 128), mask=torch.zeros(1, 1, 128, 128))


def _module(tag):
    import importlib
    A = importlib.import_module("underwater_image_restoration_amd.Models.AST")
    blk = lambda dim, heads, shift, att: A.TransformerBlock(dim=dim, input_resolution=(S, S), num_heads=heads, win_size=8, shift_size=shift,
                                                            drop_path=0.05, token_mlp="leff", att=att, sparseAtt=att)
    return {"blk_att_s0": lambda: blk(64, 2, 0, True), "blk_att_s4": lambda: blk(64, 2, 4, True), "blk_noatt": lambda: blk(32, 1, 0, False),
            "down_32": lambda: A.Downsample(32, 64), "up_64": lambda: A.Upsample(64, 32),
            "inproj": lambda: A.InputProj(in_channel=3, out_channel=32), "outproj": lambda: A.OutputProj(in_channel=64, out_channel=3)}[tag]()


CASES = {"blk_att_s0": ("astblk0", (2, S * S, 64, 171), (2, S * S, 64, 172)), "blk_att_s4": ("astblk4", (2, S * S, 64, 171), (2, S * S, 64, 172)),
         "blk_noatt": ("astblkn", (2, S * S, 32, 173), (2, S * S, 32, 174)), "down_32": ("astdown", (2, S * S, 32, 173), (2, (S // 2) ** 2, 64, 175)),
         "up_64": ("astup", (2, S * S, 64, 171), (2, 4 * S * S, 32, 176)), "inproj": ("astin", (2, 3, S * S, 177), (2, S * S, 32, 174)),
         "outproj": ("astout", (2, S * S, 64, 171), (2, 3, S * S, 178))}


def test_block_surfaces_are_the_references():
    for tag in CASES:
        assert list(_module(tag).state_dict().keys()) == list(G[tag + ".keys"]), tag


@gpu
@needs_gpu
@pytest.mark.parametrize("tag", list(CASES))
def test_hip_block_matches_reference(tag):
    """Output, input gradient and every parameter gradient of the reference's own class (eval mode: DropPath inactive)."""
    salt, xin, pin = CASES[tag]
    m = _module(tag)
    m.load_state_dict(_params(tag, salt), strict=False)
    m = m.cuda().eval()
    x = O.closed_form_input(*xin)
    if tag == "inproj":
        x = x.reshape(2, 3, S, S)
    x = x.cuda().requires_grad_(tag != "inproj")
    want_y = G[tag + ".y"]
    y = m(x)
    assert tuple(y.shape) == want_y.shape
    (y * O.closed_form_input(*pin).cuda().reshape(y.shape)).sum().backward()
    torch.cuda.synchronize()
    e_y = _rel(y, want_y)
    e_dx = _rel(x.grad, G[tag + ".dx0"]) if tag != "inproj" else 0.0
    worst = max(((n, _rel(p.grad, G[f"{tag}.grad.{n}"])) for n, p in m.named_parameters()), key=lambda t: t[1])
    print(f"{tag}: y {e_y:.2e} dx {e_dx:.2e} worst param grad {worst}")
    assert e_y < 5e-5 and e_dx < 1e-4 and worst[1] < 5e-4


@gpu
@needs_gpu
def test_hip_shifted_block_larger_and_stochastic_depth_vs_oracle():
    """TransformerBlock(128, heads 4, shift 4) on 40 x 40 tokens (25 windows: every shift region), B = 3, TRAINING mode with
    drop_path 0.5: the factors the plan drew are handed to the fp64 oracle; bitwise run-to-run for fixed factors."""
    import importlib
    A = importlib.import_module("underwater_image_restoration_amd.Models.AST")
    dim, B, S2 = 128, 3, 40
    m = A.TransformerBlock(dim=dim, input_resolution=(S2, S2), num_heads=4, shift_size=4, drop_path=0.5, att=True, sparseAtt=True)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if v.is_floating_point()}
    P = O.closed_form_params(shapes, "astblk128")
    m.load_state_dict(P, strict=False)
    m = m.cuda().train()
    x, probe = O.closed_form_input(B, S2 * S2, dim, 181), O.closed_form_input(B, S2 * S2, dim, 182)
    torch.manual_seed(11)
    xg = x.cuda().requires_grad_(True)
    y = m(xg)
    (y * probe.cuda()).sum().backward()
    torch.cuda.synchronize()
    plan = next(iter(m._plans.values()))
    dp = plan.dp_scale.cpu().double()
    assert set(np.unique(dp.numpy())) <= {0.0, 2.0}
    fn = lambda t, Pp: O.transformer_block(t, Pp, "", 4, 4, True, dp=(dp[0], dp[1]))
    y64, dx64, g64 = O.grads(fn, [x.double()], {k: v.double() for k, v in P.items()}, probe.double())
    worst = max(((n, _rel(p.grad, g64[n])) for n, p in m.named_parameters()), key=lambda t: t[1])
    print("shifted 40x40 block, training:", _rel(y, y64), _rel(xg.grad, dx64[0]), worst, "factors", dp.tolist())
    assert _rel(y, y64) < 5e-5 and _rel(xg.grad, dx64[0]) < 1e-4 and worst[1] < 5e-4
    m.eval()
    outs = []
    for _ in range(2):
        xe = x.cuda().requires_grad_(True)
        ye = m(xe)
        (ye * probe.cuda()).sum().backward()
        outs.append((ye.detach().clone(), xe.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@gpu
@needs_gpu
def test_hip_network_matches_reference():
    """1 x 3 x 128 x 128 through the whole network, closed-form weights, eval mode: output, every gradient norm and thirteen full
    gradients recorded from the reference's AST."""
    from underwater_image_restoration_amd.Models import init_model
    tag, N = "net128", 128
    m = init_model("AST")
    m.load_state_dict(_params(tag, "astnet"), strict=False)
    m = m.cuda().eval()
    x = O.closed_form_input(1, 3, N * N, 179).reshape(1, 3, N, N).cuda()
    probe = O.closed_form_input(1, 3, N * N, 180).reshape(1, 3, N, N).cuda()
    y = m(x)
    (y * probe).sum().backward()
    torch.cuda.synchronize()
    e_y = _rel(y, G[tag + ".y"])
    norms = dict(zip([str(n) for n in G[tag + ".grad_names"]], G[tag + ".grad_norms"]))
    worst_n = ("", 0.0)
    for n, p in m.named_parameters():
        assert p.grad is not None, n
        e = abs(float(p.grad.double().norm()) - norms[n]) / max(norms[n], 1e-12)
        worst_n = max(worst_n, (n, e), key=lambda t: t[1])
    worst_g = ("", 0.0)
    pd = dict(m.named_parameters())
    for k in G.files:
        if k.startswith(tag + ".grad.") and k not in (tag + ".grad_names", tag + ".grad_norms"):
            worst_g = max(worst_g, (k, _rel(pd[k[len(tag) + 6:]].grad, G[k])), key=lambda t: t[1])
    print(f"AST 128x128: y {e_y:.2e}, worst gradient norm {worst_n}, worst full gradient {worst_g}")
    assert e_y < 5e-5 and worst_n[1] < 5e-4 and worst_g[1] < 1e-3


@gpu
@needs_gpu
def test_hip_network_train_steps_vs_torch_adam():
    """Two train steps (Charbonnier, clip 1.0, Adam 3e-4) of trainer.ModuleTrainStep on --arch AST in eval-mode arithmetic
    (drop_path_rate 0) vs torch autograd + clip_grad_norm_ + torch.optim.Adam on the fp64 oracle."""
    from underwater_image_restoration_amd.Models import init_model
    from underwater_image_restoration_amd.trainer import make_train_step
    N = 128
    m = init_model("AST", drop_path_rate=0.0)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if v.is_floating_point()}
    P = O.closed_form_params(shapes, "asttrain")
    m.load_state_dict(P, strict=False)
    m = m.cuda().train()
    raw = O.closed_form_input(1, 3, N * N, 191).reshape(1, 3, N, N)
    ref = O.closed_form_input(1, 3, N * N, 192).reshape(1, 3, N, N)
    step = make_train_step(m, "charbonnier", 3e-4)
    leaves = {k: v.double().clone().requires_grad_(True) for k, v in P.items()}
    opt = torch.optim.Adam(list(leaves.values()), lr=3e-4)
    for it in range(2):
        loss, norm = step(raw.cuda(), ref.cuda())
        opt.zero_grad()
        out = O.ast_net(raw.double(), leaves)
        l64 = torch.sqrt((out - ref.double()) ** 2 + 1e-6).mean()
        l64.backward()
        n64 = torch.nn.utils.clip_grad_norm_(list(leaves.values()), 1.0)
        opt.step()
        print(f"step {it}: loss {float(loss):.8f} / {float(l64):.8f}  norm {float(norm):.6f} / {float(n64):.6f}")
        assert abs(float(loss) - float(l64)) < 2e-5 * abs(float(l64)) and abs(float(norm) - float(n64)) < 5e-4 * float(n64)
    worst = max(((n, float((p.detach().cpu().double() - leaves[n].detach()).abs().max())) for n, p in m.named_parameters()), key=lambda t: t[1])
    print("worst parameter difference after two steps", worst)
    assert worst[1] < 2e-4


@gpu
@needs_gpu
def test_model_trainer_drives_ast(tmp_path, capsys):
    """BASELINE configs[4]'s driver path (fp32, Charbonnier) on a tiny PNG dataset: ModelTrainer(dirs).train(args, "AST", ...) in
    training mode (stochastic depth on), validation, a checkpoint with the reference's keys and 274 tensor names that reloads
    bit-identically."""
    import types
    from PIL import Image
    from underwater_image_restoration_amd.ModelTrainer import ModelTrainer
    from underwater_image_restoration_amd.Models import init_model
    rng = np.random.RandomState(0)
    dirs = {}
    for d in ("a", "b", "ta", "tb"):
        p = tmp_path / d
        p.mkdir()
        dirs[d] = str(p)
        for i in range(2 if d in ("a", "b") else 1):
            Image.fromarray(rng.randint(0, 255, (128, 128, 3), dtype=np.uint8)).save(p / f"{i}_img_.png")
    args = types.SimpleNamespace(train_batch_size=1, lossf="charbonnier", lr=3e-4, arch="AST", use_dwt="Fourier", optim="adam",
                                 image_size=128, loader_workers=0)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        torch.manual_seed(3)
        model = ModelTrainer(dirs["a"], dirs["b"], dirs["ta"], dirs["tb"]).train(args, "AST", 1, 3e-4, "cuda")
    finally:
        os.chdir(cwd)
    out = capsys.readouterr().out
    assert "Batch 2/2, Loss:" in out and "Validation PSNR:" in out and "Training completed!" in out
    ckpts = [os.path.join(r, f) for r, _, fs in os.walk(tmp_path) for f in fs if f.endswith(".pth")]
    ck = torch.load(sorted(ckpts)[-1], weights_only=True)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss"}
    assert list(ck["model_state_dict"].keys()) == list(MAN["keys"])
    fresh = init_model("AST")
    fresh.load_state_dict(ck["model_state_dict"])
    fresh = fresh.cuda().eval()
    model.eval()
    x = O.closed_form_input(1, 3, 128 * 128, 151).reshape(1, 3, 128, 128).cuda()
    with torch.no_grad():
        assert torch.equal(fresh(x), model(x))


def _dp_ast_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)                         # both ranks share the one GPU of this box: gloo moves the bucket
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from underwater_image_restoration_amd.Models import init_model
    from underwater_image_restoration_amd.trainer import make_train_step
    m = init_model("AST", drop_path_rate=0.0)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if v.is_floating_point()}
    P = O.closed_form_params(shapes, "astdp")
    if rank == 1:
        P = {k: v + 0.125 for k, v in P.items()}      # rank 1 starts elsewhere: the step must broadcast rank 0's weights
    m.load_state_dict(P, strict=False)
    m = m.cuda().train()
    step = make_train_step(m, "charbonnier", 3e-4)
    N = 128
    raw = O.closed_form_input(2, 3, N * N, 201).reshape(2, 3, N, N)[rank:rank + 1].cuda()
    ref = O.closed_form_input(2, 3, N * N, 202).reshape(2, 3, N, N)[rank:rank + 1].cuda()
    losses = []
    for _ in range(2):
        loss, norm = step(raw, ref)
        losses.append((float(loss), float(norm)))
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()
    torch.save({"flat": flat, "losses": losses, "bucket": int(step.bucket.numel())}, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@gpu
@needs_gpu
def test_ast_data_parallel_two_ranks(tmp_path):
    """BASELINE configs[4]'s parallelism with the product's own step: 2 processes x 1 image (HIP forward / backward, ONE all-reduce of
    the network's gradient bucket, fused clip + Adam with gscale = 1/2) == 1 process x 2 images for the mean-type Charbonnier loss;
    replicas end bit-identical; rank 1's different initial weights are overwritten by the broadcast.  (gloo carries the bucket: two
    ranks cannot share one GPU under RCCL.)"""
    import torch.multiprocessing as mp
    from underwater_image_restoration_amd.Models import init_model
    from underwater_image_restoration_amd.trainer import make_train_step
    out = str(tmp_path / "dpast")
    port = 32500 + (os.getpid() % 2000)
    mp.spawn(_dp_ast_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0["flat"], r1["flat"]) and r0["bucket"] >= 19919507
    m = init_model("AST", drop_path_rate=0.0)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if v.is_floating_point()}
    m.load_state_dict(O.closed_form_params(shapes, "astdp"), strict=False)
    m = m.cuda().train()
    step = make_train_step(m, "charbonnier", 3e-4)
    N = 128
    raw = O.closed_form_input(2, 3, N * N, 201).reshape(2, 3, N, N).cuda()
    ref = O.closed_form_input(2, 3, N * N, 202).reshape(2, 3, N, N).cuda()
    for k in range(2):
        loss, norm = step(raw, ref)
        assert abs(float(loss) - 0.5 * (r0["losses"][k][0] + r1["losses"][k][0])) <= 2e-5 * float(loss)
        assert abs(float(norm) - r0["losses"][k][1]) <= 2e-4 * float(norm)
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()
    d = float((flat - r0["flat"]).abs().max())
    assert d <= 2e-4, d


@pytest.mark.gpu
def test_ast_train_step_with_the_fflmix_loss_of_baseline_config_4():
    """BASELINE configs[4] = --arch AST with the "perceptual+ffl mix" loss (fflMix, src/Losses/losses.py:108-117), here at 256 x 256,
    bs 1, fp32 and RANDOM VGG16 weights in torchvision's layout (the pretrained file is the user's to supply): the fused step's loss
    and pre-clip gradient norm equal the autograd path's with LossFunction("fflMix"), and two steps stay finite."""
    from oracle import perceptual_ref as R
    from underwater_image_restoration_amd.Losses import perceptual
    from underwater_image_restoration_amd.Losses.losses import LossFunction
    from underwater_image_restoration_amd.Models import init_model
    from underwater_image_restoration_amd.trainer import make_train_step
    perceptual.set_vgg16_weights(R.random_vgg16_state(5))
    try:
        N = 256
        raw = (0.5 + 0.5 * O.closed_form_input(1, 3, N * N, 291).reshape(1, 3, N, N)).cuda()
        ref = (0.5 + 0.5 * O.closed_form_input(1, 3, N * N, 292).reshape(1, 3, N, N)).cuda()
        nets = []
        for _ in range(2):
            m = init_model("AST", drop_path_rate=0.0)
            shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if v.is_floating_point()}
            m.load_state_dict(O.closed_form_params(shapes, "astfflmix"), strict=False)
            nets.append(m.cuda().train())
        step = make_train_step(nets[0], "fflMix", 3e-4)
        l0, n0 = step(raw, ref)
        l0, n0 = float(l0), float(n0)
        tup = LossFunction("fflMix", "cuda").getloss(nets[1](raw), ref)
        tup[0].backward()
        norm = torch.nn.utils.clip_grad_norm_(nets[1].parameters(), 1.0)
        assert len(tup) == 6 and abs(float(tup[0]) - l0) < 1e-5 * abs(l0), (float(tup[0]), l0)
        assert abs(float(norm) - n0) < 2e-4 * n0, (float(norm), n0)
        l1, n1 = step(raw, ref)
        assert np.isfinite(float(l1)) and np.isfinite(float(n1))
    finally:
        perceptual._weights_sd = None
        perceptual._VggPlan._cache.clear()


_MIXED_CHILD = """
import sys, torch; sys.path.insert(0, %r)
from oracle import ast_ref as O
from underwater_image_restoration_amd.Models import init_model
from underwater_image_restoration_amd.trainer import make_train_step
m = init_model("AST", drop_path_rate=0.0)
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if v.is_floating_point()}
m.load_state_dict(O.closed_form_params(shapes, "astmixed"), strict=False)
m = m.cuda().train()
step = make_train_step(m, "charbonnier", 3e-4)
N = 128
raw = O.closed_form_input(1, 3, N * N, 301).reshape(1, 3, N, N).cuda()
ref = O.closed_form_input(1, 3, N * N, 302).reshape(1, 3, N, N).cuda()
l0, n0 = step(raw, ref)
grad = step.bucket.detach().clone().cpu()
l1, n1 = step(raw, ref)
torch.cuda.synchronize()
# the same mode through gradient accumulation (the fp16 loss scale rides on the micro-batch weight): 2 x 1 image == 1 x 2 images
pair = []
raw2 = O.closed_form_input(2, 3, N * N, 303).reshape(2, 3, N, N).cuda()
ref2 = O.closed_form_input(2, 3, N * N, 304).reshape(2, 3, N, N).cuda()
for mb in (None, 1):
    mm = init_model("AST", drop_path_rate=0.0)
    mm.load_state_dict(O.closed_form_params(shapes, "astmixed"), strict=False)
    st = make_train_step(mm.cuda().train(), "charbonnier", 3e-4)
    st.micro_batch = mb
    l, n = st(raw2, ref2)
    pair += [float(l), float(n)]
torch.cuda.synchronize()
torch.save({"loss": [float(l0), float(l1)], "norm": [float(n0), float(n1)], "grad": grad, "pair": pair}, sys.argv[1])
"""


@gpu
@needs_gpu
def test_ast_train_step_in_mixed_precision_stays_close_to_fp32(tmp_path):
    """BASELINE configs[4] names "mixed fp16" (the reference itself has no autocast / GradScaler: nothing to be equal to).  UIR_MIXED =
    bf16 / fp16 runs the wide Linear layers of AST - forward, input gradient, weight gradient - with 16-bit operands and f32
    accumulation (tests/test_tiled.py pins that arithmetic exactly); here the whole train step: against the f32 step from the same
    weights the loss moves by < 1e-2 (relative), the pre-clip gradient norm by < 5e-2, the gradient direction by cosine > 0.99 - and
    it DOES move (the mode is on) -, and a second step stays finite.  One child process per mode: the library reads it once."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    res = {}
    for kind in ("0", "bf16", "fp16"):
        out = str(tmp_path / f"mixed_{kind}.pt")
        env = dict(os.environ, UIR_MIXED=kind)
        env.pop("UIR_TILED_SPLIT", None)
        p = subprocess.run([sys.executable, "-c", _MIXED_CHILD % root, out], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout + p.stderr
        res[kind] = torch.load(out)
    f = res["0"]
    for kind in ("bf16", "fp16"):
        r = res[kind]
        assert all(np.isfinite(r["loss"])) and all(np.isfinite(r["norm"]))
        dl = abs(r["loss"][0] - f["loss"][0]) / abs(f["loss"][0])
        dn = abs(r["norm"][0] - f["norm"][0]) / f["norm"][0]
        cos = float(torch.dot(r["grad"].double(), f["grad"].double()) / (r["grad"].double().norm() * f["grad"].double().norm()))
        print(f"mixed {kind}: loss rel {dl:.2e}, norm rel {dn:.2e}, gradient cosine {cos:.6f}")
        assert dl < 1e-2 and dn < 5e-2 and cos > 0.99, (kind, dl, dn, cos)
        assert not torch.equal(r["grad"], f["grad"])
    for kind, r in res.items():          # full batch == two micro-batches, in every mode (loss 1e-5, pre-clip gradient norm 1e-3)
        lA, nA, lB, nB = r["pair"]
        assert abs(lA - lB) <= 1e-5 * abs(lA) and abs(nA - nB) <= 1e-3 * nA, (kind, r["pair"])


_CFG5_CHILD = """
import sys, torch; sys.path.insert(0, %r)
from oracle import ast_ref as O
from oracle import perceptual_ref as R
from underwater_image_restoration_amd.Losses import perceptual
from underwater_image_restoration_amd.Models import init_model
from underwater_image_restoration_amd.trainer import make_train_step
perceptual.set_vgg16_weights(R.random_vgg16_state(5))
N, B = 512, 4
raw = (0.5 + 0.5 * O.closed_form_input(B, 3, N * N, 311).reshape(B, 3, N, N)).cuda()
ref = (0.5 + 0.5 * O.closed_form_input(B, 3, N * N, 312).reshape(B, 3, N, N)).cuda()
out = {}
shapes = None
for name, mb in (("full", None), ("micro", 2)):
    m = init_model("AST", drop_path_rate=0.0)
    if shapes is None:
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if v.is_floating_point()}
    m.load_state_dict(O.closed_form_params(shapes, "astcfg5"), strict=False)
    st = make_train_step(m.cuda().train(), "fflMix", 3e-4)
    st.micro_batch = mb
    l, n = st(raw, ref)
    out[name] = [float(l), float(n)]
    if name == "full":
        l2, n2 = st(raw, ref)
        out["second"] = [float(l2), float(n2)]
    del st, m
    torch.cuda.synchronize(); torch.cuda.empty_cache()
torch.save(out, sys.argv[1])
"""


@gpu
@needs_gpu
def test_ast_cfg5_as_baseline_states_it(tmp_path):
    """BASELINE configs[4] in ONE run: --arch AST, 512 x 512, bs 4, the "perceptual+ffl mix" loss (fflMix, src/Losses/losses.py:108-117;
    VGG16 weights random in torchvision's layout - the pretrained file is the user's to supply) under UIR_MIXED=fp16 (the reference has no
    autocast of its own: src/Losses/losses.py:9 is an unused import).  The fused train step is finite for two steps, its loss stays
    within 1e-2 of the f32 step's from the same weights, and 2 x 2 micro-batches equal the full batch (the static fp16 loss scale
    rides on the micro-batch weight).  One child per mode: libuir_hip.so reads the variable once."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    res = {}
    for kind in ("0", "fp16"):
        out = str(tmp_path / f"cfg5_{kind}.pt")
        env = dict(os.environ, UIR_MIXED=kind)
        env.pop("UIR_TILED_SPLIT", None)
        p = subprocess.run([sys.executable, "-c", _CFG5_CHILD % root, out], env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
        res[kind] = torch.load(out)
    f32, f16 = res["0"], res["fp16"]
    for r in (f32, f16):
        assert all(np.isfinite(v) for k in ("full", "second", "micro") for v in r[k]), r
        lA, nA = r["full"]
        lB, nB = r["micro"]
        # (full batch and micro-batches run some contractions on different kernels - the routing depends on the row count - and 30
        # blocks of random, untrained weights amplify that rounding-level difference: measured 1.4e-4 on the loss and 1.2e-2 on the
        # pre-clip norm in f32, whose L1-type terms have sign kinks; the loss terms themselves agree to 1e-8 between a batch and the
        # mean of its halves)
        assert abs(lA - lB) <= 1e-3 * abs(lA) and abs(nA - nB) <= 5e-2 * nA, r
    dl = abs(f16["full"][0] - f32["full"][0]) / abs(f32["full"][0])
    print(f"cfg5 fp16 vs f32: loss {f16['full'][0]:.6f} vs {f32['full'][0]:.6f} (rel {dl:.2e}), norm {f16['full'][1]:.4f} vs {f32['full'][1]:.4f}")
    assert dl < 1e-2, (f16, f32)
    assert f16["full"] != f32["full"]                      # the mode really is on

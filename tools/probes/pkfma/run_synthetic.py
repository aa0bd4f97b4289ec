"""Which packed-f32 instruction forms lose results next to bf16 MFMA waves?  Register-only victims (synthetic.hip: one instruction
form each, +-1 operands, exact sums) run alone and beside the split-bf16 weight-gradient contraction of the product; results are
compared bit for bit.

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/pkfma/synthetic.hip -o tools/probes/pkfma/libpkfma_synth.so
    python tools/probes/pkfma/run_synthetic.py [runs=30] [aggressor=split|f32]"""
import ctypes as C, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
AGG = sys.argv[2] if len(sys.argv) > 2 else "split"
if AGG == "split":
    os.environ["UIR_TILED_SPLIT"] = "1"
import torch
from underwater_image_restoration_amd.engine import Recorder

syn = C.CDLL(os.path.join(HERE, "libpkfma_synth.so"))
syn.form_text.restype = C.c_char_p
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
BLOCKS, ITERS = 256, 12000
nthr = BLOCKS * 256
inp = (torch.randint(0, 2, (nthr, 4), device=dev, generator=g).float() * 2 - 1).contiguous()
out = torch.empty(nthr, 8, 2, device=dev)
r2 = Recorder(dev, scratch_floats=4 << 20)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
Mg, K, N = 16384, 512, 2048
xa, ya, dwa = rnd(Mg, K), rnd(Mg, N), r2.alloc(N, K)
r2.gram("g", ya, xa, dwa, N=N, K=K, o_sn=K, o_sk=1, own_slab=True)
s2 = torch.cuda.Stream()


def victim(form):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = syn.launch_victim(form, C.c_void_p(inp.data_ptr()), C.c_void_p(out.data_ptr()), ITERS, BLOCKS, st)
    assert rc == 0, rc


print(f"aggressor: {AGG}; victims: {BLOCKS} x 256 threads, {ITERS} x 8 instructions each, {RUNS} runs per form", flush=True)
for form in range(8):
    victim(form); torch.cuda.synchronize()
    ref = out.clone()
    alone = 0
    for _ in range(5):
        out.zero_(); victim(form); torch.cuda.synchronize()
        alone += int(not torch.equal(out, ref))
    bad, rows, halves, nwrong = 0, set(), set(), 0
    for _ in range(RUNS):
        out.zero_()
        victim(form)                          # the long victim first, the contraction fills in around it
        with torch.cuda.stream(s2):
            for _k in range(3):
                r2.run(r2.fwd)
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
            idx = (out != ref).nonzero()
            nwrong += idx.shape[0]
            for t, _i, h in idx[:4096].tolist():
                rows.add((t % 64) // 16); halves.add("lo" if h == 0 else "hi")
    txt = syn.form_text(form).decode().replace("\n\t", " ; ").replace("%0", "D").replace("%1", "A").replace("%2", "B")
    print(f"form {form}: alone {alone}/5, beside {bad}/{RUNS} runs differ"
          + (f" ({nwrong} values; 16-lane rows {sorted(rows)}, halves {sorted(halves)})" if bad else "") + f"   {txt}", flush=True)

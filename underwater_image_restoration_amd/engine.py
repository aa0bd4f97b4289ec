"""Static launch plan for the SpectralTransformer train step on one MI355X.

The network (reference: src/Models/SpectralTransformer.py:213-270) is compiled ONCE per input
shape into two flat lists of kernel launches - forward and hand-written backward - over
buffers that live for the lifetime of the plan (288 GB of HBM3E makes activation
re-materialisation pointless at B=16, 256x256).  Executing a step is walking the lists and
calling the C ABI of include/uir_kernels.h with pre-built argument structs on the current HIP
stream: no tracing compiler, no per-step allocation, no host synchronisation.

Layout: activations are HWC rows, `[B*H*W, C]` fp32, NCHW only at the model boundary.
Parameters live in ONE flat fp32 buffer (live tensors first) with a parallel flat gradient
buffer: the gradient buffer is the single RCCL all-reduce bucket and the operand of the fused
clip+Adam kernel; `nn.Parameter`s of the host module are views into it.

Algebra used (exact): MDTA's `attn @ v -> project_out -> kv` and
`attn @ vf -> project_outf` are folded into two per-batch 1x1 GEMMs with composed weights
(uir_attn_fwd); the GDFN gate and LayerNorm are GEMM prologues; ifft2 of the 2x2-tiled spectrum
is evaluated on the H x W grid and zero-interleaved.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Callable, Dict, List, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from . import knobs

lib = L.lib

NUM_BLOCKS = (2, 3, 3, 4)
NUM_HEADS = (1, 2, 4, 8)
CHANNELS = (16, 32, 64, 128)
NUM_REFINEMENT = 4
EXPANSION = 2.66
LN_EPS = 1e-5
# Which tile-fused kernels each level uses is fixed here (measured choices, profiles/r01/d_ab_notes.md), not read from the
# environment: the only environment variables the product reads are listed in knobs.py.
FUSED_DFT_SIZES = (32, 64, 128)   # axis lengths of the two-stage DFT kernel, which carries UpSample's element-wise ops (UirDft2Args.pro / .epi)
_ATTNV_FUSED = (16, 32, 64, 128)   # MDTA tail as uir_attnv_fwd on every level (round 1 kept C = 16 / 128 on three launches: as fast then; since the
                                   # fused forwards' round-4 work C = 16 is 100 vs 132 us, C = 128 - 4 x 8 tiles - 24 vs ~25 us and two launches fewer)
SCRATCH_FLOATS = 96 * 1024 * 1024  # 384 MB of slab scratch shared by every reduction


# Plan memory model: every activation / gradient buffer lives for the plan's lifetime and is linear in B*H*W
# (measured by building plans of several shapes: 40.4-43.1 KB per pixel with the backward list, 18.7-19.3 KB forward only, the smaller plans
# the larger figures - the estimate's 64 MB margin covers them; 52.7-53.6 / 28.8-29.2 while the C <= 32 blocks kept their pre-depthwise tensors;
# tests/test_cpu_surface.py holds the estimate to the real Plan.bytes), plus two slab scratch areas.
PLAN_BYTES_PER_PIXEL_TRAIN = 41_000
PLAN_BYTES_PER_PIXEL_EVAL = 19_000


def plan_bytes_estimate(B: int, H: int, W: int, want_backward: bool = True) -> int:
    """Upper estimate of the device memory one Plan(B, H, W) allocates (bytes)."""
    per = PLAN_BYTES_PER_PIXEL_TRAIN if want_backward else PLAN_BYTES_PER_PIXEL_EVAL
    return B * H * W * per + 2 * SCRATCH_FLOATS * 4 + (64 << 20)


# The tile kernels address a tensor with 32-bit byte offsets: rows * row stride must stay below 2^30 floats.  The widest
# full-resolution rows are the GDFN tensors of the C = 32 stages (2 * hp = 192 floats), so one plan holds at most this many
# full-resolution pixels (5 images of 1024 x 1024, 85 of 256 x 256); larger batches run as micro-batches (TrainStep).
PLAN_MAX_PIXELS = (1 << 30) // 192 - 1


def plan_fits_addressing(B: int, H: int, W: int) -> bool:
    return B * H * W <= PLAN_MAX_PIXELS


def device_bytes_available(device) -> int:
    """Bytes a new plan can still get on `device`: free HBM plus what torch's caching allocator holds unused."""
    free, _total = torch.cuda.mem_get_info(device)
    return int(free + torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device))


def hidden_of(c: int) -> int:
    return int(c * EXPANSION)


def pad4(n: int) -> int:
    return (n + 3) & ~3


def pad16(n: int) -> int:
    return (n + 15) & ~15


def is_dead_param(name: str) -> bool:
    """Parameters the reference never uses on the live path (grad stays None):
    SpectralTransformer.py:103-106,112 (q1X1_*) and :231,233,248,250."""
    if ".attn.q1X1_" in name:
        return True
    return name.startswith(("ups_4.", "ups1.", "ups2.", "output1."))


class FlatParams:
    """All parameters in one flat fp32 buffer (live first, 16-byte aligned slots) plus a
    parallel gradient buffer.  `view(name)` / `gview(name)` are tensor views."""

    def __init__(self, shapes: "Dict[str, Tuple[int, ...]]", device, dead: "frozenset | None" = None):
        """dead: names of the parameters that never receive a gradient (default: the SpectralTransformer's, is_dead_param).
        They sit AFTER the live ones: the optimizer works on [0, n_live) only, as torch's skips `p.grad is None`."""
        isdead = is_dead_param if dead is None else (lambda n: n in dead)
        names = [n for n in shapes if not isdead(n)] + [n for n in shapes if isdead(n)]
        self.shapes = dict(shapes)
        self.slots: Dict[str, Tuple[int, int]] = {}
        off = 0
        self.n_live = 0
        for n in names:
            numel = int(np.prod(shapes[n]))
            self.slots[n] = (off, numel)
            off += pad4(numel)
            if not isdead(n):
                self.n_live = off
        self.total = off
        self.device = torch.device(device)
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.live_names = [n for n in names if not isdead(n)]
        self.dead_names = frozenset(n for n in names if isdead(n))

    def view(self, name: str) -> torch.Tensor:
        o, n = self.slots[name]
        return self.flat[o:o + n].view(self.shapes[name])

    def gview(self, name: str) -> torch.Tensor:
        o, n = self.slots[name]
        return self.grad[o:o + n].view(self.shapes[name])


_TWIDDLE_CACHE: Dict[Tuple[int, str], Tuple[torch.Tensor, torch.Tensor]] = {}


def twiddles(n: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin(2*pi*k*j/n) tables [n][n], evaluated in float64 on exact residues k*j mod n."""
    key = (n, str(device))
    if key not in _TWIDDLE_CACHE:
        kj = (np.arange(n, dtype=np.int64)[:, None] * np.arange(n, dtype=np.int64)[None, :]) % n
        ang = 2.0 * np.pi * kj.astype(np.float64) / n
        c, s = np.cos(ang), np.sin(ang)
        # exact zeros / ones where the angle is a multiple of pi/2
        q = (4 * kj) % n == 0
        c[q] = np.round(c[q]); s[q] = np.round(s[q])
        _TWIDDLE_CACHE[key] = (torch.from_numpy(c.astype(np.float32)).to(device),
                               torch.from_numpy(s.astype(np.float32)).to(device))
    return _TWIDDLE_CACHE[key]


_GRAM_WGS = 512                # workgroups (= partial rows x column groups) per Gram launch (1 024 until the end of round 5: the bulk lane's Grams run BESIDE the chain's
                               # kernels, and with half the workgroups they take less from them - 31.75 -> 31.1 ms together with BULK_DEFER 30 -> 45, profiles/r05/ab_notes.md u)
# Cross-lane ordering costs the CHAIN: every event recorded on (or waited for by) the main stream is a barrier packet between two of
# its kernels - 6-7 us of idle queue each (rocprofv3 kernel trace, tools/trace_gaps.py: 162 such gaps = 1.05 ms of a 36.4 ms step).
# BULK_DEFER: weight-gradient (bulk lane) launches are held back on the host until this many chain launches have passed since the
# first held one, then released behind ONE event (their inputs are never overwritten inside a step and nothing but the step's final
# join reads their results, so a later release is only more of the lag the lane already has).  0 = release at once.  Only where the
# chain still has 2 x BULK_DEFER launches to go: held work released near the end of a list is a tail behind its last chain kernel (the
# short per-module lists of NewBigFRFNModel / AST lost 2-5 % with an unconditional hold).
BULK_DEFER = 45                # (30 until the end of round 5; 43 - 47 measure the same, 40 and 50 are 0.2 - 0.3 ms slower: ab_notes.md u)
FAST_LANE_EVENTS = True        # lane-ordering events without the system-scope fence (see _LaneEvent)
SIDE_LOW_PRIORITY = True       # side lane on the least-priority stream torch offers (on this ROCm build priority_range() is (0, -1): the
                               # default priority - no difference was ever measured, nor with the chain on a priority -1 stream)
FUSE_QK_GRAM = True            # MDTA forward at C <= 32: q k^T and the norms inside uir_qkv_fwd (UirQkvFwdArgs.gram_slab); False = uir_gram launch
FUSE_VALUE_GRAMS = True        # MDTA backward at C <= 32: dW_B / dW_A inside the dvf / dv GEMMs (EPI_PLAIN_DW); False = separate Gram launches
_WGRAD3_SHAPES = ((32, 8), (32, 16), (16, 8), (8, 4))   # (K, N) pairs csrc/conv3m.hip uir_conv3x3_wgrad accepts (with 16-byte rows); any other
                               # shape runs the generic nine-tap Gram, which needs ITS chunking (tests/test_cpu_surface.py holds the two lists together)
_WGRAD3_WGS = 512              # workgroups of the LDS-staged 3 x 3 weight-gradient kernel (two per CU)
CONV3_SPLIT = True             # 3x3 convolutions with >= 32 channels on both sides (UpSample1, down3): forward / input gradient / weight gradient on the
                               # split-bf16 implicit-GEMM kernels of csrc/conv3b.hip; False = the nine-tap f32 MFMA paths of uir_gemm / uir_gram
FUPS_PHA_SIDE_MAXROWS = 0      # UpSample's phase MLP on the side lane beside the amplitude MLP where a level has at most this many pixels (0: never)
FUSE_VALUE_BWD = True          # MDTA value path's backward at C <= 32 as ONE tile-fused launch (csrc/attnvb.hip); False = dvf GEMM + depthwise backward + dv GEMM
RECOMPUTE_PRE = True           # C <= 32: qkv_pre / gpre (the depthwise inputs) are not written by the forward; uir_dwln_bwd recomputes its centre values (XP = NULL)
FUSE_VALUE_BWD_64 = True       # the same at level 2 (C = 64: 8 x 8 tiles, channels in two chunks); False = two Grams + two contractions + the depthwise backward
FUSE_LN_BWD = True             # deep levels (C = 64 / 128): LayerNorm backward as the epilogue of the dt1 / dt2 contraction (UIR_EPI_LN_BWD); False = uir_layernorm_bwd launches
GRAM_SLAB_CAP = True           # see Recorder._chunks
GRAM_SPLIT = True              # long 1x1 weight-gradient contractions with a narrow side on csrc/gramb.hip (split-bf16, transposed LDS reads); the library takes
                               # them when a chunk is a whole number of its 32-row stages - False shapes the chunks for csrc/gram.hip's f32 kernels instead
_CONV3B_WGS = 512              # workgroups of a uir_conv3x3_split_wgrad launch (two 6-wave workgroups per CU)
_PLAIN_DW_WGS = 768            # workgroups of a GEMM with the fused per-sample Gram (EPI_PLAIN_DW): the resident set at 3 per CU
GEMM_SLAB_ROWS = 512           # = UIR_GEMM_SLAB_ROWS of include/uir_kernels.h (max workgroups of the GEMMs with fused reductions)
FUSED_DW_MINROWS = 16 * 32 * 32     # depthwise backward as ONE pass (dX + dW) from this many pixels on (every level of the headline step since the end of round 5:
                               # -0.17 ms, profiles/r05/ab_notes.md v; 16 * 128 * 128 before); below, two kernels on two lanes
_DFT2_CACHE: dict = {}


def dft2_tables(n: int, sign: float, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """MFMA operand tables of the two-stage (16 x N2) DFT (csrc/spectral.hip uir_dft2_kernel), float64-built.
    ta[s][lane]: stage A, rows (Re|Im, k1), contraction (re|im, n1), k-step s <-> k = 2s + (lane >> 5).
    tb[j][s][lane]: stage B for the k1 pair (a_j, a_j + 4): rows (w, Re|Im, k2), k-step s = 2 n2 + (re|im), the lane half
    selects w (block diagonal); angle 2 pi n2 (k1 + 16 k2) / n, i.e. the inter-stage twiddle is folded in."""
    key = (n, float(sign), str(device))
    if key not in _DFT2_CACHE:
        n2c = n // 16
        lane = np.arange(64)
        li, h2 = lane & 31, lane >> 5

        def cs(num, den):                      # exact residues, exact 0 / +-1 at multiples of pi/2
            r = np.asarray(num, dtype=np.int64) % den
            ang = 2.0 * np.pi * r.astype(np.float64) / den
            c, s_ = np.cos(ang), np.sin(ang)
            q = (4 * r) % den == 0
            c = np.where(q, np.round(c), c)
            s_ = np.where(q, np.round(s_), s_)
            return c, sign * s_
        ta = np.zeros((16, 64))
        for s_ in range(16):
            k = 2 * s_ + h2                    # contraction index: < 16 real input n1 = k, else imaginary n1 = k - 16
            out_im, k1 = li >= 16, li & 15
            in_im, n1 = k >= 16, k & 15
            c, sg = cs(n1 * k1, 16)
            ta[s_] = np.where(out_im == in_im, c, np.where(out_im, sg, -sg))
        tb = np.zeros((8, 2 * n2c, 64))
        w, p, k2 = li >> 4, (li >> 3) & 1, li & 7
        for j in range(8):
            aj = (j & 3) + 8 * (j >> 2)
            for s_ in range(2 * n2c):
                nn2, pp = s_ >> 1, s_ & 1
                c, sg = cs(nn2 * (aj + 4 * w + 16 * k2), n)
                v = np.where(p == pp, c, np.where(p == 1, sg, -sg))
                tb[j, s_] = np.where((w == h2) & (k2 < n2c), v, 0.0)
        _DFT2_CACHE[key] = (torch.from_numpy(ta.astype(np.float32)).to(device).contiguous(),
                            torch.from_numpy(tb.astype(np.float32)).to(device).contiguous())
    return _DFT2_CACHE[key]


def _ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def _ld(t: torch.Tensor) -> int:
    assert t.dim() == 2 and (t.stride(1) == 1 or t.shape[1] == 1), (t.shape, t.stride())
    return t.stride(0)


class _LaneEvent:
    """Orders one lane's stream after another's inside a launch list.  FAST_LANE_EVENTS: an event without the system-scope fence of a
    default event's record (include/uir_kernels.h uir_event_create) - every record / wait on the main stream is a barrier packet between
    two chain kernels (tools/trace_gaps.py), and a default record also writes the caches back for readers that do not exist here."""

    def __init__(self):
        self.h = None
        self.t = None
        if FAST_LANE_EVENTS:
            h = C.c_void_p()
            L.check(lib.uir_event_create(C.byref(h)), "uir_event_create")
            self.h = h
        else:
            self.t = torch.cuda.Event()

    def record_on(self, stream) -> None:
        if self.h is not None:
            L.check(lib.uir_event_record(self.h, C.c_void_p(stream.cuda_stream)), "uir_event_record")
        else:
            self.t.record(stream)

    def make_wait(self, stream) -> None:
        if self.h is not None:
            L.check(lib.uir_stream_wait_event(C.c_void_p(stream.cuda_stream), self.h), "uir_stream_wait_event")
        else:
            stream.wait_event(self.t)

    def __del__(self):
        if self.h is not None and lib is not None:
            try:
                lib.uir_event_destroy(self.h)
            except Exception:
                pass


class Recorder:
    """Kernel-level recorder: buffers + an ordered list of C-ABI launches with pre-built
    argument structs.  `Plan` (below) records the whole network with it; the GPU unit tests
    drive single kernels through the same methods."""

    def __init__(self, device, fp: "FlatParams | None" = None, scratch_floats: int = SCRATCH_FLOATS):
        self.fp = fp
        self.dev = torch.device(device)
        self.fwd: List[tuple] = []
        self.bwd: List[tuple] = []
        self.cur = self.fwd
        self.keep: List[object] = []  # ctypes structs / tensors referenced by raw pointer
        self.bytes = 0
        # launches that depend on the weights alone (MDTA's P = W_kv[vf] W_o): recorded apart and spliced into the forward list early
        # (Plan._build), all on the side lane behind one event and one join - per block they cost the chain two barrier packets
        self.pro: List[tuple] = []
        self.meta: Dict[int, list] = {id(self.fwd): [], id(self.bwd): [], id(self.pro): []}
        # per launch: the device kernel its arguments select ("uir_gemm<taps=1,pro=1,epi=0>" ...), for per-kernel reports
        self.variants: Dict[int, list] = {id(self.fwd): [], id(self.bwd): [], id(self.pro): []}
        self.scratch = self.alloc(scratch_floats)
        # lane 0 = the critical dX chain (current stream); lane 1 = weight-gradient work that nothing
        # on the chain waits for, run on a second HIP stream with its own slab scratch
        self.lane = 0
        self.scratch_side = self.alloc(scratch_floats)
        self.lanes: Dict[int, list] = {id(self.fwd): [], id(self.bwd): [], id(self.pro): []}
        self._side_stream = None
        self._bulk_stream = None
        self._events: List[object] = []        # lane-ordering events inside a list (_LaneEvent)
        self._join_events: List[object] = []   # the two joins at the end of a list: torch events (default fences)
        self._pending: List[tuple] = []        # bulk-lane slab reductions waiting for the stage's batched launch

    # ------------------------------------------------------------------ memory
    def alloc(self, *shape, zero: bool = False) -> torch.Tensor:
        t = (torch.zeros if zero else torch.empty)(*shape, dtype=torch.float32, device=self.dev)
        self.bytes += t.numel() * 4
        self.keep.append(t)
        return t

    def P(self, name):
        return self.fp.view(name)

    def Gr(self, name):
        return self.fp.gview(name)

    # --------------------------------------------------------------- recording
    def _op(self, name: str, fn, *args, nbytes: int = 0, flops: int = 0, variant: "str | None" = None):
        """Record one launch.  nbytes/flops = ALGORITHMIC HBM bytes / flops of that launch
        (each operand counted once; used by the roofline report, never by the launch)."""
        self.cur.append((fn, args, name))
        self.meta[id(self.cur)].append((name, fn.__name__, nbytes, flops))
        self.lanes[id(self.cur)].append(self.lane)
        self.variants[id(self.cur)].append(variant or fn.__name__)

    # ---- slab reductions.  On the weight-gradient (bulk) lane nothing reads a reduced gradient before the optimizer, so they
    # are collected and issued as ONE uir_reduce_slab_batch launch per network stage (flushed at every stage mark): ~380
    # launches of 7-30 us per step become ~75.  Bit-identical to the per-tensor launches (same summation order).
    def _batching(self) -> bool:
        return self.lane == 4

    def _reduce(self, name: str, src: int, rows: int, stride: int, dst: int, n: int, immediate: bool = False):
        if self._batching() and not immediate:
            self._pending.append((name, int(src), int(rows), int(stride), int(dst), int(n)))
            return
        self._op(name, lib.uir_reduce_slab, src, rows, stride, dst, n, 0, 1.0)

    def _private_slab(self, floats: int) -> torch.Tensor:
        """Partial-slab buffer that stays intact until the stage's batched reduction has run (the shared scratch of a lane
        is overwritten by that lane's next kernel)."""
        return self.alloc(floats)

    def flush_reductions(self, tag: str = "stage") -> None:
        if not self._pending:
            return
        pend, self._pending = self._pending, []
        self._reduce_batch(tag + ".red_batch", pend, 4)

    def _reduce_batch(self, name: str, pend: "List[tuple]", lane: int) -> None:
        """ONE uir_reduce_slab_batch launch for a list of (name, slab ptr, rows, row stride, out ptr, n) reductions."""
        descs = (L.ReduceDesc * len(pend))()
        first, tot = [], 0
        for d, (_nm, src, rows, stride, dst, n) in zip(descs, pend):
            d.slab, d.out, d.stride, d.n, d.nchunks, d.accumulate, d.scale = src, dst, stride, n, rows, 0, 1.0
            first.append(tot)
            tot += (n + 63) // 64
        dt = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.dev)
        ft = torch.tensor(first, dtype=torch.int32).to(self.dev)
        self.keep += [dt, ft]
        prev, self.lane = self.lane, lane
        self._op(name, lib.uir_reduce_slab_batch, dt.data_ptr(), ft.data_ptr(), len(pend), tot,
                 variant=f"uir_reduce_slab_batch")
        self.lane = prev

    class _Side:
        def __init__(self, rec):
            self.rec = rec

        def __enter__(self):
            self.prev = self.rec.lane
            self.rec.lane = 1

        def __exit__(self, *a):
            self.rec.lane = self.prev

    class _Prologue:
        def __init__(self, rec):
            self.rec = rec

        def __enter__(self):
            self.prev = (self.rec.lane, self.rec.cur)
            self.rec.lane, self.rec.cur = 1, self.rec.pro

        def __exit__(self, *a):
            self.rec.lane, self.rec.cur = self.prev

    def prologue(self):
        return Recorder._Prologue(self)

    def splice_prologue(self, at: int) -> None:
        """Move the weight-only launches to position `at` of the forward list (once, at the end of the forward's construction)."""
        for d in (self.meta, self.lanes, self.variants):
            d[id(self.fwd)][at:at] = d[id(self.pro)]
            del d[id(self.pro)][:]
        self.fwd[at:at] = self.pro
        del self.pro[:]

    class _Bulk(_Side):
        def __enter__(self):
            self.prev = self.rec.lane
            self.rec.lane = 4

    def bulk(self):
        """Context: weight-gradient work.  Runs on a third stream, joined only at the end of the list, and
        may be re-scheduled later in the list (every buffer of the plan is written once, so it can
        start any time after its inputs exist)."""
        return Recorder._Bulk(self)

    def side(self):
        """Context: ops recorded inside run on the side stream (they may READ anything produced by
        earlier ops of the list; nothing on the main lane may read what they WRITE before the join
        at the end of the list)."""
        return Recorder._Side(self)

    def mark(self, name: str) -> None:
        """Stage marker: `run(ops, marks=[])` records an event on the main stream here (timeline of
        the real, overlapped execution; nothing is serialised)."""
        self.flush_reductions(name)
        self.cur.append((None, (), name))
        self.meta[id(self.cur)].append((name, "mark", 0, 0))
        self.lanes[id(self.cur)].append(3)
        self.variants[id(self.cur)].append(None)

    def join(self, name: str = "join") -> None:
        """Mid-list join: the main lane waits for everything recorded so far on the side lane."""
        self.cur.append((None, (), name))
        self.meta[id(self.cur)].append((name, "join", 0, 0))
        self.lanes[id(self.cur)].append(2)
        self.variants[id(self.cur)].append(None)

    def _slab(self) -> torch.Tensor:
        return self.scratch_side if self.lane in (1, 4) else self.scratch

    def _hold(self, *tensors):
        """Raw pointers are recorded: keep every operand alive for the lifetime of the plan."""
        for t in tensors:
            if isinstance(t, dict):
                self.keep.extend(t.values())
            elif isinstance(t, (tuple, list)):
                self.keep.extend(x for x in t if x is not None)
            elif t is not None:
                self.keep.append(t)

    def run(self, ops: Sequence[tuple], marks: "list | None" = None, serial: bool = False,
            timed: "list | None" = None, join_bulk: bool = True) -> None:
        """Execute a launch list.  serial=True ignores the lanes (everything on the current stream, in list
        order): the reference schedule the overlapped execution must reproduce bit for bit.
        timed=[]: a HIP event pair is recorded around every launch ON THE STREAM IT IS LAUNCHED ON, in the real
        three-lane execution (nothing is serialised across lanes); entries (op index, start event, end event)."""
        if self._pending:              # a recorder driven op by op (unit tests): nothing may stay un-reduced
            self.flush_reductions("auto")
        if self.dev.type != "cuda":
            st = C.c_void_p(0)
            for fn, args, name in ops:
                if fn is None:
                    continue
                rc = fn(*args, st)
                if rc:
                    L.check(rc, name)
            return
        if self.dev.index is not None and torch.cuda.current_device() != self.dev.index:
            # raw device pointers + the CURRENT device's stream: a mismatch would run kernels on another GPU's queue
            raise RuntimeError(f"plan lives on {self.dev} but the current device is cuda:{torch.cuda.current_device()}: "
                               "call torch.cuda.set_device(...) first (one process per GPU)")
        main = torch.cuda.current_stream()
        st0 = C.c_void_p(main.cuda_stream)
        lanes = self.lanes[id(ops)]
        if serial or (1 not in lanes and 4 not in lanes):
            for fn, args, name in ops:
                if fn is None:
                    continue
                rc = fn(*args, st0)
                if rc:
                    L.check(rc, name)
            return
        if self._side_stream is None:
            # lowest priority: the side lane fills bandwidth the critical dX chain leaves idle
            lo = max(torch.cuda.Stream.priority_range()) if hasattr(torch.cuda.Stream, "priority_range") else 0
            self._side_stream = torch.cuda.Stream(device=self.dev, priority=lo if SIDE_LOW_PRIORITY else 0)
        side = self._side_stream
        st1 = C.c_void_p(side.cuda_stream)
        if self._bulk_stream is None:
            self._bulk_stream = torch.cuda.Stream(device=self.dev)
        bulk = self._bulk_stream
        st4 = C.c_void_p(bulk.cuda_stream)
        nev = 0
        main_dirty = True           # main has launched work the side stream has not been ordered after
        bulk_dirty = True           # ... the bulk stream has not been ordered after
        side_dirty = False          # side has launched work the bulk stream has not been ordered after
        held: list = []             # bulk launches not yet released (BULK_DEFER)
        held_age = 0                # chain launches since the first of them
        chain_left = sum(1 for ln in lanes if ln in (0, 1))      # chain / side launches still to come

        def release_bulk():
            nonlocal nev, bulk_dirty, side_dirty, held_age
            if bulk_dirty:
                if nev == len(self._events):
                    self._events.append(_LaneEvent())
                ev = self._events[nev]
                nev += 1
                ev.record_on(main)
                ev.make_wait(bulk)
                bulk_dirty = False
            if side_dirty:      # bulk work may consume side-lane results (e.g. the Sups weight gradient reads `ds`)
                if nev == len(self._events):
                    self._events.append(_LaneEvent())
                ev = self._events[nev]
                nev += 1
                ev.record_on(side)
                ev.make_wait(bulk)
                side_dirty = False
            for hopi, hfn, hargs, hname in held:
                if timed is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(bulk)
                rc = hfn(*hargs, st4)
                if timed is not None:
                    e1.record(bulk)
                    timed.append((hopi, e0, e1))
                if rc:
                    L.check(rc, hname)
            held.clear()
            held_age = 0

        for opi, ((fn, args, name), lane) in enumerate(zip(ops, lanes)):
            if lane == 3:                       # stage marker
                if marks is not None:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record(main)
                    marks.append((name, e))
                continue
            if lane == 2:                       # join marker
                if nev == len(self._events):
                    self._events.append(_LaneEvent())
                ev = self._events[nev]
                nev += 1
                ev.record_on(side)
                ev.make_wait(main)
                continue
            if lane == 4:
                held.append((opi, fn, args, name))
                if BULK_DEFER <= 0 or chain_left < 2 * BULK_DEFER:
                    release_bulk()
                continue
            chain_left -= 1
            if held:
                if held_age >= BULK_DEFER or chain_left < 2 * BULK_DEFER:
                    release_bulk()
                else:
                    held_age += 1
            if lane == 1:
                if main_dirty:
                    if nev == len(self._events):
                        self._events.append(_LaneEvent())
                    ev = self._events[nev]
                    nev += 1
                    ev.record_on(main)
                    ev.make_wait(side)
                    main_dirty = False
                if timed is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(side)
                rc = fn(*args, st1)
                if timed is not None:
                    e1.record(side)
                    timed.append((opi, e0, e1))
                side_dirty = True
            else:
                if timed is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(main)
                rc = fn(*args, st0)
                if timed is not None:
                    e1.record(main)
                    timed.append((opi, e0, e1))
                main_dirty = True
                bulk_dirty = True
            if rc:
                L.check(rc, name)
        if held:
            release_bulk()
        while len(self._join_events) < 2:
            self._join_events.append(torch.cuda.Event())
        ev = self._join_events[0]
        ev.record(side)
        main.wait_event(ev)         # join: everything after this list sees the side lane's results
        if 4 in lanes:
            if join_bulk:
                ev2 = self._join_events[1]
                ev2.record(bulk)
                main.wait_event(ev2)
                self._bulk_unjoined = False
            else:
                # the weight-gradient lane runs on past the end of this list: the caller joins it (join_bulk()) before anything reads a
                # gradient or overwrites an activation - a network of many short lists (one per module) otherwise waits out the lane's
                # tail at every module boundary (NewBigFRFNModel: 27 gaps of 60-270 us per step, rocprofv3 trace)
                self._bulk_unjoined = True

    def join_bulk(self) -> None:
        """Order the current stream after everything this recorder's bulk lane has been given (see run(join_bulk=False))."""
        if getattr(self, "_bulk_unjoined", False) and self._bulk_stream is not None:
            while len(self._join_events) < 2:
                self._join_events.append(torch.cuda.Event())
            ev2 = self._join_events[1]
            ev2.record(self._bulk_stream)
            torch.cuda.current_stream().wait_event(ev2)
        self._bulk_unjoined = False

    def run_timed(self, ops: Sequence[tuple]) -> "List[Tuple[str, str, int, int, float]]":
        """Run a launch list with a HIP event pair around EVERY launch (on the stream the kernels
        are launched on).  Returns [(op name, C symbol, algorithmic bytes, flops, milliseconds)].
        Diagnostic only: the event pairs serialise the list."""
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(ops) + 1)]
        evs[0].record()
        for i, (fn, args, name) in enumerate(ops):
            if fn is not None:
                rc = fn(*args, st)
                if rc:
                    L.check(rc, name)
            evs[i + 1].record()
        torch.cuda.synchronize()
        meta = self.meta[id(ops)]
        return [(m[0], m[1], m[2], m[3], evs[i].elapsed_time(evs[i + 1])) for i, m in enumerate(meta)]

    # ---------------------------------------------------------------- kernels
    def gemm(self, name, X, W, Y, *, N, K, w_sn, w_sk, w_st=0, w_bs=0, batches=1, bias=None, R=None,
             taps=1, flip=0, hw=None, n_grp=(0, 0), k_grp=(0, 0), pro=L.PRO_NONE, gate_off=0, ln=None, stats=None,
             epi=L.EPI_PLAIN, G=None, g_off=0, act=L.ACT_NONE, dw_slab=None, pre_out=None, slope=0.1, out_scale=None, scale_rows=0,
             il16=0, dw_cols=0, dw_rows=0):
        M_total = X.shape[0]
        assert M_total % batches == 0 and Y.shape[0] == M_total
        M = M_total // batches
        self._hold(X, W, Y, bias, R, ln, stats, G)
        a = L.GemmArgs()
        a.X, a.W, a.Y = _ptr(X), _ptr(W), _ptr(Y)
        a.bias, a.R = _ptr(bias), _ptr(R)
        a.ldx, a.ldy = _ld(X), _ld(Y)
        a.ldr = _ld(R) if R is not None else 0
        a.x_bs, a.y_bs = M * a.ldx, M * a.ldy
        a.r_bs = M * a.ldr
        a.w_bs, a.w_sn, a.w_sk, a.w_st = w_bs, w_sn, w_sk, w_st
        a.M, a.N, a.K, a.batches = M, N, K, batches
        a.taps, a.flip = taps, flip
        if taps == 9:
            a.H, a.Wd = hw
            assert M % (hw[0] * hw[1]) == 0
        a.n_grp_real, a.n_grp_pad = n_grp
        a.k_grp_real, a.k_grp_pad = k_grp
        a.pro, a.gate_off = pro, gate_off
        if pro == L.PRO_LN:
            a.ln_g, a.ln_b = _ptr(ln[0]), _ptr(ln[1])
            a.stats = _ptr(stats)
        if epi == L.EPI_LN_BWD:          # LayerNorm backward on the accumulators: G = the norm's input, ln = (its weight,), stats = (mean, rstd)
            assert pro == L.PRO_NONE and G is not None and stats is not None and dw_slab is not None and N in (64, 128)
            a.ln_g, a.stats = _ptr(ln[0]), _ptr(stats)
        a.epi, a.g_off, a.act = epi, g_off, act
        a.il16 = il16
        a.dw_cols, a.dw_rows = dw_cols, dw_rows
        if G is not None:
            a.G, a.ldg = _ptr(G), _ld(G)
            a.g_bs = M * a.ldg
        a.slope, a.ln_eps = slope, LN_EPS
        if pre_out is not None:       # ACT_GELU: the pre-activation is stored too (needed by the GELU backward)
            assert act == L.ACT_GELU and batches == 1 and pre_out.shape[0] == M_total
            self._hold(pre_out)
            a.pre_out, a.ldpre = _ptr(pre_out), _ld(pre_out)
        if dw_slab is not None:
            self._hold(dw_slab)
            a.dw_slab, a.dw_stride = _ptr(dw_slab), dw_slab.stride(0)
        if out_scale is not None:      # Y = R + out_scale[row // scale_rows] * (X.W^T + bias): a residual branch's DropPath factor
            assert R is not None and epi == L.EPI_PLAIN and taps == 1 and batches == 1 and pre_out is None and scale_rows >= 32 and M % scale_rows == 0
            assert out_scale.numel() * scale_rows == M
            self._hold(out_scale)
            a.out_scale, a.scale_rows = _ptr(out_scale), scale_rows
        assert a.ldx % 4 == 0 and a.ldx >= pad4(K) and a.ldy >= N, (name, a.ldx, K, a.ldy, N)
        assert Y.shape[1] >= N or epi in (L.EPI_GATE_BWD, L.EPI_GATE_BWD_DW), (name, Y.shape, N)
        self.keep.append(a)
        rows = M * batches
        kreal = k_grp[0] * (K // k_grp[1]) if k_grp[1] else K
        nreal = n_grp[0] * (N // n_grp[1]) if n_grp[1] else N
        gcols = dw_cols if epi == L.EPI_PLAIN_DW else N * (2 if epi in (L.EPI_GATE_BWD, L.EPI_GATE_BWD_DW) else 1)
        nb = 4 * rows * (K * (2 if pro == L.PRO_GATE else 1) + N * (2 if epi in (L.EPI_GATE_BWD, L.EPI_GATE_BWD_DW) else 1)
                         + (N if R is not None else 0) + (gcols if G is not None else 0)
                         + (N if pre_out is not None else 0))
        self._op(name, lib.uir_gemm, C.byref(a), nbytes=nb, flops=2 * rows * kreal * nreal * taps,
                 variant=f"uir_gemm1x1_kernel<*,{pro},{epi},*>")     # the rocprofv3 symbol family (NT and HAS_R vary per call)

    def _chunks(self, M, ngroups, batches, split_nk=None):
        """(chunks, rows per chunk) of a pixel contraction.  split_nk = (N, K) for the cases csrc/gramb.hip takes (gram_split_case: the library
        decides which kernel runs, this only shapes the chunks): its workgroups own 128 x 64 / 128 x 128 blocks, four of them fit a CU and a
        chunk is a whole number of 32-row stages."""
        if split_nk is not None:
            big, small = max(split_nk), min(split_nk)
            blocks = ((big + 127) // 128) * (1 if small <= 64 else (small + 127) // 128)
            nch = max(1, min(M // 128, _GRAM_WGS // max(1, blocks * batches)))
            # the partial slab is written here and read by the reduction: no more of it than half the operands' own bytes (level 3: M = 16 384
            # rows against a 704 x 128 output - 128 chunks were 46 MB of slab per Gram for 54 MB of operands, and the stage's batched reduction
            # 123 us of the weight-gradient lane)
            if GRAM_SLAB_CAP:
                nch = max(1, min(nch, (M * (split_nk[0] + split_nk[1])) // (2 * split_nk[0] * split_nk[1])))
            rpc = ((M + nch - 1) // nch + 31) & ~31
        else:
            nch = max(1, min(M // 256, _GRAM_WGS // max(1, ngroups * batches)))
            rpc = ((M + nch - 1) // nch + 7) & ~7
            if not GRAM_SPLIT and rpc % 32 == 0:
                rpc += 8                       # A/B arm: never a whole number of csrc/gramb.hip's stages, so the f32 kernels keep the launch
        nch = (M + rpc - 1) // rpc
        return nch, rpc

    @staticmethod
    def gram_split_case(M, N, K, taps=1, extras=False):
        return (N, K) if (GRAM_SPLIT and taps == 1 and not extras and M >= 1024 and N >= 32 and K >= 32 and not (N >= 256 and K >= 256)) else None

    def gram(self, name, A, Bm, out, *, N, K, o_sn, o_sk, o_st=0, o_bs=0, batches=1, taps=1, hw=None,
             n_grp=(0, 0), k_grp=(0, 0), pro_b=0, gate_off=0, ln=None, stats=None, out_elems=None,
             extras: "Dict[str, torch.Tensor] | None" = None, fused_out: "torch.Tensor | None" = None,
             own_slab: bool = False, shared_slab=None):
        """out[...] = sum_m A[m][n] * pro(B[nbr(m)][k]); `out` is a flat view of out_elems floats.
        extras: {'sq_a': tensor[batches*N], 'sq_b': tensor[batches*K], 'cs_a': tensor[batches*N]}."""
        M_total = A.shape[0]
        M = M_total // batches
        self._hold(A, Bm, out, ln, stats, extras)
        tn, tk = (N + 31) // 32, (K + 31) // 32
        if tk == 1 and tn == 1:
            groups = 1
        elif tk == 1:
            groups = (tn + 3) // 4
        elif tn == 1:
            groups = (tk + 3) // 4
        else:
            groups = ((tn + 1) // 2) * ((tk + 1) // 2)
        if (taps == 9 and batches == 1 and not extras and pro_b == 0 and (K, N) in _WGRAD3_SHAPES and _ld(A) % 4 == 0 and _ld(Bm) % 4 == 0
                and _ld(Bm) >= K and _ptr(A) % 16 == 0 and _ptr(Bm) % 16 == 0):
            # csrc/conv3m.hip uir_conv3x3_wgrad: persistent workgroups over 16 x 16 pixel tiles, one slab row each
            nch = max(1, min(batches * ((hw[0] + 15) // 16) * ((hw[1] + 15) // 16) * (M // (hw[0] * hw[1])), _WGRAD3_WGS))
            rpc = ((M + nch - 1) // nch + 7) & ~7
        elif taps == 9 and tn == 1 and tk == 1 and not extras and pro_b == 0:
            nch, rpc = self._chunks(M, 1, batches)          # all 9 taps handled inside one workgroup
        elif (taps == 1 and batches == 1 and pro_b == 0 and not extras and min(N, K) >= 128 and not (n_grp[1] or k_grp[1])
              and shared_slab is None and fused_out is None):
            # wide weight gradient on csrc/tiled.hip (128 x 128 output tiles, one k split per chunk): enough chunks for ~3 workgroups
            # per CU, at least 512 token rows each
            t128 = ((N + 127) // 128) * ((K + 127) // 128)
            nch = max(1, min(M // 512, (_GRAM_WGS * 3 // 4 + t128 - 1) // t128))
            rpc = ((M + nch - 1) // nch + 7) & ~7
            nch = (M + rpc - 1) // rpc
        else:
            nch, rpc = self._chunks(M, taps * groups, batches, split_nk=self.gram_split_case(M, N, K, taps, bool(extras)))
        if out_elems is None:
            out_elems = out.numel()
        a = L.GramArgs()
        a.A, a.B, a.slab = _ptr(A), _ptr(Bm), _ptr(self._slab())
        a.lda, a.ldb = _ld(A), _ld(Bm)
        a.a_bs, a.b_bs = M * a.lda, M * a.ldb
        a.o_sn, a.o_sk, a.o_st, a.o_bs = o_sn, o_sk, o_st, o_bs
        stride = out_elems
        a.sq_a_off = a.sq_b_off = a.cs_a_off = -1
        extras = extras or {}
        if "sq_a" in extras:
            a.sq_a_off = stride; stride += batches * N
        if "sq_b" in extras:
            a.sq_b_off = stride; stride += batches * K
        if "cs_a" in extras:
            a.cs_a_off = stride; stride += batches * N
        a.slab_stride = stride
        if shared_slab is not None:
            # (tensor, column offset, chunk stride): several Grams write side by side into one slab and the caller issues
            # ONE reduction launch for all of them; returns nch so the caller can check that the chunking agrees
            slab_t, col0, cstride = shared_slab
            assert fused_out is None and col0 + stride <= cstride and nch * cstride <= slab_t.numel()
            a.slab = slab_t.data_ptr() + 4 * col0
            a.slab_stride = cstride
            self._hold(slab_t)
        elif own_slab or self._batching():     # private partial-slab buffer: the launch may run on a lane that shares its
            slab_t = self.alloc(nch * stride)  # scratch with another / its reduction is deferred to the stage's batch
            a.slab = _ptr(slab_t)
        else:
            slab_t = self._slab()
            assert nch * stride <= slab_t.numel(), (name, nch, stride)
        a.M, a.N, a.K, a.batches = M, N, K, batches
        a.n_grp_real, a.n_grp_pad = n_grp
        a.k_grp_real, a.k_grp_pad = k_grp
        a.taps = taps
        if taps == 9:
            a.H, a.Wd = hw
        a.pro_b, a.gate_off = pro_b, gate_off
        if pro_b == L.PRO_LN:
            a.stats, a.ln_g, a.ln_b = _ptr(stats), _ptr(ln[0]), _ptr(ln[1])
        a.rows_per_chunk, a.nchunks = rpc, nch
        assert a.lda >= N and a.ldb >= K
        self.keep.append(a)
        rows = M * batches
        self._op(name, lib.uir_gram, C.byref(a), nbytes=4 * rows * (N + K * (2 if pro_b == L.PRO_GATE else 1)),
                 flops=2 * rows * N * K * taps, variant="uir_gram_kernel<*>" if taps == 1 or pro_b or extras else "uir_gram_taps9*")
        if shared_slab is not None:
            return nch
        sp = slab_t.data_ptr()
        if fused_out is not None:
            # `fused_out` = [out | sq_a | sq_b | cs_a] contiguous, i.e. exactly one slab row: one reduction launch
            assert fused_out.numel() == stride and fused_out.data_ptr() == out.data_ptr()
            self._hold(fused_out)
            self._reduce(name + ".red", sp, nch, stride, _ptr(fused_out), stride)
            return
        self._reduce(name + ".red", sp, nch, stride, _ptr(out), out_elems)
        for key, off in (("sq_a", a.sq_a_off), ("sq_b", a.sq_b_off), ("cs_a", a.cs_a_off)):
            if key in extras:
                t = extras[key]
                self._reduce(name + ".red." + key, sp + 4 * off, nch, stride, _ptr(t), t.numel())

    # ---- 3 x 3 convolutions with >= 32 channels on both sides on split-bf16 MFMA (csrc/conv3b.hip): weight planes once per step,
    # the halo'd pixel tile in LDS once per 32-channel chunk, nine taps as row offsets
    @staticmethod
    def conv3_split_ok(N: int, K: int, dims) -> bool:
        """uir_conv3x3_split's contract for N outputs / K inputs on dims = (B, H, W)."""
        _B, H, W = dims
        return (CONV3_SPLIT and knobs.flag("UIR_FUSED_SPLIT") and N >= 32 and K >= 32 and N % 32 == 0 and K % 32 == 0 and W % 16 == 0
                and H % (4 if N % 64 == 0 else 8) == 0)

    @staticmethod
    def conv3_split_wgrad_ok(N: int, K: int, dims) -> bool:
        _B, H, W = dims
        return CONV3_SPLIT and knobs.flag("UIR_FUSED_SPLIT") and N >= 64 and K >= 32 and N % 64 == 0 and K % 32 == 0 and W % 16 == 0 and H % 4 == 0

    def conv3_wsplit(self, name, Wt, Cout, Cin, want_bwd: bool):
        """bf16 planes of a native [Cout][Cin][3][3] weight: (forward planes, input-gradient planes or None); one launch per step."""
        n16 = 3 * 9 * Cout * Cin
        pf = self.alloc((n16 + 1) // 2)
        pb = self.alloc((n16 + 1) // 2) if want_bwd else None
        self._hold(Wt)
        self._op(name, lib.uir_conv3x3_wsplit, _ptr(Wt), Cout, Cin, _ptr(pf), _ptr(pb), nbytes=4 * Cout * Cin * 9 + 2 * n16 * (2 if want_bwd else 1),
                 variant="uir_conv3b_wsplit")
        return pf, pb

    def conv3_split(self, name, X, planes, Y, *, N, K, dims, R=None, tile=0):
        B, H, W = dims
        assert X.shape[0] == B * H * W and Y.shape[0] == B * H * W and self.conv3_split_ok(N, K, dims)
        self._hold(X, planes, Y, R)
        a = L.Conv3bArgs()
        a.X, a.Wp, a.Y, a.R = _ptr(X), _ptr(planes), _ptr(Y), _ptr(R)
        a.ldx, a.ldy, a.ldr = _ld(X), _ld(Y), (_ld(R) if R is not None else 0)
        a.B, a.H, a.Wd, a.N, a.K, a.tile = B, H, W, N, K, tile
        self.keep.append(a)
        rows = B * H * W
        self._op(name, lib.uir_conv3x3_split, C.byref(a), nbytes=4 * rows * (K + N + (N if R is not None else 0)), flops=2 * rows * N * K * 9,
                 variant="uir_conv3b_kernel<*>")

    def conv3_split_wgrad(self, name, dY, X, out, *, N, K, dims):
        """out (native [N][K][3][3], contiguous) = the weight gradient; partial slab rows + one reduction like uir_gram."""
        B, H, W = dims
        assert dY.shape[0] == B * H * W and X.shape[0] == B * H * W and self.conv3_split_wgrad_ok(N, K, dims) and out.numel() == N * K * 9
        self._hold(dY, X, out)
        ntiles = B * (H // 4) * (W // 16)
        nch = max(1, min(ntiles, _CONV3B_WGS // ((N // 64) * (K // 32))))
        stride = N * K * 9
        if self._batching():
            slab_t = self.alloc(nch * stride)
        else:
            slab_t = self._slab()
            assert nch * stride <= slab_t.numel(), (name, nch, stride)
        a = L.Conv3bWgradArgs()
        a.dY, a.X, a.slab, a.slab_stride = _ptr(dY), _ptr(X), _ptr(slab_t), stride
        a.ldy, a.ldx = _ld(dY), _ld(X)
        a.B, a.H, a.Wd, a.N, a.K, a.nchunks = B, H, W, N, K, nch
        self.keep.append(a)
        rows = B * H * W
        self._op(name, lib.uir_conv3x3_split_wgrad, C.byref(a), nbytes=4 * rows * (N + K), flops=2 * rows * N * K * 9, variant="uir_conv3b_wgrad_kernel")
        self._reduce(name + ".red", slab_t.data_ptr(), nch, stride, _ptr(out), stride)

    # ---- wide 3 x 3 convolutions on few pixels as ONE plain contraction over patches (uir_patch3x3: k = cin * 9 + tap, the native
    # weight's own order, so weight.view(Cout, 9 Cin) and its gradient are used in place).  The in-place nine-tap kernels tile the
    # OUTPUT (128 x 128): 2 048 pixels x 1 024 channels are 128 workgroups and the flipped-tap weights of an input gradient are read
    # 4 bytes at a time (NewBigFRFNModel upsample_3: 0.55 / 1.39 ms forward / input gradient for 19 GFLOP each); the patch matrix
    # is M x 9 Cin floats (38 MB there) and the K = 9 Cin GEMM splits k (csrc/tiled.hip).
    @staticmethod
    def conv_as_patches(M: int, Cin: int, Cout: int) -> bool:
        return Cin >= 256 and Cin % 4 == 0 and Cout >= 128 and Cout % 4 == 0 and 256 <= M <= 16384

    def patches3x3(self, name, x, Cin, dims):
        B, H, W = dims
        P = self.alloc(B * H * W, 9 * Cin)
        self._hold(x)
        self._op(name, lib.uir_patch3x3, _ptr(x), _ld(x), _ptr(P), _ld(P), B, H, W, Cin, nbytes=4 * (x.shape[0] * Cin + P.numel()))
        return P

    def conv3x3_dx_from_patches(self, name, dy, Wt, dx, Cin, Cout, dims):
        """dx = the input gradient of a 3 x 3 convolution: dP = dy . weight.view(Cout, 9 Cin), folded back by uir_patch3x3_bwd."""
        B, H, W = dims
        dP = self.alloc(B * H * W, 9 * Cin)
        self.gemm(name, dy, Wt, dP, N=9 * Cin, K=Cout, w_sn=1, w_sk=9 * Cin)
        self._hold(dx)
        self._op(name + ".fold", lib.uir_patch3x3_bwd, _ptr(dP), _ld(dP), _ptr(dx), _ld(dx), B, H, W, Cin, nbytes=4 * (dP.numel() + dx.shape[0] * Cin))

    def dwconv(self, name, X, Wt, Y, *, C_, dims, grp=(0, 0), flip=0, act_in=0, bias=None, G=None):
        """act_in = 1: X is the pre-activation of a GELU applied on load; G (with bias): second output gelu(Y + bias)."""
        self._hold(X, Wt, Y, bias, G)
        a = L.DwArgs()
        a.X, a.Wt, a.Y = _ptr(X), _ptr(Wt), _ptr(Y)
        a.ldx, a.ldy = _ld(X), _ld(Y)
        a.B, a.H, a.Wd = dims
        a.C = C_
        a.grp_real, a.grp_pad = grp
        a.flip = flip
        a.act_in = act_in
        if G is not None:
            assert act_in == 1 and bias is not None
            a.bias, a.G, a.ldg = _ptr(bias), _ptr(G), _ld(G)
        assert C_ % 4 == 0 and a.ldx >= C_ and a.ldy >= C_ and X.shape[0] == dims[0] * dims[1] * dims[2]
        self.keep.append(a)
        rows = dims[0] * dims[1] * dims[2]
        self._op(name, lib.uir_dwconv, C.byref(a), nbytes=(8 + (4 if G is not None else 0)) * rows * C_, flops=18 * rows * C_)

    def gdfn_fwd(self, name, X, ln, Win, Wdw, Wout, Y, *, C_, h, hp, dims, T1=None, T2=None, stats=None, eps=1e-5, il=0):
        """Tile-fused GDFN forward (csrc/gdfn.hip): Y = X + W_out.(gelu(dw(W_in.LN(X))[:h]) * dw(...)[h:])."""
        self._hold(X, ln[0], ln[1], Win, Wdw, Wout, Y, T1, T2, stats)
        a = L.GdfnFwdArgs()
        a.X, a.ln_g, a.ln_b, a.Win, a.Wdw, a.Wout, a.Y = (_ptr(X), _ptr(ln[0]), _ptr(ln[1]), _ptr(Win), _ptr(Wdw),
                                                           _ptr(Wout), _ptr(Y))
        a.T1 = _ptr(T1) if T1 is not None else None
        a.T2 = _ptr(T2) if T2 is not None else None
        a.stats = _ptr(stats) if stats is not None else None
        a.ldx, a.ldy = _ld(X), _ld(Y)
        a.ldt = _ld(T2) if T2 is not None else 0
        a.B, a.H, a.Wd = dims
        a.C, a.h, a.hp = C_, h, hp
        a.ln_eps = eps
        a.il = il
        rows = dims[0] * dims[1] * dims[2]
        assert X.shape[0] == rows and C_ in (16, 32, 64, 128) and hp % 16 == 0 and hp >= h
        # T2 alone: the backward recomputes the pre-depthwise tensor (C <= 32, uir_dwln_bwd with XP = None)
        assert (T1 is None or T2 is not None) and (T2 is None or _ld(T2) >= 2 * hp) and (T1 is None or _ld(T1) == _ld(T2))
        self.keep.append(a)
        stored = (8 * hp if T1 is not None else 0) + (8 * hp if T2 is not None else 0)
        self._op(name, lib.uir_gdfn_fwd, C.byref(a), nbytes=rows * (12 * C_ + stored),
                 flops=rows * (2 * C_ * 2 * hp + 18 * 2 * hp + 2 * hp * C_))

    @staticmethod
    def qkv_tiles(dims) -> int:
        """Tiles per image of uir_qkv_fwd at C <= 32 (12 x 16 pixels): rows per sample of its Gram slab."""
        return ((dims[1] + 11) // 12) * ((dims[2] + 15) // 16)

    def qkv_fwd(self, name, X, ln, Wqkv, Wdw, Ypre, Y, *, C_, dims, stats=None, eps=1e-5, gram_out=None):
        """Tile-fused norm1 -> qkv 1x1 -> depthwise 3x3 (csrc/gdfn.hip uir_qkv_fwd).
        gram_out = (G [B][C*C], sum q^2 [B][C], sum k^2 [B][C]) (C <= 32): the attention logits' Gram and norms come out of the same
        launch as per-tile partials and are summed per sample by ONE batched reduction right behind it."""
        self._hold(X, ln[0], ln[1], Wqkv, Wdw, Ypre, Y, stats)
        a = L.QkvFwdArgs()
        a.X, a.ln_g, a.ln_b, a.Wqkv, a.Wdw, a.Ypre, a.Y = (_ptr(X), _ptr(ln[0]), _ptr(ln[1]), _ptr(Wqkv), _ptr(Wdw),
                                                            _ptr(Ypre), _ptr(Y))
        a.stats = _ptr(stats) if stats is not None else None
        a.ldx, a.ldp, a.ldq = _ld(X), (_ld(Ypre) if Ypre is not None else 0), _ld(Y)       # Ypre None: not stored (uir_dwln_bwd recomputes it)
        a.B, a.H, a.Wd = dims
        a.C = C_
        a.ln_eps = eps
        rows = dims[0] * dims[1] * dims[2]
        assert X.shape[0] == rows and C_ in (16, 32, 64, 128)
        pend = []
        if gram_out is not None:
            assert C_ <= 32
            tpi, stride = self.qkv_tiles(dims), C_ * C_ + 2 * C_
            slab = self._slab()
            assert dims[0] * tpi * stride <= slab.numel()
            a.gram_slab, a.gram_stride = _ptr(slab), stride
            self._hold(*gram_out)
            for b in range(dims[0]):
                base = _ptr(slab) + 4 * b * tpi * stride
                for off, t, n in ((0, gram_out[0], C_ * C_), (C_ * C_, gram_out[1], C_), (C_ * C_ + C_, gram_out[2], C_)):
                    pend.append((name, base + 4 * off, tpi, stride, _ptr(t) + 4 * b * n, n))
        self.keep.append(a)
        self._op(name, lib.uir_qkv_fwd, C.byref(a), nbytes=rows * 4 * (C_ + (6 if Ypre is not None else 3) * C_), flops=rows * (2 * C_ * 3 * C_ + 18 * 3 * C_))
        if pend:
            self._reduce_batch(name + ".gram.red", pend, self.lane)

    def attnv_fwd(self, name, V, WA, WB, Wdw, R, U, VF, Y, *, C_, dims):
        """Tile-fused MDTA tail (csrc/gdfn.hip uir_attnv_fwd): U = V.WA[b]^T, VF = dw3x3(U), Y = VF.WB[b]^T + R."""
        self._hold(V, WA, WB, Wdw, R, U, VF, Y)
        a = L.AttnvFwdArgs()
        a.V, a.WA, a.WB, a.Wdw, a.R, a.U, a.VF, a.Y = (_ptr(V), _ptr(WA), _ptr(WB), _ptr(Wdw), _ptr(R), _ptr(U), _ptr(VF), _ptr(Y))
        a.ldv, a.ldr, a.ldu, a.ldf, a.ldy = _ld(V), _ld(R), _ld(U), _ld(VF), _ld(Y)
        a.B, a.H, a.Wd = dims
        a.C = C_
        rows = dims[0] * dims[1] * dims[2]
        assert V.shape[0] == rows and C_ in (16, 32, 64, 128)
        self.keep.append(a)
        self._op(name, lib.uir_attnv_fwd, C.byref(a), nbytes=rows * 4 * 5 * C_, flops=rows * (4 * C_ * C_ + 18 * C_))

    @staticmethod
    def attnv_bwd_ok(C_: int, dims) -> bool:
        _B, H, W = dims
        if not (FUSE_VALUE_BWD and knobs.flag("UIR_FUSED_SPLIT") and H % 8 == 0):
            return False
        return (C_ in (16, 32) and W % 16 == 0) or (C_ == 64 and FUSE_VALUE_BWD_64 and W % 8 == 0)

    def attnv_bwd(self, name, dX1, WA, WB, Wdw, U, VF, V, dV, slab_ab, dWdw, *, C_, dims):
        """The value path's backward in one launch (csrc/attnvb.hip): dV, the per-sample dW_B | dW_A partial rows into slab_ab
        [uir_attnv_bwd_rows][2 B C^2] (the caller reduces them in front of uir_attn_bwd) and the depthwise weight gradient (bulk-lane reduction)."""
        B, H, W = dims
        rows = lib.uir_attnv_bwd_rows_c(H, W, C_)
        assert slab_ab.shape[0] == rows and slab_ab.shape[1] >= 2 * B * C_ * C_ and dWdw.numel() == C_ * 9 and dWdw.is_contiguous()
        slab_dw = self.alloc(B * rows, C_ * 9)
        self._hold(dX1, WA, WB, Wdw, U, VF, V, dV, slab_ab, dWdw)
        a = L.AttnvBwdArgs()
        a.dX1, a.WA, a.WB, a.Wdw, a.U, a.VF, a.V, a.dV = _ptr(dX1), _ptr(WA), _ptr(WB), _ptr(Wdw), _ptr(U), _ptr(VF), _ptr(V), _ptr(dV)
        a.slab_ab, a.slab_dw, a.slab_ab_stride, a.nab = _ptr(slab_ab), _ptr(slab_dw), slab_ab.stride(0), B * C_ * C_
        a.ldx, a.ldu, a.ldf, a.ldv, a.ldd = _ld(dX1), _ld(U), _ld(VF), _ld(V), _ld(dV)
        a.B, a.H, a.Wd, a.C = B, H, W, C_
        self.keep.append(a)
        M = B * H * W
        self._op(name, lib.uir_attnv_bwd, C.byref(a), nbytes=4 * M * 5 * C_, flops=M * (8 * C_ * C_ + 36 * C_), variant="uir_attnv_bwd")
        with self.bulk():
            self._reduce(name + ".dWkvdw.red", slab_dw.data_ptr(), B * rows, C_ * 9, _ptr(dWdw), C_ * 9)

    def dwln_bwd(self, name, G, XP, Wdw, W1, X, stats, gamma, beta, dRes, dX, dWdw, dW1, dgamma, dbeta, *, C_, dims, h=0, hp=0, ch=0,
                 nblocks=512, il=0):
        """Backward of LayerNorm -> 1x1 conv -> depthwise 3x3 in one pass (csrc/dwln.hip): dX = dRes + LN'(dw^T(G) . W1) and the four
        parameter gradients (per-workgroup slab rows, summed on the bulk lane).  hp > 0: GDFN column layout (x1 half at [0, h), x2
        half at [hp, hp + h)); hp == 0: `ch` consecutive channels."""
        self._hold(G, XP, Wdw, W1, X, stats, gamma, beta, dRes, dX, dWdw, dW1, dgamma, dbeta)
        B, H, W = dims
        chreal = 2 * h if hp else ch
        a = L.DwLnBwdArgs()
        a.G, a.XP, a.Wdw, a.W1, a.X, a.stats = _ptr(G), _ptr(XP), _ptr(Wdw), _ptr(W1), _ptr(X), _ptr(stats)
        a.gamma, a.beta, a.dRes, a.dX = _ptr(gamma), _ptr(beta), _ptr(dRes), _ptr(dX)
        a.ldg, a.ldp, a.ldx, a.ldr, a.ldo = _ld(G), (_ld(XP) if XP is not None else 0), _ld(X), _ld(dRes), _ld(dX)     # XP None: recomputed on chip
        a.B, a.H, a.Wd, a.C, a.h, a.hp, a.ch = B, H, W, C_, h, hp, ch
        a.il = il
        tiles = B * ((H + 7) // 8) * ((W + 15) // 16)
        a.nblocks = max(1, min(nblocks, tiles))
        stride = chreal * 9 + chreal * C_ + 2 * C_
        slab = self.alloc(a.nblocks, stride)          # private: the reductions run later, on the bulk lane
        a.slab, a.slab_stride = _ptr(slab), stride
        assert X.shape[0] == B * H * W and G.shape[0] == X.shape[0] and dWdw.numel() == chreal * 9 and dW1.numel() == chreal * C_
        self.keep.append(a)
        rows = B * H * W
        cols = 2 * hp if hp else ch
        self._op(name, lib.uir_dwln_bwd, C.byref(a), nbytes=4 * rows * ((2 if XP is not None else 1) * cols + 3 * C_),
                 flops=rows * (36 * cols + 4 * cols * C_), variant="uir_dwln_bwd_kernel<*>")
        sp = slab.data_ptr()
        with self.bulk():
            self._reduce(name + ".dWdw", sp, a.nblocks, stride, _ptr(dWdw), chreal * 9)
            self._reduce(name + ".dW1", sp + 4 * chreal * 9, a.nblocks, stride, _ptr(dW1), chreal * C_)
            off = chreal * 9 + chreal * C_
            if dbeta.data_ptr() == dgamma.data_ptr() + 4 * C_:
                self._reduce(name + ".dgb", sp + 4 * off, a.nblocks, stride, _ptr(dgamma), 2 * C_)
            else:
                self._reduce(name + ".dg", sp + 4 * off, a.nblocks, stride, _ptr(dgamma), C_)
                self._reduce(name + ".db", sp + 4 * (off + C_), a.nblocks, stride, _ptr(dbeta), C_)

    def dwconv_wgrad(self, name, dY, X, out, *, C_, creal, dims, grp=(0, 0)):
        self._hold(dY, X, out)
        a = L.DwWgradArgs()
        a.dY, a.X, a.slab = _ptr(dY), _ptr(X), _ptr(self._slab())
        a.ldy, a.ldx = _ld(dY), _ld(X)
        a.B, a.H, a.Wd = dims
        a.C, a.creal = C_, creal
        a.grp_real, a.grp_pad = grp
        hs = 16                                                            # mirror uir_dw_strip() in dwconv.hip
        while hs > 2 and dims[0] * ((dims[1] + hs - 1) // hs) * ((dims[2] + 3) // 4) * (C_ // 4) < 256 * 1024:
            hs //= 2
        items = dims[0] * ((dims[1] + hs - 1) // hs) * ((dims[2] + 3) // 4)   # (strip of hs rows, run of 4 pixels)
        ny = 256 // (C_ // 4)
        a.nblocks = max(1, min(2048, items // max(1, ny)))
        slab = self._private_slab(a.nblocks * creal * 9) if self._batching() else self._slab()
        a.slab = _ptr(slab)
        assert a.nblocks * creal * 9 <= slab.numel()
        self.keep.append(a)
        rows = dims[0] * dims[1] * dims[2]
        self._op(name, lib.uir_dwconv_wgrad, C.byref(a), nbytes=8 * rows * C_, flops=18 * rows * C_)
        self._reduce(name + ".red", _ptr(slab), a.nblocks, creal * 9, _ptr(out), creal * 9)

    def dwconv_bwd(self, name, dY, X, Wt, dX, dW, *, C_, creal, dims, grp=(0, 0), min_rows=None, act=0, dbias=None, gate_u=None,
                   gate_bias=None, dbias_gate=None):
        """Fused input-gradient + weight-gradient of a depthwise 3x3 (dY read once).  act = 1: X is the pre-activation p of a GELU
        in front of the convolution: dW uses gelu(p), dX = conv^T(dY) * gelu'(p), dbias (optional) = column sums of dX."""
        # gate_u / gate_bias (with act): dY is the gradient of gelu(gate_u + gate_bias) behind the convolution and is multiplied by that
        # GELU's derivative on load; dbias_gate = the depthwise bias gradient
        if act:
            min_rows = 0              # the activation-fused form exists as one pass only
        # fused only where bytes bind (full and half resolution); at the deep levels the chain is latency-bound and the
        # split keeps the weight-gradient half off it
        rows_total = dims[0] * dims[1] * dims[2]
        if rows_total < (FUSED_DW_MINROWS if min_rows is None else min_rows):
            # two kernels (input-gradient on the chain, weight-gradient on the bulk lane): dY is read twice.  The fused
            # kernel (2-pixel runs, window rotating by register name, ~240 VGPRs, 2 waves/SIMD) reads it once:
            # A/B on one MI355X 51.3 vs 51.9 ms/step (profiles/r01/d_ab_notes.md); the first fused version
            # (4-pixel runs, > 256 VGPRs, 1 wave/SIMD) had been slower than the split.
            with self.bulk():
                self.dwconv_wgrad(name + ".dW", dY, X, dW, C_=C_, creal=creal, dims=dims, grp=grp)
            self.dwconv(name + ".dX", dY, Wt, dX, C_=C_, dims=dims, grp=grp, flip=1)
            return
        self._hold(dY, X, Wt, dX, dW)
        a = L.DwBwdArgs()
        a.dY, a.X, a.Wt, a.dX = _ptr(dY), _ptr(X), _ptr(Wt), _ptr(dX)
        a.ldy, a.ldx, a.ldo = _ld(dY), _ld(X), _ld(dX)
        a.B, a.H, a.Wd = dims
        a.C, a.creal = C_, creal
        a.grp_real, a.grp_pad = grp
        hs = 16                                                            # mirror uir_dw_strip() in dwconv.hip
        while hs > 2 and dims[0] * ((dims[1] + hs - 1) // hs) * ((dims[2] + 3) // 4) * (C_ // 4) < 256 * 1024:
            hs //= 2
        items = dims[0] * ((dims[1] + hs - 1) // hs) * ((dims[2] + 3) // 4)
        ny = 256 // (C_ // 4)
        a.nblocks = max(1, min(512, items // max(1, ny)))
        # private partial-slab buffer: the kernel runs on the chain, its (tiny) reduction on the bulk lane, so the
        # shared chain scratch may be overwritten in between
        slab = self.alloc(a.nblocks * creal * 9)
        a.slab = _ptr(slab)
        a.act = act
        dbs = dbu = None
        if act and dbias is not None:
            dbs = self.alloc(a.nblocks * creal)
            a.db_slab = _ptr(dbs)
        if gate_u is not None:
            assert act == 1 and gate_bias is not None
            self._hold(gate_u, gate_bias)
            a.U, a.gbias, a.ldu = _ptr(gate_u), _ptr(gate_bias), _ld(gate_u)
            if dbias_gate is not None:
                dbu = self.alloc(a.nblocks * creal)
                a.dbu_slab = _ptr(dbu)
        self.keep.append(a)
        rows = dims[0] * dims[1] * dims[2]
        self._op(name, lib.uir_dwconv_bwd_fused, C.byref(a), nbytes=(12 + (4 if gate_u is not None else 0)) * rows * C_, flops=36 * rows * C_)
        with self.bulk():
            self._reduce(name + ".red", _ptr(slab), a.nblocks, creal * 9, _ptr(dW), creal * 9)
            if dbs is not None:
                self._reduce(name + ".db", _ptr(dbs), a.nblocks, creal, _ptr(dbias), creal)
            if dbu is not None:
                self._reduce(name + ".dbu", _ptr(dbu), a.nblocks, creal, _ptr(dbias_gate), creal)

    @staticmethod
    def gemm_ln_bwd_ok(M: int, C_: int, K: int) -> bool:
        """UIR_EPI_LN_BWD's contract (csrc/tsgemm.hip): the deep levels' dt1 / dt2 contractions."""
        return FUSE_LN_BWD and knobs.flag("UIR_FUSED_SPLIT") and knobs.get("UIR_MIXED") == "0" and C_ in (64, 128) and 4096 <= M <= 65536 and K >= 64 and K % 4 == 0

    def gemm_ln_bwd(self, name, Xg, W, *, X, stats, gamma, dRes, dX, dgamma, dbeta, C_, K, w_sn, w_sk, k_grp=(0, 0)):
        """dX = dRes + LayerNorm'(Xg . W^T) in ONE launch (UIR_EPI_LN_BWD) + the dgamma | dbeta partial rows reduced on the bulk lane."""
        M = Xg.shape[0]
        slab = torch.zeros((M + 63) // 64, 2 * C_, device=self.dev, dtype=torch.float32)     # rows past the launch's row tiles stay zero
        self.keep.append(slab)
        self.bytes += slab.numel() * 4
        self.gemm(name, Xg, W, dX, N=C_, K=K, w_sn=w_sn, w_sk=w_sk, k_grp=k_grp, epi=L.EPI_LN_BWD, G=X, stats=stats, ln=(gamma,), R=dRes, dw_slab=slab)
        with self.bulk():
            if dbeta.data_ptr() == dgamma.data_ptr() + 4 * C_:
                self._reduce(name + ".dgb", slab.data_ptr(), slab.shape[0], 2 * C_, _ptr(dgamma), 2 * C_)
            else:
                self._reduce(name + ".dg", slab.data_ptr(), slab.shape[0], 2 * C_, _ptr(dgamma), C_)
                self._reduce(name + ".db", slab.data_ptr() + 4 * C_, slab.shape[0], 2 * C_, _ptr(dbeta), C_)

    def ln_bwd(self, name, dT, X, stats, gamma, dRes, dX, dgamma, dbeta, *, C_):
        self._hold(dT, X, stats, gamma, dRes, dX, dgamma, dbeta)
        a = L.LnBwdArgs()
        a.dT, a.X, a.stats, a.gamma, a.dRes, a.dX = (_ptr(dT), _ptr(X), _ptr(stats), _ptr(gamma), _ptr(dRes), _ptr(dX))
        a.ldt, a.ldx, a.ldo = _ld(dT), _ld(X), _ld(dX)
        a.ldr = _ld(dRes) if dRes is not None else 0
        a.M, a.C = X.shape[0], C_
        rows_per_block = 256 // (C_ // 4)
        a.nblocks = max(1, min(512, (a.M + rows_per_block * 4 - 1) // (rows_per_block * 4)))
        # private partial slab: the (tiny) dgamma/dbeta reductions leave the chain for the bulk lane
        slab = self.alloc(a.nblocks * 2 * C_)
        a.slab = _ptr(slab)
        self.keep.append(a)
        self._op(name, lib.uir_layernorm_bwd, C.byref(a), nbytes=4 * a.M * C_ * (4 if dRes is not None else 3), flops=16 * a.M * C_)
        sp = slab.data_ptr()
        with self.bulk():
            if dbeta.data_ptr() == dgamma.data_ptr() + 4 * C_:      # weight | bias adjacent in the flat gradient bucket
                self._reduce(name + ".dgb", sp, a.nblocks, 2 * C_, _ptr(dgamma), 2 * C_)
            else:
                self._reduce(name + ".dg", sp, a.nblocks, 2 * C_, _ptr(dgamma), C_)
                self._reduce(name + ".db", sp + 4 * C_, a.nblocks, 2 * C_, _ptr(dbeta), C_)

    def colsum(self, name, src, col, C_, out):
        self._hold(src, out)
        M = src.shape[0]
        slots = max(1, 256 // (C_ // 4 if C_ % 4 == 0 else C_))      # rows per workgroup pass (float4 lanes when C_ % 4 == 0)
        nb = max(1, min(1024, M // max(1, slots * 8)))
        slab = self._private_slab(nb * C_) if self._batching() else self._slab()
        self._op(name, lib.uir_colsum_partials, _ptr(src), _ld(src), col, M, C_, _ptr(slab), nb, nbytes=4 * M * C_)
        self._reduce(name + ".red", _ptr(slab), nb, C_, _ptr(out), C_)

    def copy_cols(self, name, src, scol, dst, dcol, C_, accumulate=0):
        self._hold(src, dst)
        self._op(name, lib.uir_copy_cols, _ptr(src), _ld(src), scol, _ptr(dst), _ld(dst), dcol, src.shape[0], C_, accumulate,
                 nbytes=4 * src.shape[0] * C_ * (3 if accumulate else 2))

    def dft(self, name, in_re, in_im, out_re, out_im, *, O, N, I, C_, sign, scale, pro=0, epi=0, aux=(), aux_out=None):
        """pro / epi (L.DFT_PRO_* / L.DFT_EPI_*, two-stage kernel only): the element-wise op in front of / behind the transform rides on it
        (include/uir_kernels.h UirDft2Args); aux = its extra operands, aux_out = |out| of DFT_EPI_CABS."""
        self._hold(in_re, in_im, out_re, out_im, aux_out, *aux)
        elems = O * N * I * C_
        planes = (1 if in_im is None else 2) + (1 if out_im is None else 2) + len(aux) + (1 if aux_out is not None else 0)
        assert (pro == 0 and epi == 0) or N in (32, 64, 128), "fused element-wise forms exist on the two-stage DFT kernel only"
        if N in (32, 64, 128):
            a = L.Dft2Args()
            a.in_re, a.in_im, a.out_re, a.out_im = _ptr(in_re), _ptr(in_im), _ptr(out_re), _ptr(out_im)
            a.pro, a.epi = pro, epi
            ax = list(aux) + [None] * (3 - len(aux))
            a.aux0, a.aux1, a.aux2, a.aux_out = _ptr(ax[0]), _ptr(ax[1]), _ptr(ax[2]), _ptr(aux_out)
            ta, tb = dft2_tables(N, sign, self.dev)
            self.keep += [ta, tb]
            a.ta, a.tb = _ptr(ta), _ptr(tb)
            a.O, a.N, a.I, a.C = O, N, I, C_
            a.scale = scale
            self.keep.append(a)
            self._op(name, lib.uir_dft_axis2, C.byref(a), nbytes=4 * elems * planes,
                     flops=2 * elems * (16 + N // 16) * (1 if in_im is None else 2) * 2)
            return
        a = L.DftArgs()
        a.in_re, a.in_im, a.out_re, a.out_im = _ptr(in_re), _ptr(in_im), _ptr(out_re), _ptr(out_im)
        ct, st = twiddles(N, self.dev)
        self.keep += [ct, st]
        a.cos_t, a.sin_t = _ptr(ct), _ptr(st)
        a.O, a.N, a.I, a.C = O, N, I, C_
        a.sign, a.scale = sign, scale
        self.keep.append(a)
        elems = O * N * I * C_
        planes = (1 if in_im is None else 2) + (1 if out_im is None else 2)
        self._op(name, lib.uir_dft_axis, C.byref(a), nbytes=4 * elems * planes,
                 flops=2 * elems * N * (1 if in_im is None else 2) * (1 if out_im is None else 2))



class Plan(Recorder):
    """Launch lists + buffers of the whole network for one (B, H, W)."""

    def __init__(self, fp: FlatParams, B: int, H: int, W: int, want_backward: bool = True):
        assert H % 8 == 0 and W % 8 == 0, "H and W must be divisible by 8 (three 2x down-samplings)"
        if not plan_fits_addressing(B, H, W):
            raise RuntimeError(f"a plan for batch {B} at {H}x{W} has {B * H * W} full-resolution pixels; the kernels' 32-bit offsets "
                               f"allow {PLAN_MAX_PIXELS} per plan - use micro-batches (TrainStep does so by itself)")
        super().__init__(fp.device, fp)
        self.B, self.H, self.W = B, H, W
        self.want_backward = want_backward
        self._p_state = 1
        self._build()

    @classmethod
    def bare(cls, fp: FlatParams, scratch_floats: int = SCRATCH_FLOATS) -> "Plan":
        """A Plan with no network recorded: tests use it to record single sub-graphs."""
        self = cls.__new__(cls)
        Recorder.__init__(self, fp.device, fp, scratch_floats)
        return self

    _p_state = 0          # 0: _block records its own P launch + join; 1 / 2: first / later block of a full plan (see _block)

    # ------------------------------------------------------------ sub-graphs
    def _block(self, pfx: str, x: torch.Tensor, C_: int, heads: int, dims, y_out: "torch.Tensor | None" = None) -> Tuple[torch.Tensor, Callable]:
        """TransformerBlock (SpectralTransformer.py:142-148) on HWC rows x [M, C].  y_out: where the block's output goes (a column
        slice of a decoder's concat buffer for the last block of an encoder level: the skip `torch.cat` of :264-267 costs no copy)."""
        B, H, W = dims
        M, HW = B * H * W, H * W
        h = hidden_of(C_)
        hp = pad16(h)      # full 128-B rows: A/B 57.6 ms/step vs 59.7 with hp = pad4(h) inside a 128-B aligned row stride
        ldg = (2 * hp + 31) & ~31          # row stride of the GDFN tensors: 128-byte aligned rows, only 2*hp columns touched
        P, Gr = self.P, self.Gr
        a = f"{pfx}.attn"
        Wqkv, Wdw = P(f"{a}.qkv.weight"), P(f"{a}.qkv_conv.weight")
        Wo, Wkv, Wkvdw, Wof = (P(f"{a}.project_out.weight"), P(f"{a}.kv.weight"), P(f"{a}.kv_conv.weight"),
                               P(f"{a}.project_outf.weight"))
        Wkvf = Wkv.view(2 * C_, C_)[C_:]                 # vf half of kv (SpectralTransformer.py:107)
        Wkvdw_f = Wkvdw.view(2 * C_, 9)[C_:]
        temp = P(f"{a}.temperature")
        g1, b1 = P(f"{pfx}.norm1.weight"), P(f"{pfx}.norm1.bias")
        g2, b2 = P(f"{pfx}.norm2.weight"), P(f"{pfx}.norm2.bias")
        Win, Wffdw, Wout = P(f"{pfx}.ffn.project_in.weight"), P(f"{pfx}.ffn.conv.weight"), P(f"{pfx}.ffn.project_out.weight")

        # C <= 32 (full resolution, level 1): the GDFN tensors t1 / t2 / dg are kept in the chunk-interleaved column layout
        # (UirGdfnFwdArgs.il): their three kernels - uir_gdfn_fwd, the gate-backward GEMM, uir_dwln_bwd - walk them one
        # 16 + 16 channel chunk at a time, and interleaved that is one 128-byte line per pixel instead of two 64-byte halves
        # (tools/probes/vmem/halfline: 3.5 vs 5-6.5 TB/s).  The deep levels' separate kernels keep the split layout.
        il = 1 if (C_ <= 32 and hp <= 96 and M % 128 == 0) else 0
        stats1, stats2 = self.alloc(M, 2), self.alloc(M, 2)
        # C <= 32: the pre-depthwise tensors are not kept - uir_dwln_bwd recomputes the values it needs from the LayerNorm'd tile it holds anyway
        recompute_pre = RECOMPUTE_PRE and C_ <= 32 and knobs.flag("UIR_FUSED_SPLIT")       # (the recompute runs on split-bf16 MFMA: knob 0 keeps the tensors)
        qkv_pre, qkv = (None if recompute_pre else self.alloc(M, 3 * C_)), self.alloc(M, 3 * C_)
        gbuf = self.alloc(B * C_ * C_ + 2 * B * C_)
        Gm, sqq, sqk = gbuf[:B * C_ * C_], gbuf[B * C_ * C_:B * C_ * C_ + B * C_], gbuf[B * C_ * C_ + B * C_:]
        Pm = self.alloc(C_, C_)
        attn, WA, WB = self.alloc(B * C_ * (C_ // heads)), self.alloc(B, C_ * C_), self.alloc(B, C_ * C_)
        u, vf, x1 = self.alloc(M, C_), self.alloc(M, C_), self.alloc(M, C_)
        gpre, g = (None if recompute_pre else self.alloc(M, ldg)[:, :2 * hp]), self.alloc(M, ldg)[:, :2 * hp]
        y = self.alloc(M, C_) if y_out is None else y_out
        assert tuple(y.shape) == (M, C_)
        q_, k_, v_ = qkv[:, :C_], qkv[:, C_:2 * C_], qkv[:, 2 * C_:]

        n = pfx
        # ---- MDTA (live path of :93-100,107,110,113) ----
        # C <= 32: q k^T and the two norms come out of the qkv launch (the q and k tiles are on chip there); wider levels: a Gram launch
        fuse_qk = FUSE_QK_GRAM and C_ <= 32
        self.qkv_fwd(n + ".qkv+conv", x, (g1, b1), Wqkv, Wdw, qkv_pre, qkv, C_=C_, dims=dims, stats=stats1,
                     gram_out=(Gm, sqq, sqk) if fuse_qk else None)
        # P = W_kv[vf] @ W_o depends on weights only: side lane.  A full plan (Plan._build, _p_state >= 1) records ALL 24 of them at the very
        # start of the forward list (Recorder.prologue, spliced at index 0): the side lane runs them beside the embedding and the first
        # block's qkv kernel, and ONE join in front of the first block's attention covers every block.  (Round 4 kept the first block's in
        # place and the others behind its join: tools/chain_gaps_events.py showed the chain waiting 43 us there for the first two - a side
        # launch starts only after everything the chain had launched when it was recorded.)  A block recorded by itself (unit tests) keeps
        # its own launch + join
        hoist = self._p_state >= 1
        with (self.prologue() if hoist else self.side()):
            self.gemm(n + ".P", Wkvf, Wo, Pm, N=C_, K=C_, w_sn=1, w_sk=C_)
        fa = L.AttnFwdArgs()
        if not fuse_qk:
            self.gram(n + ".qk_gram", q_, k_, Gm, N=C_, K=C_, o_sn=C_, o_sk=1, o_bs=C_ * C_, batches=B,
                      out_elems=B * C_ * C_, extras={"sq_a": sqq, "sq_b": sqk}, fused_out=gbuf)
        if self._p_state <= 1:
            self.join(n + ".P.join")
        if self._p_state == 1:
            self._p_state = 2
        fa.G, fa.sq_q, fa.sq_k, fa.temperature, fa.P, fa.Wof = _ptr(Gm), _ptr(sqq), _ptr(sqk), _ptr(temp), _ptr(Pm), _ptr(Wof)
        fa.attn, fa.WA, fa.WB = _ptr(attn), _ptr(WA), _ptr(WB)
        fa.B, fa.C, fa.heads = B, C_, heads
        self.keep.append(fa)
        self._op(n + ".attn_fwd", lib.uir_attn_fwd, C.byref(fa))
        if C_ in _ATTNV_FUSED:
            self.attnv_fwd(n + ".WA+kv_conv+WB", v_, WA, WB, Wkvdw_f, x, u, vf, x1, C_=C_, dims=dims)
        else:
            self.gemm(n + ".WA", v_, WA, u, N=C_, K=C_, w_sn=C_, w_sk=1, w_bs=C_ * C_, batches=B)
            self.dwconv(n + ".kv_conv", u, Wkvdw_f, vf, C_=C_, dims=dims)
            self.gemm(n + ".WB", vf, WB, x1, N=C_, K=C_, w_sn=C_, w_sk=1, w_bs=C_ * C_, batches=B, R=x)
        # ---- GDFN (:127-130) ----
        # one pass over the pixels; t1/t2 written once for the backward kernels, never re-read in the forward
        self.gdfn_fwd(n + ".gdfn", x1, (g2, b2), Win, Wffdw, Wout, y, C_=C_, h=h, hp=hp, dims=dims,
                      T1=gpre, T2=g, stats=stats2, il=il)

        def backward(dy: torch.Tensor) -> torch.Tensor:
            # C <= 32 (full resolution and level 1, ~85 % of the step's bytes): depthwise-transpose -> 1x1-transpose -> LayerNorm
            # backward with all four parameter gradients in ONE kernel (uir_dwln_bwd): the depthwise input gradient never reaches HBM
            fuse_dwln = C_ <= 32
            dg = self.alloc(M, ldg)[:, :2 * hp]
            dgpre = None if fuse_dwln else self.alloc(M, ldg)[:, :2 * hp]
            dt = None if fuse_dwln else self.alloc(M, C_)
            dx1 = self.alloc(M, C_)
            dWBA = self.alloc(2 * B * C_ * C_)                      # dWB | dWA adjacent: one reduction writes both
            dWB, dWA = dWBA[:B * C_ * C_].view(B, C_ * C_), dWBA[B * C_ * C_:].view(B, C_ * C_)
            # dvf / du stay on chip (csrc/attnvb.hip): C <= 32 in place of the GEMMs with fused Grams, C = 64 in place of two Grams, two GEMMs and the depthwise backward
            value_fused = self.attnv_bwd_ok(C_, dims) and (C_ == 64 or (FUSE_VALUE_GRAMS and HW % 128 == 0))
            dvf, du = (None, None) if value_fused else (self.alloc(M, C_), self.alloc(M, C_))
            dqkv = self.alloc(M, 3 * C_)
            dqkv_pre = None if fuse_dwln else self.alloc(M, 3 * C_)
            Mqk = self.alloc(B, 4 * C_ * C_)
            dPb, dWofb, dTb, dPsum = self.alloc(B, C_ * C_), self.alloc(B, C_ * C_), self.alloc(B, heads), self.alloc(C_, C_)
            dx = self.alloc(M, C_)
            # GDFN
            # LayerNorm backward as the epilogue of the GEMM that produces its output gradient (one 32-column tile)
            fuse_dwout = C_ <= 32 and hp <= 96 and M % 128 == 0
            assert bool(il) == bool(fuse_dwln and fuse_dwout)
            if not fuse_dwout:
                with self.bulk():
                    self.gram(n + ".dWout", dy, g, Gr(f"{pfx}.ffn.project_out.weight"), N=C_, K=h, o_sn=h, o_sk=1,
                              pro_b=L.PRO_GATE, gate_off=hp)
            if fuse_dwout:
                # project_out weight gradient inside the gate-backward GEMM (its accumulator layout is the
                # B-operand layout of the pixel contraction): the bulk lane keeps only the tiny slab reduction.
                # Rows beyond the launch's gridDim.x are never written: the slab is zero-filled once, here.
                dws = torch.zeros(GEMM_SLAB_ROWS, C_ * h, device=self.dev, dtype=torch.float32)
                self.gemm(n + ".dgate+dWout", dy, Wout, dg, N=hp, K=C_, w_sn=1, w_sk=h, n_grp=(h, hp),
                          epi=L.EPI_GATE_BWD_DW, G=g, g_off=16 if il else hp, dw_slab=dws, il16=il)
                with self.bulk():
                    self._reduce(n + ".dWout.red", _ptr(dws), GEMM_SLAB_ROWS, C_ * h,
                             _ptr(Gr(f"{pfx}.ffn.project_out.weight")), C_ * h)
            else:
                self.gemm(n + ".dgate", dy, Wout, dg, N=hp, K=C_, w_sn=1, w_sk=h, n_grp=(h, hp), epi=L.EPI_GATE_BWD, G=g,
                          g_off=hp)
            if fuse_dwln:
                self.dwln_bwd(n + ".ffn_dwln_bwd", dg, gpre, Wffdw, Win, x1, stats2, g2, b2, dy, dx1, Gr(f"{pfx}.ffn.conv.weight"),
                              Gr(f"{pfx}.ffn.project_in.weight"), Gr(f"{pfx}.norm2.weight"), Gr(f"{pfx}.norm2.bias"),
                              C_=C_, dims=dims, h=h, hp=hp, il=il)
            else:
                self.dwconv_bwd(n + ".dgpre+dWffdw", dg, gpre, Wffdw, dgpre, Gr(f"{pfx}.ffn.conv.weight"), C_=2 * hp, creal=2 * h,
                                dims=dims, grp=(h, hp))
                fuse_ln = self.gemm_ln_bwd_ok(M, C_, 2 * hp) and self.gemm_ln_bwd_ok(M, C_, 3 * C_)
                if fuse_ln:
                    self.gemm_ln_bwd(n + ".dt2+ln2_bwd", dgpre, Win, X=x1, stats=stats2, gamma=g2, dRes=dy, dX=dx1, dgamma=Gr(f"{pfx}.norm2.weight"),
                                     dbeta=Gr(f"{pfx}.norm2.bias"), C_=C_, K=2 * hp, w_sn=1, w_sk=C_, k_grp=(h, hp))
                else:
                    self.gemm(n + ".dt2", dgpre, Win, dt, N=C_, K=2 * hp, w_sn=1, w_sk=C_, k_grp=(h, hp))
                # released after the GEMM that shares its dgpre operand (a Gram starting beside a persistent 2-WG/CU GEMM cost it 60 %)
                with self.bulk():
                    self.gram(n + ".dWin", dgpre, x1, Gr(f"{pfx}.ffn.project_in.weight"), N=2 * hp, K=C_, o_sn=C_, o_sk=1,
                              n_grp=(h, hp), pro_b=L.PRO_LN, ln=(g2, b2), stats=stats2)
                if not fuse_ln:
                    self.ln_bwd(n + ".ln2_bwd", dt, x1, stats2, g2, dy, dx1, Gr(f"{pfx}.norm2.weight"), Gr(f"{pfx}.norm2.bias"), C_=C_)
            # MDTA
            # dWB | dWA partials side by side in one slab: a single reduction launch (before uir_attn_bwd) instead of two
            nab = B * C_ * C_
            tn_ = (C_ + 31) // 32
            assert dWA.data_ptr() == dWB.data_ptr() + 4 * nab
            # C <= 32: each per-sample Gram rides in the epilogue of the GEMM that reads the same gradient rows (EPI_PLAIN_DW): dx1 and du are
            # read once instead of twice
            fuse_ab = FUSE_VALUE_GRAMS and C_ <= 32 and HW % 128 == 0
            fuse_vb = value_fused
            if fuse_vb:
                # the whole value path in one pass (dvf and du never reach HBM); its slab rows have the layout of the shared slab below
                nch_ab = lib.uir_attnv_bwd_rows_c(H, W, C_)
                slab_ab = self.alloc(nch_ab, 2 * nab)
                self.attnv_bwd(n + ".value_bwd", dx1, WA, WB, Wkvdw_f, u, vf, v_, dqkv[:, 2 * C_:], slab_ab,
                               Gr(f"{a}.kv_conv.weight").view(2 * C_, 9)[C_:], C_=C_, dims=dims)
            else:
                if fuse_ab:
                    nch_ab = max(1, min(_PLAIN_DW_WGS // B, HW // 128))
                    slab_ab = self.alloc(nch_ab, 2 * nab)
                    self.gemm(n + ".dvf+dWB", dx1, WB, dvf, N=C_, K=C_, w_sn=1, w_sk=C_, w_bs=C_ * C_, batches=B, epi=L.EPI_PLAIN_DW, G=vf,
                              dw_slab=slab_ab, dw_cols=C_, dw_rows=nch_ab)
                else:
                    nch_ab, _ = self._chunks(HW, 1 if tn_ == 1 else ((tn_ + 1) // 2) ** 2, B, split_nk=self.gram_split_case(HW, C_, C_))
                    slab_ab = self.alloc(nch_ab * 2 * nab)
                    got = self.gram(n + ".dWB", dx1, vf, dWB, N=C_, K=C_, o_sn=C_, o_sk=1, o_bs=C_ * C_, batches=B, out_elems=nab,
                                    shared_slab=(slab_ab, 0, 2 * nab))
                    assert got == nch_ab
                    self.gemm(n + ".dvf", dx1, WB, dvf, N=C_, K=C_, w_sn=1, w_sk=C_, w_bs=C_ * C_, batches=B)
                self.dwconv_bwd(n + ".du+dWkvdw", dvf, u, Wkvdw_f, du, Gr(f"{a}.kv_conv.weight").view(2 * C_, 9)[C_:], C_=C_, creal=C_,
                                dims=dims)
                if fuse_ab:
                    self.gemm(n + ".dv+dWA", du, WA, dqkv[:, 2 * C_:], N=C_, K=C_, w_sn=1, w_sk=C_, w_bs=C_ * C_, batches=B, epi=L.EPI_PLAIN_DW,
                              G=v_, dw_slab=slab_ab[:, nab:], dw_cols=C_, dw_rows=nch_ab)
                else:
                    got = self.gram(n + ".dWA", du, v_, dWA, N=C_, K=C_, o_sn=C_, o_sk=1, o_bs=C_ * C_, batches=B, out_elems=nab,
                                    shared_slab=(slab_ab, nab, 2 * nab))
                    assert got == nch_ab
            self._reduce(n + ".dWAB.red", _ptr(slab_ab), nch_ab, 2 * nab, _ptr(dWB), 2 * nab)
            if not fuse_ab and not fuse_vb:
                self.gemm(n + ".dv", du, WA, dqkv[:, 2 * C_:], N=C_, K=C_, w_sn=1, w_sk=C_, w_bs=C_ * C_, batches=B)
            ba = L.AttnBwdArgs()
            ba.dWA, ba.dWB, ba.G, ba.sq_q, ba.sq_k, ba.temperature = _ptr(dWA), _ptr(dWB), _ptr(Gm), _ptr(sqq), _ptr(sqk), _ptr(temp)
            ba.attn, ba.P, ba.Wof = _ptr(attn), _ptr(Pm), _ptr(Wof)
            ba.Mqk, ba.dP, ba.dWof, ba.dT = _ptr(Mqk), _ptr(dPb), _ptr(dWofb), _ptr(dTb)
            ba.B, ba.C, ba.heads = B, C_, heads
            self.keep.append(ba)
            self._op(n + ".attn_bwd", lib.uir_attn_bwd, C.byref(ba))
            with self.bulk():
                self._reduce(n + ".dWof", _ptr(dWofb), B, C_ * C_, _ptr(Gr(f"{a}.project_outf.weight")), C_ * C_)
                self._reduce(n + ".dPsum", _ptr(dPb), B, C_ * C_, _ptr(dPsum), C_ * C_, immediate=True)   # consumed by the next two launches
                self._reduce(n + ".dT", _ptr(dTb), B, heads, _ptr(Gr(f"{a}.temperature")), heads)
                # P = Wkvf @ Wo  ->  dWkvf = dP Wo^T ; dWo = Wkvf^T dP
                self.gemm(n + ".dWkvf", dPsum, Wo, Gr(f"{a}.kv.weight").view(2 * C_, C_)[C_:], N=C_, K=C_, w_sn=C_, w_sk=1)
                self.gram(n + ".dWo", Wkvf, dPsum, Gr(f"{a}.project_out.weight"), N=C_, K=C_, o_sn=C_, o_sk=1)
            self.gemm(n + ".dqk", qkv[:, :2 * C_], Mqk, dqkv[:, :2 * C_], N=2 * C_, K=2 * C_, w_sn=2 * C_, w_sk=1,
                      w_bs=4 * C_ * C_, batches=B)
            if fuse_dwln:
                self.dwln_bwd(n + ".qkv_dwln_bwd", dqkv, qkv_pre, Wdw, Wqkv, x, stats1, g1, b1, dx1, dx, Gr(f"{a}.qkv_conv.weight"),
                              Gr(f"{a}.qkv.weight"), Gr(f"{pfx}.norm1.weight"), Gr(f"{pfx}.norm1.bias"), C_=C_, dims=dims, ch=3 * C_)
                return dx
            self.dwconv_bwd(n + ".dqkv_pre+dWqkvdw", dqkv, qkv_pre, Wdw, dqkv_pre, Gr(f"{a}.qkv_conv.weight"), C_=3 * C_,
                            creal=3 * C_, dims=dims)
            if self.gemm_ln_bwd_ok(M, C_, 2 * hp) and self.gemm_ln_bwd_ok(M, C_, 3 * C_):
                self.gemm_ln_bwd(n + ".dt1+ln1_bwd", dqkv_pre, Wqkv, X=x, stats=stats1, gamma=g1, dRes=dx1, dX=dx, dgamma=Gr(f"{pfx}.norm1.weight"),
                                 dbeta=Gr(f"{pfx}.norm1.bias"), C_=C_, K=3 * C_, w_sn=1, w_sk=C_)
            else:
                self.gemm(n + ".dt1", dqkv_pre, Wqkv, dt, N=C_, K=3 * C_, w_sn=1, w_sk=C_)
                self.ln_bwd(n + ".ln1_bwd", dt, x, stats1, g1, dx1, dx, Gr(f"{pfx}.norm1.weight"), Gr(f"{pfx}.norm1.bias"), C_=C_)
            with self.bulk():      # after the GEMM that shares its dqkv_pre operand (see dWin above)
                self.gram(n + ".dWqkv", dqkv_pre, x, Gr(f"{a}.qkv.weight"), N=3 * C_, K=C_, o_sn=C_, o_sk=1,
                          pro_b=L.PRO_LN, ln=(g1, b1), stats=stats1)
            return dx

        return y, backward

    def _stack(self, pfx, x, nb, C_, heads, dims, y_out=None):
        bws = []
        for j in range(nb):
            x, bw = self._block(f"{pfx}.{j}", x, C_, heads, dims, y_out=y_out if j == nb - 1 else None)
            bws.append(bw)

        def backward(dy):
            for bw in reversed(bws):
                dy = bw(dy)
            return dy
        return x, backward

    def _conv3(self, name, x, Wname, Cin, Cout, dims, y=None, k_grp=(0, 0), n_grp=(0, 0), Kp=None, Np=None):
        """dense 3x3 conv, pad 1, no bias (native weight [Cout][Cin][3][3])."""
        M = x.shape[0]
        Kp = Kp or Cin
        Np = Np or Cout
        Wt = self.P(Wname)
        if y is None:
            y = self.alloc(M, Np)
        plain = Kp == Cin and Np == Cout and not k_grp[1] and not n_grp[1]
        # >= 32 channels on both sides (UpSample1 at every level, down3): csrc/conv3b.hip, each direction by its own contract
        split_f = plain and self.conv3_split_ok(Cout, Cin, dims)
        split_b = plain and self.want_backward_convs() and self.conv3_split_ok(Cin, Cout, dims)
        split_w = plain and self.want_backward_convs() and self.conv3_split_wgrad_ok(Cout, Cin, dims)
        pf = pb = None
        if split_f or split_b:
            pf, pb = self.conv3_wsplit(name + ".wsplit", Wt, Cout, Cin, want_bwd=split_b)
        if split_f:
            self.conv3_split(name, x, pf, y, N=Cout, K=Cin, dims=dims)
        else:
            self.gemm(name, x, Wt, y, N=Np, K=Kp, w_sn=Cin * 9, w_sk=9, w_st=1, taps=9, hw=dims[1:], k_grp=k_grp, n_grp=n_grp)

        def backward(dy, need_dx=True, dw_on_chain=False, R=None):
            """R: a gradient to ADD to the input gradient (the skip path's share: no separate accumulate pass)."""
            def dw():
                if split_w:
                    self.conv3_split_wgrad(name + ".dW", dy, x, self.Gr(Wname), N=Cout, K=Cin, dims=dims)
                else:
                    self.gram(name + ".dW", dy, x, self.Gr(Wname), N=Np, K=Kp, o_sn=Cin * 9, o_sk=9, o_st=1, taps=9, hw=dims[1:],
                              n_grp=n_grp, k_grp=k_grp)
            if dw_on_chain:      # last op of the step: the chain has nothing left to do, the bulk lane still has a queue
                dw()
            else:
                with self.bulk():
                    dw()
            if not need_dx:
                return None
            dx = self.alloc(M, Kp)
            if split_b:
                self.conv3_split(name + ".dX", dy, pb, dx, N=Cin, K=Cout, dims=dims, R=R)
            else:
                self.gemm(name + ".dX", dy, Wt, dx, N=Kp, K=Np, w_sn=9, w_sk=Cin * 9, w_st=1, taps=9, flip=1, hw=dims[1:],
                          n_grp=k_grp, k_grp=n_grp, R=R)
            return dx
        return y, backward

    def want_backward_convs(self) -> bool:
        return getattr(self, "want_backward", True)

    def _down(self, pfx, x, C_, dims):
        """DownSample (:154-158): conv3x3 C->C/2 + PixelUnshuffle(2)."""
        B, H, W = dims
        t, conv_bw = self._conv3(pfx + ".conv", x, f"{pfx}.body.0.weight", C_, C_ // 2, dims)
        y = self.alloc(B * (H // 2) * (W // 2), 2 * C_)
        self._op(pfx + ".unshuffle", lib.uir_pixel_unshuffle, _ptr(t), _ld(t), 0, _ptr(y), _ld(y), B, H // 2, W // 2, C_ // 2)

        def backward(dy, R=None):
            dt = self.alloc(B * H * W, C_ // 2)
            self._op(pfx + ".unshuffle_bwd", lib.uir_pixel_shuffle, _ptr(dy), _ld(dy), _ptr(dt), _ld(dt), 0, B, H // 2, W // 2, C_ // 2)
            return conv_bw(dt, R=R)
        return y, backward

    def _fuse(self, pfx, x, C_):
        """conv1x1+bias -> LeakyReLU(0.1) -> conv1x1+bias (:165-168)."""
        M = x.shape[0]
        W0, b0, W2, b2 = (self.P(f"{pfx}.0.weight"), self.P(f"{pfx}.0.bias"), self.P(f"{pfx}.2.weight"), self.P(f"{pfx}.2.bias"))
        hmid, out = self.alloc(M, C_), self.alloc(M, C_)
        self.gemm(pfx + ".0", x, W0, hmid, N=C_, K=C_, w_sn=C_, w_sk=1, bias=b0, act=L.ACT_LEAKY)
        self.gemm(pfx + ".2", hmid, W2, out, N=C_, K=C_, w_sn=C_, w_sk=1, bias=b2)

        def backward(dout):
            dh, dx = self.alloc(M, C_), self.alloc(M, C_)
            with self.bulk():
                self.gram(pfx + ".2.dW", dout, hmid, self.Gr(f"{pfx}.2.weight"), N=C_, K=C_, o_sn=C_, o_sk=1)
                self.colsum(pfx + ".2.db", dout, 0, C_, self.Gr(f"{pfx}.2.bias"))
            self.gemm(pfx + ".2.dX", dout, W2, dh, N=C_, K=C_, w_sn=1, w_sk=C_, epi=L.EPI_LEAKY_BWD, G=hmid)
            with self.bulk():
                self.gram(pfx + ".0.dW", dh, x, self.Gr(f"{pfx}.0.weight"), N=C_, K=C_, o_sn=C_, o_sk=1)
                self.colsum(pfx + ".0.db", dh, 0, C_, self.Gr(f"{pfx}.0.bias"))
            self.gemm(pfx + ".0.dX", dh, W0, dx, N=C_, K=C_, w_sn=1, w_sk=C_)
            return dx
        return out, backward

    def _ups(self, pfx, x, C_, dims, y_out: torch.Tensor):
        """UpS (:201-210) = reduce(cat[UpSample(x) (FFT, :174-188), UpSample1(x) (:194-198)]).
        Writes its [4M, C/2] result into y_out (a column slice of the decoder's concat buffer)."""
        B, H, W = dims
        M = B * H * W
        Ch = C_ // 2
        f = f"{pfx}.Fups"
        ar, ai = self.alloc(M, C_), self.alloc(M, C_)
        mag, pha = self.alloc(M, C_), self.alloc(M, C_)
        # the spectrum's element-wise neighbours ride on the transforms where the two-stage DFT kernel runs them (H, W in {32, 64,
        # 128}: every level of a 256 x 256 input): mag / pha, the polar form, |.| and their adjoints never make a round trip
        # through HBM as separate launches (round 4)
        fuse_ew = H in FUSED_DFT_SIZES and W in FUSED_DFT_SIZES
        cat = self.alloc(4 * M, C_)
        # the 3x3 shuffle branch is independent of the long spectral chain: side lane, joined before `reduce`.  Recorded FIRST: a side
        # launch is ordered after everything the chain has launched when it is recorded - behind the spectral chain it started only when
        # that had finished and the chain then idled for the whole branch (190-235 us per level, tools/trace_gaps.py)
        with self.side():
            s, sups_bw = self._conv3(pfx + ".Sups", x, f"{pfx}.Sups.body.0.weight", C_, 2 * C_, dims)
            self._op(pfx + ".Sups.shuffle", lib.uir_pixel_shuffle, _ptr(s), _ld(s), _ptr(cat), _ld(cat), Ch, B, H, W, Ch)
        self.dft(f + ".dftW", x, None, ar, ai, O=B * H, N=W, I=1, C_=C_, sign=-1.0, scale=1.0)
        if fuse_ew:
            self.dft(f + ".dftH+magpha", ar, ai, mag, pha, O=B, N=H, I=W, C_=C_, sign=-1.0, scale=1.0, epi=L.DFT_EPI_MAGPHA)
        else:
            fr, fi = self.alloc(M, C_), self.alloc(M, C_)
            self.dft(f + ".dftH", ar, ai, fr, fi, O=B, N=H, I=W, C_=C_, sign=-1.0, scale=1.0)
            self._op(f + ".magpha", lib.uir_magpha_fwd, _ptr(fr), _ptr(fi), _ptr(mag), _ptr(pha), B, H, W, C_)
        # the two 1x1 MLPs are independent: on few pixels (latency-bound 10-20 us launches) the phase one runs on the side lane beside the
        # amplitude one.  Recorded FIRST: a side launch is ordered behind everything the chain has launched when it is recorded
        pha_side = M <= FUPS_PHA_SIDE_MAXROWS
        if pha_side:
            with self.side():
                Pha, pha_bw = self._fuse(f + ".pha_fuse", pha, C_)
        Mag, amp_bw = self._fuse(f + ".amp_fuse", mag, C_)
        if pha_side:
            self.join(f + ".pha_fuse.join")
        else:
            Pha, pha_bw = self._fuse(f + ".pha_fuse", pha, C_)
        cr, ci, tr, ti = (self.alloc(M, C_) for _ in range(4))
        t = self.alloc(M, C_)
        if fuse_ew:
            self.dft(f + ".polar+idftH", Mag, Pha, cr, ci, O=B, N=H, I=W, C_=C_, sign=1.0, scale=1.0, pro=L.DFT_PRO_POLAR)
            self.dft(f + ".idftW+cabs", cr, ci, tr, ti, O=B * H, N=W, I=1, C_=C_, sign=1.0, scale=1.0 / (H * W), epi=L.DFT_EPI_CABS, aux_out=t)
        else:
            zr, zi = self.alloc(M, C_), self.alloc(M, C_)
            self._op(f + ".polar", lib.uir_polar_fwd, _ptr(Mag), _ptr(Pha), _ptr(zr), _ptr(zi), M * C_)
            self.dft(f + ".idftH", zr, zi, cr, ci, O=B, N=H, I=W, C_=C_, sign=1.0, scale=1.0)
            self.dft(f + ".idftW", cr, ci, tr, ti, O=B * H, N=W, I=1, C_=C_, sign=1.0, scale=1.0 / (H * W))
            self._op(f + ".cabs", lib.uir_cabs_fwd, _ptr(tr), _ptr(ti), _ptr(t), M * C_)
        Wpost, bpost = self.P(f + ".post.weight"), self.P(f + ".post.bias")
        po = self.alloc(M, Ch)
        self.gemm(f + ".post", t, Wpost, po, N=Ch, K=C_, w_sn=C_, w_sk=1, bias=bpost)
        self._op(f + ".scatter", lib.uir_scatter_even, _ptr(po), _ld(po), _ptr(bpost), _ptr(cat), _ld(cat), 0, B, H, W, Ch)
        self.join(pfx + ".join")
        Wred = self.P(f"{pfx}.reduce.weight")
        self.gemm(pfx + ".reduce", cat, Wred, y_out, N=Ch, K=C_, w_sn=C_, w_sk=1)

        def backward(dy: torch.Tensor) -> torch.Tensor:
            dcat = self.alloc(4 * M, C_)
            with self.bulk():
                self.gram(pfx + ".reduce.dW", dy, cat, self.Gr(f"{pfx}.reduce.weight"), N=Ch, K=C_, o_sn=C_, o_sk=1)
            self.gemm(pfx + ".reduce.dX", dy, Wred, dcat, N=C_, K=Ch, w_sn=1, w_sk=C_)
            # shuffle branch
            ds = self.alloc(M, 2 * C_)
            with self.side():
                self._op(pfx + ".Sups.shuffle_bwd", lib.uir_pixel_unshuffle, _ptr(dcat), _ld(dcat), Ch, _ptr(ds), _ld(ds), B, H, W, Ch)
                dx = sups_bw(ds)
            # spectral branch
            with self.bulk():
                self.colsum(f + ".post.db", dcat, 0, Ch, self.Gr(f + ".post.bias"))
            dpo = self.alloc(M, Ch)
            self._op(f + ".gather", lib.uir_gather_even, _ptr(dcat), _ld(dcat), 0, _ptr(dpo), _ld(dpo), B, H, W, Ch)
            with self.bulk():
                self.gram(f + ".post.dW", dpo, t, self.Gr(f + ".post.weight"), N=Ch, K=C_, o_sn=C_, o_sk=1)
            dtt = self.alloc(M, C_)
            self.gemm(f + ".post.dX", dpo, Wpost, dtt, N=C_, K=Ch, w_sn=1, w_sk=C_)
            gcr, gci = self.alloc(M, C_), self.alloc(M, C_)
            gMag, gPha = self.alloc(M, C_), self.alloc(M, C_)
            if fuse_ew:
                self.dft(f + ".cabs_bwd+idftW_bwd", dtt, None, gcr, gci, O=B * H, N=W, I=1, C_=C_, sign=-1.0, scale=1.0 / (H * W),
                         pro=L.DFT_PRO_CABS_BWD, aux=(tr, ti, t))
                self.dft(f + ".idftH_bwd+polar_bwd", gcr, gci, gMag, gPha, O=B, N=H, I=W, C_=C_, sign=-1.0, scale=1.0,
                         epi=L.DFT_EPI_POLAR_BWD, aux=(Mag, Pha))
            else:
                gtr, gti, gzr, gzi = (self.alloc(M, C_) for _ in range(4))
                self._op(f + ".cabs_bwd", lib.uir_cabs_bwd, _ptr(tr), _ptr(ti), _ptr(t), _ptr(dtt), _ptr(gtr), _ptr(gti), M * C_)
                self.dft(f + ".idftW_bwd", gtr, gti, gcr, gci, O=B * H, N=W, I=1, C_=C_, sign=-1.0, scale=1.0 / (H * W))
                self.dft(f + ".idftH_bwd", gcr, gci, gzr, gzi, O=B, N=H, I=W, C_=C_, sign=-1.0, scale=1.0)
                self._op(f + ".polar_bwd", lib.uir_polar_bwd, _ptr(Mag), _ptr(Pha), _ptr(gzr), _ptr(gzi), _ptr(gMag), _ptr(gPha), M * C_)
            if pha_side:
                with self.side():
                    gpha = pha_bw(gPha)
                gmag = amp_bw(gMag)
                self.join(f + ".pha_fuse.join_bwd")
            else:
                gmag = amp_bw(gMag)
                gpha = pha_bw(gPha)
            gar, gai = self.alloc(M, C_), self.alloc(M, C_)
            if fuse_ew:
                self.dft(f + ".magpha_bwd+dftH_bwd", gmag, gpha, gar, gai, O=B, N=H, I=W, C_=C_, sign=1.0, scale=1.0,
                         pro=L.DFT_PRO_MAGPHA_BWD, aux=(mag, pha))
            else:
                gfr, gfi = self.alloc(M, C_), self.alloc(M, C_)
                self._op(f + ".magpha_bwd", lib.uir_magpha_bwd, _ptr(mag), _ptr(pha), _ptr(gmag), _ptr(gpha), _ptr(gfr), _ptr(gfi), M * C_)
                self.dft(f + ".dftH_bwd", gfr, gfi, gar, gai, O=B, N=H, I=W, C_=C_, sign=1.0, scale=1.0)
            if fuse_ew:
                # the last transform adds its result to the shuffle branch's input gradient (UirDft2Args: aux0 with a real output): no dx_sum
                # pass.  The branch (side lane: unshuffle + one convolution) is long done when the spectral chain gets here
                self.join(pfx + ".join_bwd")
                self.dft(f + ".dftW_bwd+dx_sum", gar, gai, dx, None, O=B * H, N=W, I=1, C_=C_, sign=1.0, scale=1.0, aux=(dx,))
                return dx
            dxf = self.alloc(M, C_)
            self.dft(f + ".dftW_bwd", gar, gai, dxf, None, O=B * H, N=W, I=1, C_=C_, sign=1.0, scale=1.0)
            self.join(pfx + ".join_bwd")
            self.copy_cols(pfx + ".dx_sum", dxf, 0, dx, 0, C_, accumulate=1)
            return dx
        return backward

    # ----------------------------------------------------------------- model
    def _build(self):
        B, H, W = self.B, self.H, self.W
        P = self.P
        d0, d1, d2, d3 = (B, H, W), (B, H // 2, W // 2), (B, H // 4, W // 4), (B, H // 8, W // 8)
        M0, M1, M2 = B * H * W, B * H * W // 4, B * H * W // 16
        self.in_nchw = self.alloc(B, 3, H, W)
        self.out_nchw = self.alloc(B, 3, H, W)
        self.dout_nchw = self.alloc(B, 3, H, W)
        self.cur = self.fwd
        # embed_conv_rgb (3 -> 16, 3x3) on 3x3 patches [M][28]: one K = 27 GEMM forward and one <1,1> Gram for its weight
        # gradient (the last launch of the step) instead of nine 4-channel taps on 32x32 MFMA tiles
        x27 = self.alloc(M0, 28)
        Wemb = P("embed_conv_rgb.weight")
        self._op("in.im2col", lib.uir_im2col3x3_c3, _ptr(self.in_nchw), _ptr(x27), B, H, W, nbytes=4 * M0 * 31)
        self.mark("fwd.start")
        f0 = self.alloc(M0, 16)
        self.gemm("embed", x27, Wemb, f0, N=16, K=27, w_sn=27, w_sk=1)

        def embed_bw(dy, need_dx=False, dw_on_chain=True):
            self.gram("embed.dW", dy, x27, self.Gr("embed_conv_rgb.weight"), N=16, K=27, o_sn=27, o_sk=1)
        # the decoders' concat buffers (SpectralTransformer.py:264-267) exist from the start: each encoder level's last block writes its
        # output straight into the skip half, and in the backward the skip half of the concat gradient is ADDED by the down-sampler's
        # input-gradient GEMM (R operand) - the two copy passes and the accumulate pass per level of rounds 1-2 are gone
        cat1, cat2, cat3 = self.alloc(M0, 32), self.alloc(M1, 64), self.alloc(M2, 128)
        e1, enc1_bw = self._stack("encoders.0", f0, 2, 16, 1, d0, y_out=cat1[:, 16:])
        self.mark("fwd.enc0(L0,C16)")
        t1, down1_bw = self._down("down1", e1, 16, d0)
        e2, enc2_bw = self._stack("encoders.1", t1, 3, 32, 2, d1, y_out=cat2[:, 32:])
        self.mark("fwd.enc1(L1)")
        t2, down2_bw = self._down("down2", e2, 32, d1)
        e3, enc3_bw = self._stack("encoders.2", t2, 3, 64, 4, d2, y_out=cat3[:, 64:])
        self.mark("fwd.enc2(L2)")
        t3, down3_bw = self._down("down3", e3, 64, d2)
        e4, enc4_bw = self._stack("encoders.3", t3, 4, 128, 8, d3)
        self.mark("fwd.enc3(L3)")
        # decoder level 3: cat[ups_1(e4), e3] -> reduces1 (SpectralTransformer.py:263-264)
        ups1_bw = self._ups("ups_1", e4, 128, d3, cat3[:, :64])
        r3 = self.alloc(M2, 64)
        self.gemm("reduces1", cat3, P("reduces1.weight"), r3, N=64, K=128, w_sn=128, w_sk=1)
        self.mark("fwd.ups_1")
        dd3, dec3_bw = self._stack("decoders.0", r3, 3, 64, 4, d2)
        self.mark("fwd.dec0(L2)")
        ups2_bw = self._ups("ups_2", dd3, 64, d2, cat2[:, :32])
        r2 = self.alloc(M1, 32)
        self.gemm("reduces2", cat2, P("reduces2.weight"), r2, N=32, K=64, w_sn=64, w_sk=1)
        self.mark("fwd.ups_2")
        dd2, dec2_bw = self._stack("decoders.1", r2, 3, 32, 2, d1)
        self.mark("fwd.dec1(L1)")
        ups3_bw = self._ups("ups_3", dd2, 32, d1, cat1[:, :16])
        self.mark("fwd.ups_3")
        fd, dec1_bw = self._stack("decoders.2", cat1, 2, 32, 1, d0)
        self.mark("fwd.dec2(L0,C32)")
        fr, ref_bw = self._stack("refinement", fd, 4, 32, 1, d0)
        self.mark("fwd.refine(L0,C32)")
        y8, outl_bw = self._conv3("outputl", fr, "outputl.weight", 32, 8, d0)
        y4, out_bw = self._conv3("output", y8, "output.weight", 8, 3, d0, n_grp=(3, 4), Np=4)
        self._op("out.to_nchw", lib.uir_hwc4_to_nchw3, _ptr(y4), _ptr(self.out_nchw), B, H, W)
        self.mark("fwd.output")
        self.splice_prologue(0)
        if not self.want_backward:
            return
        # ------------------------------ backward ------------------------------
        self.cur = self.bwd
        dy4 = self.alloc(M0, 4)
        self._op("dout.to_hwc", lib.uir_nchw3_to_hwc4, _ptr(self.dout_nchw), _ptr(dy4), B, H, W)
        self.mark("bwd.start")
        dy8 = out_bw(dy4)
        dfr = outl_bw(dy8)
        self.mark("bwd.output")
        dfd = ref_bw(dfr)
        self.mark("bwd.refine(L0,C32)")
        dcat1 = dec1_bw(dfd)
        self.mark("bwd.dec2(L0,C32)")
        ddd2 = ups3_bw(dcat1[:, :16])
        self.mark("bwd.ups_3")
        dr2 = dec2_bw(ddd2)
        self.mark("bwd.dec1(L1)")
        dcat2 = self.alloc(M1, 64)
        with self.bulk():
            self.gram("reduces2.dW", dr2, cat2, self.Gr("reduces2.weight"), N=32, K=64, o_sn=64, o_sk=1)
        self.gemm("reduces2.dX", dr2, P("reduces2.weight"), dcat2, N=64, K=32, w_sn=1, w_sk=64)
        ddd3 = ups2_bw(dcat2[:, :32])
        self.mark("bwd.ups_2")
        dr3 = dec3_bw(ddd3)
        self.mark("bwd.dec0(L2)")
        dcat3 = self.alloc(M2, 128)
        with self.bulk():
            self.gram("reduces1.dW", dr3, cat3, self.Gr("reduces1.weight"), N=64, K=128, o_sn=128, o_sk=1)
        self.gemm("reduces1.dX", dr3, P("reduces1.weight"), dcat3, N=128, K=64, w_sn=1, w_sk=128)
        de4 = ups1_bw(dcat3[:, :64])
        self.mark("bwd.ups_1")
        dt3 = enc4_bw(de4)
        self.mark("bwd.enc3(L3)")
        de3 = down3_bw(dt3, R=dcat3[:, 64:])        # + the skip path's gradient
        dt2 = enc3_bw(de3)
        self.mark("bwd.enc2(L2)")
        de2 = down2_bw(dt2, R=dcat2[:, 32:])
        dt1 = enc2_bw(de2)
        self.mark("bwd.enc1(L1)")
        de1 = down1_bw(dt1, R=dcat1[:, 16:])
        df0 = enc1_bw(de1)
        self.mark("bwd.enc0(L0,C16)")
        embed_bw(df0, need_dx=False, dw_on_chain=True)
        self.mark("bwd.embed")
        self.cur = self.fwd

    # ------------------------------------------------------------------- run
    def forward(self, x_nchw: torch.Tensor, serial: bool = False) -> torch.Tensor:
        self.in_nchw.copy_(x_nchw)
        self.run(self.fwd, serial=serial)
        return self.out_nchw

    def backward(self, dout_nchw: torch.Tensor, serial: bool = False) -> None:
        """Fills fp.grad (every live parameter's slot is overwritten, not accumulated)."""
        self.dout_nchw.copy_(dout_nchw)
        self.run(self.bwd, serial=serial)

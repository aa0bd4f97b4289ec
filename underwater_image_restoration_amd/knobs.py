"""The ONLY environment variables the product reads (tests/test_cpu_surface.py greps the package for any other).

A clean environment reads none of them: every kernel / schedule choice of the train step is a constant in engine.py or in the
library.  Each knob below is covered by a parity test with the knob ON; `report()` (called once by _lib at import) prints the
ones that are set, so a stray variable cannot silently change what trains.
"""
from __future__ import annotations

import os
import sys

KNOBS = {
    # name: (default, what it does, the test that holds results with it on)
    "UIR_GRAPH": ("0", "1: the train step replays ONE captured hipGraph (forward + loss + backward + clip/Adam) instead of eager launch lists",
                  "tests/test_hip_model.py::test_adam_scalars_survive_an_unsynchronised_host"),
    "UIR_TILED_SPLIT": ("2", "read by libuir_hip.so: where the wide contractions of csrc/tiled.hip run on split-bf16 MFMA (3 bf16 planes per f32 "
                        "operand, 6 products, f32-accurate).  2 (default) = where it is faster: full 128 x 128 tiles, >= 256 workgroups, "
                        ">= 64 k per split; 0 = never (f32 MFMA everywhere); 1 = every tiled call; 3 = only the calls the dispatchers mark",
                        "tests/test_tiled.py::test_split_bf16_mode_is_f32_accurate"),
    "UIR_FUSED_SPLIT": ("1", "read by libuir_hip.so and engine.py: 1 (default) = the SpectralTransformer step's split-bf16 kernels (3 bf16 planes per f32 "
                        "operand, 6 products, f32-accurate: csrc/uir_common.h) - the tile-fused forwards uir_gdfn_fwd / uir_qkv_fwd / uir_attnv_fwd in "
                        "registers, the deep levels' contractions on csrc/tsgemm.hip (uir_ts_gemm declines when 0, so every plain contraction with "
                        "4 096 - 65 536 rows moves to csrc/gemm.hip; with it the LayerNorm-backward epilogue UIR_EPI_LN_BWD), the 3 x 3 convolutions of "
                        "csrc/conv3b.hip, the weight-gradient Grams of csrc/gramb.hip, the one-launch value-path backward of csrc/attnvb.hip and the recompute of "
                        "the pre-depthwise values inside uir_dwln_bwd (with 0 the C <= 32 blocks keep those tensors and the value path runs as three "
                        "launches); 0 = everything on the f32 MFMA (the arithmetic of rounds 1-3)",
                        "tests/test_hip_kernels.py::test_fused_kernels_agree_between_split_bf16_and_f32_mfma"),
    "UIR_MIXED": ("0", "read by libuir_hip.so: bf16 | fp16 = MIXED precision - every plain Linear contraction with >= 64 input channels, >= 64 outputs and "
                  ">= 256 rows (csrc/gemm.hip routes them all to csrc/tiled.hip in this mode: the Linear layers of AST / NewBigFRFNModel "
                  "from 64 -> 256 upwards, their input and weight gradients) takes its operands rounded to that "
                  "16-bit type on the bf16 / f16 matrix pipe, f32 accumulation, f32 master weights, everything else f32 (BASELINE "
                  "configs[4]'s 'mixed fp16'); results are NOT f32-accurate and no f32 number is measured with it; any other value is refused",
                  "tests/test_tiled.py::test_mixed_mode_equals_the_rounded_operand_products"),
    "UIR_DEBUG": ("0", "1: TrainStep / ModuleTrainStep refuse to start on a non-default stream while the device's default stream has work in flight (the checkable "
                  "part of INTEGRATION.md section 6: torch kernels must not run beside the step's bf16-MFMA kernels); no effect on results",
                  "tests/test_hip_model.py::test_debug_mode_refuses_a_step_beside_work_on_the_default_stream"),
    "UIR_FORCE_ALLREDUCE": ("0", "1: run the gradient all-reduce even in a world of one process (exercises the RCCL call path on a 1-GPU box)",
                            "tests/test_hip_model.py::test_rccl_allreduce_call_path_world1"),
}
# process plumbing, not knobs: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run), HIPCC / UIR_LLVM_BIN (build.py)
PLUMBING = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HIPCC", "UIR_LLVM_BIN")


# file locations (not numerics): where the user's torchvision VGG16 weights may be found for the perceptual loss (Losses/perceptual.py);
# TORCH_HOME is torch's own variable for the hub cache the reference's vgg16(pretrained=True) would have filled
PATHS = ("UIR_VGG16_WEIGHTS", "TORCH_HOME")


def path(name: str) -> "str | None":
    assert name in PATHS, f"{name} is not a documented path variable (knobs.py)"
    return os.environ.get(name) or None


def get(name: str) -> str:
    assert name in KNOBS, f"{name} is not a documented knob (knobs.py)"
    return os.environ.get(name, KNOBS[name][0])


def flag(name: str) -> bool:
    return get(name) == "1"


_reported = False


def report() -> None:
    """Print, once per process, every knob whose value differs from its default, and every other UIR_* variable (ignored)."""
    global _reported
    if _reported:
        return
    _reported = True
    for k, (default, what, _test) in KNOBS.items():
        v = os.environ.get(k)
        if v is not None and v != default:
            print(f"[uir] {k}={v} (default {default}): {what}", file=sys.stderr, flush=True)
    stray = sorted(k for k in os.environ if k.startswith("UIR_") and k not in KNOBS and k not in PLUMBING and k not in PATHS and k not in ("UIR_CPU_THREADS", "UIR_CPU_BATCH"))
    if stray:
        print(f"[uir] ignored environment variables (not knobs of this build): {', '.join(stray)}", file=sys.stderr, flush=True)

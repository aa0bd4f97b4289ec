"""Timing ablations of uir_dwln_bwd (csrc/dwln.hip): each variant is the product source with ONE phase removed by a text edit (results
are WRONG by construction - these objects never ship; they answer "what does this phase cost inside the real kernel").
    python tools/probes/dwln_ablate/make.py      (build container: writes tools/probes/dwln_ablate/libdwln_<name>.so)
    python tools/probes/dwln_ablate/run.py       (GPU box)"""
import os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
SRC = open(os.path.join(ROOT, "underwater_image_restoration_amd", "csrc", "dwln.hip")).read()
SRC = SRC.replace('#include "uir_common.h"', f'#include "{ROOT}/underwater_image_restoration_amd/csrc/uir_common.h"')
SRC = SRC.replace('#include "../../include/uir_kernels.h"', f'#include "{ROOT}/include/uir_kernels.h"')


def cut(src, start, end, repl=""):
    a = src.index(start)
    b = src.index(end, a)
    return src[:a] + repl + src[b:]


VARIANTS = {
    "base": lambda s: s,
    # (d) removed: no dW1 MFMAs
    "no_d": lambda s: cut(s, "      // ---- (d) dW1 chunk granule", "    // ---- tile epilogue", "    }\n\n"),
    # (c) removed: no dT MFMAs (the epilogue then works on zeros)
    "no_c": lambda s: cut(s, "      // ---- (c) dT += dP chunk", "      // ---- (d) dW1 chunk granule"),
    # depthwise weight-gradient terms and their reduction removed
    "no_dwdw": lambda s: cut(cut(s, "      // sum over the wave's four strips", "      lds_barrier2();\n      for (int e = t_; e < 16 * 18", "      lds_barrier2();\n#if 0\n").replace(
        "        DWA[ci * 288 + e] += (WR[e] + WR[288 + e]) + (WR[576 + e] + WR[864 + e]);\n", "        DWA[ci * 288 + e] += 0.f;\n#endif\n"),
        "#pragma unroll\n          for (int dx = 0; dx < 3; ++dx) dwa[dx] =", "          // p0 is free now").replace(
        "            dwa[3 + kx] = __builtin_elementwise_fma(xc, p1[kx], dwa[3 + kx]);\n            dwa[6 + kx] = __builtin_elementwise_fma(xc, p2[kx], dwa[6 + kx]);\n", ""),
    # (b) removed: no depthwise walk (dP in LDS stays whatever it was; the prefetched operands are kept alive through dwa)
    "no_b": lambda s: cut(s, "      // ---- (b) transposed depthwise 3x3", "      load_xp(cn, cp, sx, last ? tn : tc);",
                          "      f32x2 dwa[9];\n#pragma unroll\n      for (int t = 0; t < 9; ++t) dwa[t] = xpv[t & 7] * wf[t] * xm;\n"),
    # (b), (c), (d) removed: loads, LDS staging, barriers, reductions and the epilogue only
    "mem_only": lambda s: cut(cut(cut(s, "      // ---- (d) dW1 chunk granule", "    // ---- tile epilogue", "    }\n\n"),
                                  "      // ---- (c) dT += dP chunk", "      // ---- (d) dW1 chunk granule" if False else "    }\n\n    // ---- tile epilogue",
                                  "      acc[0] += w1v[0] + w1v[5] + w1v[10] + w1v[15];\n"),
                              "      // ---- (b) transposed depthwise 3x3", "      load_xp(cn, cp, sx, last ? tn : tc);",
                              "      f32x2 dwa[9];\n#pragma unroll\n      for (int t = 0; t < 9; ++t) dwa[t] = xpv[t & 7] * wf[t] * xm;\n"),
    # depthwise weight-gradient FMAs kept, their cross-lane / cross-wave reduction removed (the sums are folded into dbet to stay alive)
    "no_dwdw_reduce": lambda s: cut(s, "      // sum over the wave's four strips", "      // ---- (c) dT += dP chunk",
                                    "#pragma unroll\n      for (int t = 0; t < 9; ++t) dbet += dwa[t].x + dwa[t].y;\n      lds_barrier2();\n"),
    # reduction kept, the FMAs removed
    "no_dwdw_fma": lambda s: cut(s, "#pragma unroll\n          for (int dx = 0; dx < 3; ++dx) dwa[dx] =", "          // p0 is free now").replace(
        "            dwa[3 + kx] = __builtin_elementwise_fma(xc, p1[kx], dwa[3 + kx]);\n            dwa[6 + kx] = __builtin_elementwise_fma(xc, p2[kx], dwa[6 + kx]);\n", ""),
    # tile epilogue (LayerNorm backward) removed
    "no_epi": lambda s: cut(s, "    // ---- tile epilogue", "  // ---- slab row of this workgroup", "    for (int r = 0; r < 16; ++r) dgam += acc[r];\n  }\n\n"),
}
for name, fn in VARIANTS.items():
    src = fn(SRC)
    p = os.path.join(HERE, f"_dwln_{name}.hip")
    open(p, "w").write(src)
    out = os.path.join(HERE, f"libdwln_{name}.so")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", p, "-o", out,
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    spills = re.findall(r"VGPRs Spill: (\d+)", r.stderr)
    vg = re.findall(r"VGPRs: (\d+)", r.stderr)
    print(name, "ok" if r.returncode == 0 else r.stderr[-2000:], "VGPRs", vg[:1], "spills", spills[:1])
    os.remove(p)

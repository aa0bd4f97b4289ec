"""Round 3: (1) the product reads no environment variable outside knobs.py's documented, parity-tested set; (2) build.py's ISA lint
refuses the packed-f32 instruction form of the gfx950 op_sel erratum and passes every shipped object."""
import os, re, subprocess, sys, tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "underwater_image_restoration_amd")
sys.path.insert(0, ROOT)


def _sources():
    for d, _dirs, files in os.walk(PKG):
        if "_build" in d or "__pycache__" in d:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                yield os.path.join(d, f)


def test_clean_environment_reads_no_undocumented_variable():
    from underwater_image_restoration_amd import knobs
    allowed = set(knobs.KNOBS) | set(knobs.PLUMBING)
    pat = re.compile(r"""(?:environ(?:\.get|\.setdefault)?\s*[\[(]|getenv\s*\()\s*["']([A-Za-z_0-9]+)["']""")
    seen = {}
    for path in _sources():
        if path.endswith("knobs.py"):
            continue
        for name in pat.findall(open(path).read()):
            seen.setdefault(name, []).append(os.path.relpath(path, ROOT))
    extra = {k: v for k, v in seen.items() if k not in allowed}
    assert not extra, f"environment variables read outside knobs.py's documented set: {extra}"
    # every knob names the test that holds results with it switched on, and that test exists
    for k, (_default, _what, test) in knobs.KNOBS.items():
        f, fn = test.split("::")
        assert fn in open(os.path.join(ROOT, f)).read(), (k, test)


def test_knob_report_names_non_default_and_stray_variables():
    env = dict(os.environ, UIR_GRAPH="1", UIR_SKIP_SIDE="1")
    code = ("import importlib.util, sys; spec = importlib.util.spec_from_file_location('knobs', sys.argv[1]); "
            "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); m.report(); m.report()")
    out = subprocess.run([sys.executable, "-c", code, os.path.join(PKG, "knobs.py")], env=env, capture_output=True, text=True, check=True).stderr
    assert out.count("UIR_GRAPH=1") == 1 and "UIR_SKIP_SIDE" in out and "ignored" in out


def test_isa_lint_refuses_the_erratum_form_and_passes_the_library():
    from underwater_image_restoration_amd import build
    # the archived first depthwise weight-gradient kernel: hipcc's SLP vectoriser emits two v_pk_fma_f32 with op_sel:[0,1,0]
    src = os.path.join(ROOT, "tools", "probes", "pkfma", "old_dwconv_wgrad.hip")
    with tempfile.TemporaryDirectory() as td:
        obj = os.path.join(td, "old.o")
        subprocess.run([build.HIPCC, *build.FLAGS, "-c", src, "-o", obj], check=True)
        with pytest.raises(RuntimeError, match="half-swapped"):
            build.lint_isa(obj)
        obj2 = os.path.join(td, "old_noslp.o")
        subprocess.run([build.HIPCC, *build.FLAGS, "-fno-slp-vectorize", "-c", src, "-o", obj2], check=True)
        assert build.lint_isa(obj2) > 100
    # every object of the shipped library
    build.build()
    scanned = {n: build.lint_isa(os.path.join(build.OBJ, n + ".o")) for n in build.SOURCES}
    assert scanned["abi_layout"] == 0 and min(v for k, v in scanned.items() if k != "abi_layout") > 50, scanned


def test_pipelined_kernels_keep_their_loads_in_flight():
    """tools/scan_isa_waits.py on the built library (CPU only, disassembly): the kernels whose operand loads were found serialised in
    round 3 - one `global_load; s_waitcnt vmcnt(0)` round trip per MFMA - and were fixed with a sched_barrier between the load loop
    and its consumers must not fall back, and the hot kernels of the headline step that were clean stay
    clean: no chain of four or more single-load full waits."""
    import importlib.util
    import tempfile
    spec = importlib.util.spec_from_file_location("scan_isa_waits", os.path.join(ROOT, "tools", "scan_isa_waits.py"))
    scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scan)
    from underwater_image_restoration_amd import build
    build.build()                       # no-op when the objects are fresh; a fresh checkout compiles here (hipcc cross-compiles on CPU)
    watched = {"winattn.o": ("winattn_fwd_kernel",), "spectral.o": ("uir_dft2_kernel",), "dwln.o": ("uir_dwln_bwd_kernel",),
               "gdfn.o": ("uir_gdfn_fwd_kernel",), "gram.o": ("uir_gram_kernel",)}
    seen = 0
    with tempfile.TemporaryDirectory() as tmp:
        for obj, names in watched.items():
            path = os.path.join(build.OBJ, obj)
            assert os.path.exists(path), f"{path}: build the library first (__graft_entry__.build())"
            for name, body in scan.kernels(scan.disassemble(path, tmp)):
                if any(n in name for n in names):
                    seen += 1
                    _nloads, prof = scan.wait_profile(body)
                    assert scan.longest_chain(prof) < 4, (name, prof)
    assert seen >= 10


def test_trace_gaps_tool_on_a_synthetic_trace(tmp_path, capsys):
    """tools/trace_gaps.py (the chain-queue idle analysis of DESIGN.md 17.2 (6)) on a hand-made rocprofv3 kernel-trace CSV: two steps
    delimited by the optimizer kernel, the busiest queue is the chain, gaps are attributed to the kernel in front of them."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("trace_gaps", os.path.join(ROOT, "tools", "trace_gaps.py"))
    tg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tg)
    rows = ['"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name","Correlation_Id","Start_Timestamp","End_Timestamp"']
    t = 1000

    def k(name, q, dur, gap=0):
        nonlocal t
        t += gap
        rows.append(f'"KERNEL_DISPATCH","Agent 2",{q},0,1,1,1,"{name}",1,{t},{t + dur}')
        t += dur

    k("clip_adam_kernel", 1, 10)
    for _ in range(2):
        k("a_kernel", 1, 100_000)
        k("b_kernel", 1, 50_000, gap=7_000)            # 7 us behind a_kernel
        rows.append(f'"KERNEL_DISPATCH","Agent 2",3,0,1,1,1,"bulk_kernel",1,{t - 40_000},{t - 10_000}')
        k("c_kernel", 1, 20_000, gap=200_000)          # 200 us behind b_kernel
        k("clip_adam_kernel", 1, 10_000)
    p = tmp_path / "t_kernel_trace.csv"
    p.write_text("\n".join(rows) + "\n")
    sys.argv = ["trace_gaps.py", str(p), "2"]
    tg.main()
    out = capsys.readouterr().out
    assert "chain queue 1" in out and "idle 0.207 ms" in out and "busy 0.180 ms" in out
    assert "other queues busy 0.030 ms" in out
    lines = [ln for ln in out.splitlines() if ln.strip().startswith(("a_kernel", "b_kernel"))]
    assert lines[0].split()[0] == "b_kernel" and abs(float(lines[0].split()[1]) - 0.200) < 1e-9
    assert lines[1].split()[0] == "a_kernel" and abs(float(lines[1].split()[1]) - 0.007) < 1e-9

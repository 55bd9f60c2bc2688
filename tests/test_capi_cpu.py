"""CPU: the C-ABI library loads, exports every symbol include/deseq2_mi355x.h declares, and
fails loudly (no CPU fallback) when no GPU is present.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    h = open(os.path.join(ROOT, "include", "deseq2_mi355x.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(dsq_[a-z0-9_]+)\s*\(", h)))


def test_library_exports_every_declared_symbol():
    from deseq2_amd import _lib
    L = _lib.lib()
    syms = _header_symbols()
    assert len(syms) >= 16
    assert sorted(_lib.EXPORTED_SYMBOLS) == syms
    for s in syms:
        assert hasattr(L, s), "missing export " + s
    assert L.dsq_version() == 100


def test_struct_layout_matches_header():
    """ctypes mirrors of the argument blocks: sizes and the offset of every field as the C compiler lays them out"""
    import subprocess
    import tempfile
    from deseq2_amd import _lib
    names = ["DsqFitBetaArgs", "DsqFitBetaOut", "DsqFitDispArgs", "DsqFitDispOut", "DsqFitDispGridArgs",
             "DsqFitDispGridOut", "DsqPrefitArgs", "DsqPrefitOut", "DsqLogLikeArgs", "DsqInterceptArgs",
             "DsqInterceptOut", "DsqOptimArgs", "DsqOptimOut", "DsqCooksArgs", "DsqCooksOut", "DsqReplaceArgs", "DsqReplaceOut", "DsqDeseqArgs",
             "DsqDeseqOut"]
    lines = []
    for nm in names:
        t = getattr(_lib, nm)
        lines.append('printf("%%zu", sizeof(%s));' % nm)
        for f, _ in t._fields_:
            lines.append('printf(" %%zu", offsetof(%s, %s));' % (nm, "lambda" if f == "lambda_" else f))
        lines.append('printf("\\n");')
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "deseq2_mi355x.h"\nint main(void){\n%s\nreturn 0; }\n' % "\n".join(lines)
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c"); exe = os.path.join(td, "s")
        open(c, "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().strip().splitlines()
    for nm, line in zip(names, out):
        t = getattr(_lib, nm)
        got = list(map(int, line.split()))
        want = [ctypes.sizeof(t)] + [getattr(t, f).offset for f, _ in t._fields_]
        assert got == want, nm


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from deseq2_amd import _lib, native
    y = np.ones((4, 6), dtype=np.int32)
    x = np.column_stack([np.ones(6), np.repeat([0, 1], 3)]).astype(float)
    with pytest.raises(_lib.DsqError) as ei:
        native.fitBeta(y, x, np.ones((4, 6)), np.full(4, 0.1), [1, 0], np.zeros((4, 2)), [1e-6, 1e-6],
                       np.ones((4, 6)), False, 1e-8, 100, True, 0.5)
    assert ei.value.code == 3            # DSQ_ERR_DEVICE
    with pytest.raises(_lib.DsqError):
        native.fitDisp(y, x, np.ones((4, 6)), np.zeros(4), np.zeros(4), 1.0, -20.0, 1.0, 1e-6, 100, False,
                       np.ones((4, 6)), False, 1e-2, True)


def test_argument_validation():
    from deseq2_amd import native
    y = np.ones((4, 6), dtype=np.int32)
    x = np.ones((5, 2))
    with pytest.raises(ValueError):
        native.fitBeta(y, x, np.ones((4, 6)), np.full(4, 0.1), [1, 0], np.zeros((4, 2)), [1e-6, 1e-6],
                       np.ones((4, 6)), False, 1e-8, 100, True, 0.5)


def test_product_never_imports_oracle():
    """the oracle is test infrastructure: nothing under deseq2_amd/ may reference it"""
    for dp, _, files in os.walk(os.path.join(ROOT, "deseq2_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".c", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "orc_" not in txt, f


def test_r_shim_compiles_warning_free_against_the_mock_r_headers():
    """R is not installed here: the .Call shim is compiled (gcc -Wall -Wextra -Werror, no link) against the mock R API of
    tests/r_mock/ -- tests/test_r_shim.py then EXECUTES it on that mock runtime"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Wextra", "-Wno-cast-function-type", "-Werror", "-DDSQ_HAVE_R",
                        "-I", os.path.join(root, "tests", "r_mock"),
                        "-I", os.path.join(root, "include"), os.path.join(root, "deseq2_amd", "csrc", "r_shim.c")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.parametrize("case", ["standard_two_group", "standard_two_factors", "expanded_three_levels", "expanded_two_factors",
                                  "ties_and_zero_rows"])
def test_beta_prior_var_equals_the_host_mirror(case):
    """dsq_beta_prior_var (csrc/beta_prior.hip: estimateBetaPriorVar of R/core.R:1601-1689 inside the library, pure host
    code -- runs without a GPU) against core.estimateBetaPriorVar, the line-cited mirror: same stable order, same
    sequential sums, so the same bits."""
    from deseq2_amd import core, native
    rng = np.random.Generator(np.random.PCG64(sum(map(ord, case))))
    n = 3000
    if case in ("standard_two_group",):
        factors = {"condition": np.repeat([0, 1], 4)}
    elif case in ("expanded_three_levels", "ties_and_zero_rows"):
        factors = {"condition": np.repeat([0, 1, 2], 3)}
    else:
        factors = {"batch": np.tile([0, 1, 2], 4), "condition": np.repeat([0, 1, 2, 3], 3)}
    x, names = core.standard_model_matrix(factors)
    p = x.shape[1]
    mle = rng.normal(0, 1.5, (n, p)) * rng.choice([1.0, 1.0, 1.0, 8.0], (n, 1))       # some |beta| >= 10
    mle[:, 0] = rng.normal(5, 2, n)
    bm = np.exp(rng.normal(3, 2, n))
    dfit = 0.05 + 3.0 / bm
    az = np.zeros(n, bool)
    if case == "ties_and_zero_rows":
        mle[::7, 1] = 0.25                                     # exact ties of |beta| ...
        mle[1::7, 1] = -0.25                                   # ... across signs
        mle[::11, 2] = np.nan                                  # rows without a coefficient
        az[::13] = True
    expanded = case.startswith("expanded") or case == "ties_and_zero_rows"
    mmt = "expanded" if expanded else "standard"
    view = type("V", (), {"mcols": {"baseMean": bm[~az], "dispFit": dfit[~az]}})()
    ref, _ = core.estimateBetaPriorVar(view, mle[~az], names, modelMatrixType=mmt, factors=factors)
    got = native.estimateBetaPriorVarHost(
        mle, bm, dfit, az, native.coef_factor_codes(factors),
        prior_coef_factor=native.coef_factor_codes(factors, expanded=True) if expanded else None)
    assert got.shape == np.asarray(ref).shape
    assert (got == np.asarray(ref)).all(), (got, ref)
    assert got[0] == 1e6 and (got > 0).all()


def test_beta_prior_var_of_a_one_gene_object():
    """nrow(betaMatrix) == 1 (R/core.R:1647,1662): the prior variances are the squared coefficients themselves (the
    intercept still 1e6, :1669-1671) -- library and mirror alike"""
    from deseq2_amd import core, native
    factors = {"condition": np.repeat([0, 1], 4)}
    x, names = core.standard_model_matrix(factors)
    mle = np.array([[5.0, -1.75], [np.nan, np.nan]])
    bm, dfit, az = np.array([30.0, 0.0]), np.array([0.2, 0.2]), np.array([False, True])       # the second row is all zero
    view = type("V", (), {"mcols": {"baseMean": bm[~az], "dispFit": dfit[~az]}})()
    ref, _ = core.estimateBetaPriorVar(view, mle[~az], names, modelMatrixType="standard", factors=factors)
    got = native.estimateBetaPriorVarHost(mle, bm, dfit, az, native.coef_factor_codes(factors))
    assert list(ref) == [1e6, 1.75 ** 2] and list(got) == [1e6, 1.75 ** 2]


def test_shipped_library_passes_the_exec_lint():
    """profiles/r04_exec_remat.md: the toolchain can re-materialise a constant above the exec restore of a join block (it
    did, in two kernels of round 4: sqrt() returned its argument).  tools/exec_lint.py looks for that pattern in the
    disassembly of every gfx950 code object of the library that ships."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lint = os.path.join(root, "tools", "exec_lint.py")
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm-objdump")
    from deseq2_amd import _lib
    r = subprocess.run([sys.executable, lint, _lib.SO_PATH], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "0 suspicious" in r.stdout


def test_exec_lint_flags_the_round_4_pattern_and_not_a_masked_body():
    """the lint on three synthetic disassemblies: (a) round 4's defect -- constants of the next statement written at the
    fall-through of a lane-dependent loop latch, above the exec restore; (b) a constant at the target of a forward
    s_cbranch_execz, above the restore; (c) the body of a masked region entered through a forward s_cbranch_execnz (the
    weighted fit_disp<4>'s `pivot = 1.0` of a dropped column): legitimate, not reported."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("exec_lint", os.path.join(root, "tools", "exec_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)

    def dis(rows, base=0x1000):
        out = ["%016x <k>:" % base]
        for i, r in enumerate(rows):
            out.append("\t%-60s// %012X: 00000000%s" % (r[0], base + 4 * i, "" if len(r) < 2 else " <k+0x%x>" % (4 * r[1])))
        return "\n".join(out)

    a = dis([("v_add_f64 v[0:1], v[0:1], v[2:3]",), ("s_andn2_b64 exec, exec, s[2:3]",), ("s_cbranch_execnz 65533", 0),
             ("s_mov_b32 s80, 0",), ("v_mov_b32_e32 v116, 0x260",), ("s_or_b64 exec, exec, s[0:1]",), ("s_endpgm",)])
    b = dis([("s_and_saveexec_b64 s[8:9], vcc",), ("s_cbranch_execz 2", 3), ("v_add_f64 v[0:1], v[0:1], v[2:3]",),
             ("v_mov_b64_e32 v[4:5], 1.0",), ("s_or_b64 exec, exec, s[8:9]",), ("s_endpgm",)])
    c = dis([("s_and_saveexec_b64 s[8:9], s[0:1]",), ("s_cbranch_execnz 1", 3), ("s_branch 1", 4),
             ("v_mov_b64_e32 v[60:61], 1.0",), ("s_or_b64 exec, exec, s[8:9]",), ("s_endpgm",)])
    hits = [[t for _, _, t in lint.scan(lint.blocks_of_disassembly(x, "t"))] for x in (a, b, c)]
    assert hits[0] == ["v_mov_b32_e32 v116, 0x260"], hits
    assert hits[1] == ["v_mov_b64_e32 v[4:5], 1.0"], hits
    assert hits[2] == [], hits

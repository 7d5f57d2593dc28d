"""Host-side logic of the Python mirror that needs no GPU: argument validation of the keyword surface
(src/fit.jl:84-101, src/utilities.jl:902-918, 975-993), helpers, parameter packing."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import FIX


def test_standardize_and_links(mih):
    z = np.loadtxt(os.path.join(FIX, "covariates.txt"), delimiter=",")
    s = mih.standardize(z[:, 1:])
    assert abs(s.mean()) < 1e-12 and s.std(ddof=1) == pytest.approx(1.0, rel=1e-12)      # test/utilities_test.jl:143-164
    assert repr(mih.canonicallink(mih.Normal)) == "IdentityLink()"
    assert repr(mih.canonicallink(mih.Bernoulli())) == "LogitLink()"
    assert repr(mih.canonicallink(mih.Poisson)) == "LogLink()"
    assert repr(mih.canonicallink(mih.Gamma)) == "InverseLink()"
    assert mih.NegativeBinomial(3.0).r == 3.0 and mih.ProbitLink().code == 3 and mih.SqrtLink().code == 8


def test_parameter_packing_and_validation(mih):
    from mendeliht_amd import api
    keep = []
    prm = api._params(7, 2, mih.Poisson(), mih.LogLink(), 1e-5, 150, 6, 4, ":Newton", [1, 0], np.ones(10), None, 2, 10, keep,
                      init_beta=False, debias=True)
    assert (prm.k, prm.J, prm.dist, prm.link, prm.max_iter, prm.min_iter, prm.max_step, prm.est_r, prm.debias) == \
        (7, 2, 2, 2, 150, 6, 4, 2, 1)
    assert prm.tol == 1e-5 and prm.zkeep and prm.weight and not prm.group and not prm.comm
    with pytest.raises(mih.MendelIHTError):            # k (vector) without groups: utilities.jl:902-918
        api._params([2, 3], 1, mih.Normal(), mih.IdentityLink(), 1e-4, 100, 5, 3, "None", None, None, None, 1, 10, [])
    with pytest.raises(mih.MendelIHTError):
        api._params(-1, 1, mih.Normal(), mih.IdentityLink(), 1e-4, 100, 5, 3, "None", None, None, None, 1, 10, [])
    with pytest.raises(mih.MendelIHTError):            # est_r other than :MM / :Newton / :None
        api._params(3, 1, mih.NegativeBinomial(), mih.LogLink(), 1e-4, 100, 5, 3, ":Foo", None, None, None, 1, 10, [])
    with pytest.raises(mih.MendelIHTError):            # weight / zkeep / group of the wrong length
        api._params(3, 1, mih.Normal(), mih.IdentityLink(), 1e-4, 100, 5, 3, "None", None, np.ones(9), None, 1, 10, [])
    with pytest.raises(mih.MendelIHTError):
        api._params(3, 1, mih.Normal(), mih.IdentityLink(), 1e-4, 100, 5, 3, "None", [1, 1, 1], None, None, 2, 10, [])
    with pytest.raises(mih.MendelIHTError):
        api._params(3, 1, mih.Normal(), mih.IdentityLink(), 1e-4, 100, 5, 3, "None", None, None, np.ones(9, int), 1, 10, [])


def test_checky_and_bed_reader(mih, tmp_path):
    from mendeliht_amd import api
    api._checky(np.array([0.0, 1.0]), mih.Bernoulli())
    for y, d in ((np.array([0.0, 2.0]), mih.Bernoulli()), (np.array([1.5]), mih.Poisson()), (np.array([-1.0]), mih.NegativeBinomial()),
                 (np.array([0.0, 1.0]), mih.Gamma()), (np.array([-2.0]), mih.InverseGaussian())):
        with pytest.raises(mih.MendelIHTError):        # GLM.checky (fit.jl:91)
            api._checky(y, d)
    cols = mih.read_bed(os.path.join(FIX, "normal.bed"), 1000)
    assert cols.shape == (10000, 250) and cols.dtype == np.uint8
    bad = tmp_path / "x.bed"
    bad.write_bytes(b"\x00\x01\x02" + bytes(10))
    with pytest.raises(mih.MendelIHTError):
        mih.read_bed(str(bad), 8)
    with pytest.raises(mih.MendelIHTError):            # size does not match n
        mih.read_bed(os.path.join(FIX, "normal.bed"), 990)


def test_struct_layout_matches_header_order(mih):
    """Field order of the ctypes mirrors = declaration order in include/mendeliht_hip.h."""
    import re

    from conftest import ROOT
    from mendeliht_amd import api
    header = open(os.path.join(ROOT, "include", "mendeliht_hip.h")).read()

    def fields(name):
        body = header[header.index("typedef struct " + name):]
        body = body[body.index("{") + 1:body.index("} " + name)]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            m = re.search(r"\(\*\s*(\w+)\)\s*\(", decl)            # function pointer member
            if m:
                out.append(m.group(1))
                continue
            if not decl:
                continue
            for part in decl.split(","):
                w = re.findall(r"(\w+)\s*$", part.strip())
                if w:
                    out.append(w[0])
        return out

    assert fields("mih_fit_params") == [f[0] for f in api._FitParams._fields_]
    assert fields("mih_fit_result") == [f[0] for f in api._FitResult._fields_]
    assert fields("mih_comm") == [f[0] for f in api._Comm._fields_]
    assert C.sizeof(api._Comm) == 48



# ---- the Julia glue (julia/MendelIHTHip.jl) against the header: it cannot be executed here (no Julia in the image), so its
# struct mirrors and every ccall are parsed and compared with include/mendeliht_hip.h -------------------------------------
def _header_structs_and_functions():
    import re

    from conftest import ROOT
    header = open(os.path.join(ROOT, "include", "mendeliht_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)

    def cls(ctype):
        t = ctype.strip()
        if "*" in t or "(" in t:
            return "ptr"
        if re.search(r"\b(double|float)\b", t):
            return "f64" if "double" in t else "f32"
        if re.search(r"\b(int64_t|uint64_t|size_t)\b", t):
            return "i64"
        if re.search(r"\b(int32_t|uint32_t|int)\b", t):
            return "i32"
        raise AssertionError(f"unclassified C type {ctype!r}")

    structs = {}
    for name in ("mih_fit_params", "mih_fit_result", "mih_mv_result", "mih_comm"):
        body = header[header.index("typedef struct " + name):]
        body = body[body.index("{") + 1:body.index("} " + name)]
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.search(r"\(\*\s*(\w+)\)\s*\(", decl)
            if m:
                fields.append((m.group(1), "ptr"))
                continue
            base = re.match(r"((?:const\s+)?(?:struct\s+)?\w+)", decl).group(1)
            rest = decl[len(base):]
            for part in rest.split(","):
                part = part.strip()
                nm = re.findall(r"(\w+)$", part)[0]
                fields.append((nm, "ptr" if "*" in part else cls(base)))
        structs[name] = fields
    funcs = {}
    for m in re.finditer(r"^int\s+(mih_\w+)\s*\((.*?)\)\s*;", header, flags=re.M | re.S):
        args = " ".join(m.group(2).split())
        params = []
        if args and args != "void":
            depth, cur = 0, ""
            for ch in args:
                if ch == "(":
                    depth += 1
                if ch == ")":
                    depth -= 1
                if ch == "," and depth == 0:
                    params.append(cur)
                    cur = ""
                else:
                    cur += ch
            params.append(cur)
        out = []
        for prm in params:
            prm = prm.strip()
            ty = prm if ("*" in prm or "(" in prm) else re.sub(r"\s*\w+$", "", prm)      # drop the parameter name
            out.append(cls(ty))
        funcs[m.group(1)] = out
    return structs, funcs


def _julia_class(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")):
        return "ptr"
    return {"Int64": "i64", "UInt64": "i64", "Csize_t": "i64", "Int32": "i32", "Cint": "i32", "UInt32": "i32",
            "Float64": "f64", "Cdouble": "f64", "Float32": "f32", "Cfloat": "f32"}[t]


def _split_top(s):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        if ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return [q.strip() for q in parts]


def test_julia_glue_struct_mirrors_match_the_header():
    """Field order AND field types (pointer / Int64 / Int32 / Float64) of the Julia struct mirrors = include/mendeliht_hip.h."""
    import re

    from conftest import ROOT
    structs, _ = _header_structs_and_functions()
    jl = open(os.path.join(ROOT, "julia", "MendelIHTHip.jl")).read()
    jl = re.sub(r"#[^\n]*", "", jl)
    for jname, cname in (("MihFitParams", "mih_fit_params"), ("MihFitResult", "mih_fit_result"), ("MihMvResult", "mih_mv_result"),
                         ("MihComm", "mih_comm")):
        m = re.search(r"(?:mutable\s+)?struct\s+" + jname + r"\b(.*?)\nend", jl, flags=re.S)
        assert m, jname
        fields = []
        for decl in re.split(r"[;\n]", m.group(1)):
            decl = decl.strip()
            if decl:
                nm, ty = decl.split("::")
                fields.append((nm.strip(), _julia_class(ty)))
        assert fields == structs[cname], (jname, fields, structs[cname])


def test_julia_glue_ccalls_match_the_header():
    """Every ccall of the glue names a function the header declares, with the same number of arguments and the same argument
    classes (pointer / 64-bit integer / 32-bit integer / double) in the same order, and returns Cint."""
    import re

    from conftest import ROOT
    _, funcs = _header_structs_and_functions()
    jl = open(os.path.join(ROOT, "julia", "MendelIHTHip.jl")).read()
    jl = "\n".join(line.split("#")[0] if "ccall" not in line.split("#")[0] else line.split(" #")[0] for line in jl.split("\n"))
    seen = set()
    for m in re.finditer(r"ccall\(\(:(\w+),\s*LIB\),\s*(\w+),\s*\(", jl):
        name, ret = m.group(1), m.group(2)
        i, depth = m.end(), 1
        while depth:                                                  # the argument-type tuple
            depth += {"(": 1, ")": -1}.get(jl[i], 0)
            i += 1
        types = _split_top(jl[m.end():i - 1])
        j, depth = i, 1                                               # the remaining call arguments, up to the ccall's ")"
        while depth:
            depth += {"(": 1, ")": -1}.get(jl[j], 0)
            j += 1
        values = _split_top(jl[i:j - 1].lstrip(", \n"))
        assert name in funcs, f"ccall of {name}: not declared in include/mendeliht_hip.h"
        assert ret == "Cint", name
        got = [_julia_class(t) for t in types]
        assert got == funcs[name], (name, got, funcs[name])
        assert len(values) == len(types), (name, values, types)
        seen.add(name)
    # the path's entry points are all bound
    for must in ("mih_snp_create", "mih_mat_destroy", "mih_xtv", "mih_xtv_batched", "mih_fit_iht", "mih_fit_mv", "mih_cv_iht",
                 "mih_cv_mv", "mih_cv_iht_multi", "mih_cv_meanloss", "mih_fit_iht_path", "mih_abi_sizes", "mih_last_error"):
        assert must in seen, must


def test_julia_glue_has_the_file_level_wrappers():
    """iht(plinkfile, k, d; ...) / cross_validate(plinkfile, d; ...) (src/wrapper.jl:52-120, 301-349) are the REFERENCE's own
    functions: the glue restates none of their parsing or file writing -- it adds methods of MendelIHT.fit_iht /
    MendelIHT.cv_iht that are more specific than the reference's (x::SnpLinAlg{Float64} and its Transpose), which build the
    device matrix from the SAME SnpArray with the SAME center / scale / impute switches and call the GPU methods."""
    import re
    from conftest import ROOT
    jl = open(os.path.join(ROOT, "julia", "MendelIHTHip.jl")).read()
    code = re.sub(r"#[^\n]*", "", jl)
    for restated in ("writedlm", "parse_genotypes", "parse_phenotypes", "parse_covariates", "Estimated_beta", "cviht.summary"):
        assert restated not in code, restated                      # the wrappers' bodies are not restated any more
    for sig in ("function MendelIHT.fit_iht(y::AbstractVector{Float64}, x::SnpLinAlg{Float64}, z::AbstractVecOrMat{Float64}; kwargs...)",
                "function MendelIHT.fit_iht(y::AbstractMatrix{Float64}, x::Transpose{Float64, <:SnpLinAlg{Float64}}, z::AbstractVecOrMat{Float64}; kwargs...)",
                "function MendelIHT.cv_iht(y::AbstractVector{Float64}, x::SnpLinAlg{Float64}, z::AbstractVecOrMat{Float64}; kwargs...)",
                "function MendelIHT.cv_iht(y::AbstractMatrix{Float64}, x::Transpose{Float64, <:SnpLinAlg{Float64}}, z::AbstractVecOrMat{Float64}; kwargs...)"):
        assert sig in jl, sig
    assert "HipSnpLinAlg{Float64}(x.s; center=x.center, scale=x.scale, impute=x.impute, device=device)" in jl
    assert "const hip_iht = MendelIHT.iht" in jl and "const hip_cross_validate = MendelIHT.cross_validate" in jl
    # the reference's methods these specialise (src/fit.jl:60-63, src/cross_validation.jl:60-63): x::AbstractMatrix{T}
    assert "Tuple{AbstractVecOrMat{Float64}, AbstractMatrix{Float64}, AbstractVecOrMat{Float64}}" in jl


def test_julia_glue_caches_the_device_matrix_and_falls_back_to_the_cpu():
    """ (1) the device copy of a SnpArray is cached behind HipSnpLinAlg(x::SnpLinAlg) -- a
    WeakKeyDict keyed by the SnpArray, per (center, scale, impute, device) -- so a second fit_iht / cv_iht on the same SnpLinAlg
    uploads nothing, and the device memory goes when the HipSnpLinAlg's finalizer runs (mih_mat_destroy); (2) the redirect of
    the reference's fit_iht / cv_iht only takes calls the GPU methods can serve: the keyword lists it checks ARE the keyword
    lists of those methods (parsed here), anything else -- another model, an unknown keyword, memory_efficient = false -- falls
    back to the reference's own method via invoke instead of raising."""
    import re
    from conftest import ROOT
    jl = open(os.path.join(ROOT, "julia", "MendelIHTHip.jl")).read()
    code = re.sub(r"#[^\n]*", "", jl)
    assert "const DEVICE_COPIES = WeakKeyDict{SnpArray, Dict{NTuple{4, Int}, HipSnpLinAlg{Float64}}}()" in code
    assert "get!(() -> HipSnpLinAlg{Float64}(x.s; center=x.center, scale=x.scale, impute=x.impute, device=device), per, key)" in code
    assert re.search(r"finalizer\(x -> ccall\(\(:mih_mat_destroy, LIB\)", code)
    assert "forget_device_copies!()" in code and "use_device!(on::Bool=true)" in code

    def keywords_of(signature_start):
        i = code.index(signature_start)
        j = code.index(";", i)
        depth, k = 1, j
        while depth:                                     # to the parenthesis that closes the argument list
            k += 1
            depth += {"(": 1, ")": -1}.get(code[k], 0)
        body = code[j + 1:k]
        body = re.sub(r"\([^()]*\)", "", re.sub(r"\([^()]*\)", "", body))      # default values with calls in them
        body = re.sub(r"\{[^{}]*\}", "", body)
        return {m.group(1) for m in re.finditer(r"(?:^|,)\s*([A-Za-z_][A-Za-z_0-9]*)\s*(?:::|=)", body)}

    def listed(name):
        m = re.search(name + r" = \(([^)]*)\)", code)
        return {t.strip().lstrip(":") for t in m.group(1).split(",") if t.strip()}

    assert listed("const FIT_KEYWORDS") == keywords_of("function fit_iht(y::AbstractVector{Float64}, x::HipSnpLinAlg{Float64}")
    assert listed("const CV_KEYWORDS") == keywords_of("function cv_iht(y::AbstractVector{Float64}, x::HipSnpLinAlg{Float64}")
    assert listed("const MV_CV_KEYWORDS") == keywords_of("function cv_iht(Y::AbstractMatrix{Float64}, xt::Transpose{Float64, HipSnpLinAlg{Float64}}")
    # the multivariate fit takes the univariate fit's keywords
    assert listed("const FIT_KEYWORDS") == keywords_of("function fit_iht(Y::AbstractMatrix{Float64}, xt::Transpose{Float64, HipSnpLinAlg{Float64}}")
    assert code.count("|| return invoke(MendelIHT.fit_iht, REF_SIG, y, x, z; kwargs...)") == 2
    assert code.count("|| return invoke(MendelIHT.cv_iht, REF_SIG, y, x, z; kwargs...)") == 2
    assert "x.model == ADDITIVE_MODEL && all(kw -> kw in allowed, keys(kwargs)) && get(kwargs, :memory_efficient, true) === true" in code


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without a launcher starts N ranks itself (a child torch.distributed.run, before any GPU call)
    and relays their output; --dry-run prints each rank's launcher environment.  Under a launcher whose WORLD_SIZE disagrees
    with --gpus it still fails loudly."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [json.loads(q) for q in r.stdout.splitlines() if q.startswith("{")]
    assert sorted(q["rank"] for q in lines) == [0, 1]
    assert all(q["world_size"] == 2 and q["n_gpus"] == 2 and q["master_addr"] == "127.0.0.1" for q in lines)
    assert len({q["master_port"] for q in lines}) == 1
    assert sorted(q["local_rank"] for q in lines) == [0, 1]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    # a failing rank's exit code comes back through the launcher (no GPU here: every rank exits with "bench.py needs a GPU")
    import torch
    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode != 0 and "GPU(s)" in r.stderr              # fewer GPUs than ranks: said so before anything is started
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       env=dict(env, MIH_BENCH_ONE_DEVICE="1", MIH_BENCH_BACKEND="gloo"), timeout=600)
    assert r.returncode != 0                                       # the ranks start, find no GPU, and their exit code comes back


def test_wrapper_parsers_follow_the_reference(mih, tmp_path):
    """parse_phenotypes / parse_covariates of the host mirror (src/wrapper.jl:136-249) -- no GPU needed."""
    from mendeliht_amd import api
    prefix = str(tmp_path / "t")
    rows = [("1.5", "0"), ("-9", "1"), ("2.5", "1"), ("NA", "0"), ("-1.0", "1")]
    with open(prefix + ".fam", "w") as f:
        for i, (a, b) in enumerate(rows):
            f.write(f"f{i} i{i} 0 0 1 {a} {b}\n")
    y = api.parse_phenotypes(prefix, 6, mih.Normal(), 5)
    np.testing.assert_allclose(y, [1.5, 1.0, 2.5, 1.0, -1.0])                    # missing -> mean of the observed (wrapper.jl:171-192)
    assert np.array_equal(api.parse_phenotypes(prefix, 7, mih.Bernoulli(), 5), [0, 1, 1, 0, 1])
    with pytest.raises(mih.MendelIHTError):
        api.parse_phenotypes(prefix, 6, mih.Poisson(), 5)                         # no imputation for count / binary traits
    with pytest.raises(mih.MendelIHTError):
        api.parse_phenotypes(prefix, 6, mih.MvNormal(), 5)                        # one column is not multivariate
    Y = api.parse_phenotypes(prefix, [6, 7], mih.MvNormal(), 5)
    assert Y.shape == (2, 5) and Y[0, 1] == 1.0 and Y[1, 1] == 1.0
    (tmp_path / "ph.csv").write_text("1.0,2.0\n3.0,4.0\n5.0,6.0\n")
    assert api.parse_phenotypes(prefix, str(tmp_path / "ph.csv"), mih.MvNormal(), 3).shape == (2, 3)
    (tmp_path / "ph1.csv").write_text("1.0\n3.0\n5.0\n")
    assert api.parse_phenotypes(prefix, str(tmp_path / "ph1.csv"), mih.Normal(), 3).shape == (3,)
    (tmp_path / "cov.csv").write_text("1,2.0,10\n1,4.0,20\n1,6.0,60\n")
    z = api.parse_covariates(str(tmp_path / "cov.csv"))
    assert np.array_equal(z[:, 0], [1, 1, 1])                                     # the intercept is never standardized
    np.testing.assert_allclose(z[:, 1], [-1.0, 0.0, 1.0])                        # (x - mean) / sample sd (utilities.jl:494-530)
    z = api.parse_covariates(str(tmp_path / "cov.csv"), exclude_std_idx=[3])
    assert np.array_equal(z[:, 2], [10, 20, 60]) and abs(z[:, 1].mean()) < 1e-15
    z = api.parse_covariates(str(tmp_path / "cov.csv"), exclude_std_idx=[False, True, False])
    assert np.array_equal(z[:, 1], [2, 4, 6])
    import io
    buf = io.StringIO()
    api.print_cv_results(buf, np.array([3.5, 2.25]), [4, 9], 9)
    assert buf.getvalue() == "\n\nCrossvalidation Results:\n\tk\tMSE\n\t4\t3.5\n\t9\t2.25\n\nBest k = 9\n\n"   # data_structures.jl:327-335


def test_lockstep_coroutine_scheduler_on_cpu(mih, tmp_path):
    """LaneSched (csrc/fit_common.h) -- the coroutines the fits of a lock-step lane run as -- is host logic: round-robin
    interleaving at yields, no switch inside an allocation scope, every task completes even when one fails, the first error is
    the one reported, stacks reused.  tests/lanesched_harness.cpp, built with hipcc as host code against the product library."""
    import subprocess
    from conftest import ROOT
    exe = tmp_path / "lanesched_harness"
    libdir = os.path.join(ROOT, "mendeliht.jl_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-I", os.path.join(libdir, "csrc"),
                           os.path.join(ROOT, "tests", "lanesched_harness.cpp"), "-o", str(exe), "-L", libdir, "-lmendeliht_hip",
                           f"-Wl,-rpath,{libdir}", "-Wno-unused-result"], stderr=subprocess.DEVNULL)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "lanesched: OK" in r.stdout, r.stdout + r.stderr


def test_callback_signatures_of_the_bindings_match_the_header():
    """The function-pointer fields of the header (progress, choose; the exchange callbacks of mih_comm) against what the bindings
    hand in: the glue's @cfunction tuples and the mirror's CFUNCTYPE declarations -- return class and every argument class."""
    import ctypes as C
    import re

    from conftest import ROOT
    header = open(os.path.join(ROOT, "include", "mendeliht_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)

    def cls(t):
        t = t.strip()
        if "*" in t:
            return "ptr"
        if re.search(r"\b(double)\b", t):
            return "f64"
        if re.search(r"\b(int64_t|uint64_t|size_t)\b", t):
            return "i64"
        if re.search(r"\b(int32_t|uint32_t|int)\b", t):
            return "i32"
        if t == "void":
            return "void"
        raise AssertionError(t)

    sigs = {}
    for m in re.finditer(r"(\w+)\s*\(\*\s*(\w+)\)\s*\((.*?)\)\s*;", header, flags=re.S):
        args = [" ".join(a.split()) for a in m.group(3).split(",")]
        sigs[m.group(2)] = (cls(m.group(1)), [cls(a if "*" in a else re.sub(r"\s*\w+$", "", a)) for a in args])
    assert {"progress", "choose", "allreduce", "allgather"} <= set(sigs), sigs.keys()
    assert sigs["choose"] == ("i32", ["ptr", "i32", "ptr", "i64", "i64", "ptr"])
    # Julia: @cfunction(name, Ret, (Args...))
    jl = re.sub(r"#[^\n]*", "", open(os.path.join(ROOT, "julia", "MendelIHTHip.jl")).read())
    jcls = {"Cint": "i32", "Int32": "i32", "Int64": "i64", "Cvoid": "void", "Float64": "f64", "Cdouble": "f64"}
    found = {}
    for m in re.finditer(r"@cfunction\(\s*(\w+)\s*,\s*(\w+)\s*,\s*\((.*?)\)\s*\)", jl, flags=re.S):
        args = [a.strip() for a in _split_top(m.group(3))]
        found[m.group(1)] = (jcls[m.group(2)], ["ptr" if a.startswith(("Ptr{", "Ref{")) else jcls[a] for a in args])
    assert found.get("choose_cb") == sigs["choose"], found
    # ... and the Julia function behind it takes exactly those argument types and returns Cint
    m = re.search(r"function choose_cb\((.*?)\)::(\w+)", jl)
    assert m and m.group(2) == "Cint"
    jargs = [a.split("::")[1].strip() for a in _split_top(m.group(1))]
    assert ["ptr" if a.startswith("Ptr{") else jcls[a] for a in jargs] == sigs["choose"][1]
    # ctypes mirror
    from mendeliht_amd import api

    def ccls(t):
        if t is None:
            return "void"
        if t in (C.c_void_p,) or hasattr(t, "contents") or (isinstance(t, type) and issubclass(t, C._Pointer)):
            return "ptr"
        return {C.c_int: "i32", C.c_int32: "i32", C.c_int64: "i64", C.c_double: "f64"}[t]
    for name, proto in (("choose", api._CHOOSE), ("progress", api._PROGRESS), ("allreduce", api._ALLREDUCE), ("allgather", api._ALLGATHER)):
        got = (ccls(proto._restype_), [ccls(a) for a in proto._argtypes_])
        assert got == sigs[name], (name, got, sigs[name])


def _julia_block_balance(src):
    """(openers, ends, final bracket depth) of Julia source: strings, chars and comments skipped; `for` / `if` / `end` / `begin`
    inside brackets are comprehension / indexing tokens, not blocks."""
    i, n = 0, len(src)
    depth_br = []          # stack of '(' '[' '{'
    opens = ends = 0
    stack = []
    import re
    toks = re.compile(r"[A-Za-z_@][A-Za-z_0-9!]*")
    line = 1
    while i < n:
        ch = src[i]
        if ch == "\n":
            line += 1; i += 1; continue
        if ch == "#":
            if src.startswith("#=", i):
                j = src.index("=#", i) + 2
                line += src.count("\n", i, j); i = j
            else:
                while i < n and src[i] != "\n": i += 1
            continue
        if ch == '"':
            if src.startswith('"""', i):
                j = src.index('"""', i + 3) + 3
                line += src.count("\n", i, j); i = j; continue
            i += 1
            while src[i] != '"':
                if src[i] == "\\": i += 1
                elif src[i] == "$" and src[i+1] == "(":       # interpolation: skip to the matching paren
                    d = 0
                    while True:
                        if src[i] == "(": d += 1
                        if src[i] == ")":
                            d -= 1
                            if d == 0: break
                        i += 1
                i += 1
            i += 1; continue
        if ch == "'" and (i == 0 or not (src[i-1].isalnum() or src[i-1] in ")]}_'")):     # a character literal, not a transpose
            j = i + 1
            if src[j] == "\\": j += 1
            j += 1
            if j < n and src[j] == "'": i = j + 1; continue
        if ch in "([{":
            depth_br.append(ch); i += 1; continue
        if ch in ")]}":
            depth_br.pop(); i += 1; continue
        m = toks.match(src, i)
        if m:
            w = m.group(0)
            prev = src[i-1] if i else "\n"
            i = m.end()
            if prev in ".:" and w in ("end", "begin", "if", "for"):       # a.end / :end symbols
                continue
            in_idx = "[" in depth_br
            in_par = bool(depth_br)
            if w in ("function", "struct", "while", "try", "let", "module", "quote", "macro", "do", "baremodule"):
                opens += 1; stack.append((w, line))
            elif w == "begin" and not in_idx:
                opens += 1; stack.append((w, line))
            elif w in ("if", "for") and not in_par:
                opens += 1; stack.append((w, line))
            elif w == "end" and not in_idx:
                ends += 1
                if stack: stack.pop()
            continue
        i += 1
    return opens, ends, len(depth_br), stack


def test_julia_glue_blocks_and_brackets_balance():
    """No Julia toolchain in the image: at least every function / struct / if / for / do / begin / try of the glue has its `end`
    and every bracket closes (strings, comments, comprehension `for` / `if` and indexing `end` are not counted) -- and the
    checker notices when one is taken away."""
    from conftest import ROOT
    src = open(os.path.join(ROOT, "julia", "MendelIHTHip.jl")).read()
    opens, ends, brackets, unclosed = _julia_block_balance(src)
    assert opens == ends and opens > 40 and brackets == 0 and not unclosed, (opens, ends, brackets, unclosed)
    cut = src.rindex("\nend", 0, src.index("function make_params"))          # drop the `end` of the function before make_params
    o2, e2, _, left = _julia_block_balance(src[:cut] + src[cut + 4:])
    assert o2 == e2 + 1 and left


def test_sharded_fit_hands_every_shard_its_own_weights_and_group_labels(mih, monkeypatch):
    """dist.fit_iht_sharded (host logic, no device): `weight` and `group` are given for ALL p_global columns -- the reference's
    keywords -- and a shard's fit must get the entries of its own columns only (mih_fit_params: h, weight, group cover the LOCAL
    columns; J, k and a vector k stay the whole matrix's); the full-length beta is assembled from the shard's block."""
    import types
    import mendeliht_amd                        # noqa: F401  (the alias module registers the package)
    from mendeliht_amd import api as api_mod, dist as D
    seen = {}

    def fake_fit(y, x, z, weight=None, comm=None, **kw):
        seen.update(weight=weight, comm=comm, kw=kw)
        beta = np.zeros(x.p)
        beta[[0, x.p - 1]] = [1.5, -2.5]
        return types.SimpleNamespace(beta=beta)
    monkeypatch.setattr(api_mod, "fit_iht", fake_fit)
    p_global, lo, cnt = 50, 20, 17
    shard = types.SimpleNamespace(p=cnt, device=0)
    w = np.arange(p_global, dtype=float) + 1.0
    g = np.repeat(np.arange(1, 11), 5)
    kvec = np.arange(1, 11)
    res = D.fit_iht_sharded(np.zeros(4), shard, None, col_offset=lo, p_global=p_global, weight=w, group=g, J=3, k=kvec, debias=True)
    assert np.array_equal(seen["weight"], w[lo:lo + cnt])
    assert np.array_equal(seen["kw"]["group"], g[lo:lo + cnt]) and seen["kw"]["J"] == 3 and np.array_equal(seen["kw"]["k"], kvec) and seen["kw"]["debias"] is True
    assert seen["comm"].world == 1 and seen["comm"]._c.col_offset == lo and seen["comm"]._c.p_global == p_global
    want = np.zeros(p_global); want[lo] = 1.5; want[lo + cnt - 1] = -2.5
    assert np.array_equal(res.beta, want)

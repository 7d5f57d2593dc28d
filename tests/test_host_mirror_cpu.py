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


def test_bench_refuses_gpus_without_a_launcher():
    """`bench.py --gpus N` outside torch.distributed.run used to measure ONE GPU silently: it must exit non-zero,
    before touching any GPU, and print the launcher command."""
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node 4" in r.stderr

"""The C-ABI library loads and exports every symbol of include/mendeliht_hip.h (no GPU needed)."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol(mih):
    import __graft_entry__ as g
    g.build()
    header = open(os.path.join(ROOT, "include", "mendeliht_hip.h")).read()
    declared = set(re.findall(r"^int\s+(mih_\w+)\s*\(", header, flags=re.M))
    from mendeliht_amd import api
    assert declared == set(api.exported_symbols())
    L = mih.lib()
    for s in sorted(declared):
        assert getattr(L, s) is not None


def test_probe_entry_points_live_in_the_measurement_build_only(mih):
    """Kernel-shape knobs and timing probes are not part of the product: libmendeliht_hip.so exports exactly the header's
    functions, the measurement build (-DMIH_PROBES) those plus include/mendeliht_hip_probes.h; the product has no mutable
    process-wide selectors (mih_set_*) and reads none of the MENDELIHT_* A/B switches."""
    import ctypes as C
    import subprocess
    from mendeliht_amd import api
    probes_h = open(os.path.join(ROOT, "include", "mendeliht_hip_probes.h")).read()
    probe_syms = set(re.findall(r"^int\s+(mih_\w+)\s*\(", probes_h, flags=re.M))
    assert probe_syms == set(api.probe_symbols())

    def exported(path):
        out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
        return {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("mih_")}
    prod, meas = exported(mih.library_path()), exported(mih.probes_library_path())
    assert prod == set(api.exported_symbols())
    assert meas == prod | probe_syms
    assert not any(s.startswith("mih_set_") or s.startswith("mih_probe_") for s in prod)
    blob = open(mih.library_path(), "rb").read()
    for switch in (b"MENDELIHT_XTV_MAX_OPS", b"MENDELIHT_XTV_SLICES", b"MENDELIHT_XTV_NO_HALF", b"MENDELIHT_CV_LANES", b"MENDELIHT_CV_NO_MERGE",
                   b"MENDELIHT_CV_NO_INIT_SHARE", b"MENDELIHT_CV_TRACE", b"MENDELIHT_NO_SPIN", b"MENDELIHT_NO_ARENA", b"MENDELIHT_TOPK_RADIX8",
                   b"MENDELIHT_XV_MULTI", b"MENDELIHT_CV_ASSIGN", b"MENDELIHT_CV_NO_COOP", b"MENDELIHT_COOP_SPIN_US", b"MENDELIHT_INGEST_TRACE",
                   b"MENDELIHT_NO_RESERVE", b"MENDELIHT_INGEST_THREADS"):
        assert switch not in blob, switch
        assert switch in open(mih.probes_library_path(), "rb").read(), switch
    #  the product reads ONE environment variable: where to find librccl.  The reserve of device memory is an
    # argument now (mih_mat_reserve), the number of upload workers is fixed by the library
    assert set(re.findall(rb"MENDELIHT_[A-Z0-9_]+", blob)) == {b"MENDELIHT_RCCL_LIB"}
    assert b"MENDELIHT_RESERVE_MIN_BYTES" not in open(mih.probes_library_path(), "rb").read()
    assert C.sizeof(api._PassRecord) == 80


def test_struct_mirrors_match_the_library(mih):
    """ctypes mirrors of the C structs have the library's sizes (catches a field added on one side only)."""
    import ctypes as C

    from mendeliht_amd import api
    sizes = (C.c_int64 * 4)()
    assert mih.lib().mih_abi_sizes(sizes, 4) == 0
    assert list(sizes) == [C.sizeof(api._FitParams), C.sizeof(api._FitResult), C.sizeof(api._MvResult), C.sizeof(api._Comm)]


def test_no_device_fails_loudly(mih):
    if mih.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(mih.MendelIHTError):
        mih.project_k(np.arange(10.0), 3)
    with pytest.raises(mih.MendelIHTError):
        mih.SnpLinAlg(np.zeros((4, 3), dtype=np.uint8), n=10, center=True, scale=True)


def test_product_does_not_touch_the_oracle():
    """The oracle is test infrastructure: nothing under the product package may reference it."""
    pkg = os.path.join(ROOT, "mendeliht.jl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".inc")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="replace").read().lower()
                assert "oracle" not in txt and "libiht_oracle" not in txt, os.path.join(dp, f)


def test_host_argument_checks(mih):
    """Argument errors raised on the host side mirror fit.jl:87-101 / utilities.jl:554."""
    with pytest.raises(mih.MendelIHTError):
        mih.project_k(np.arange(5.0), -1)
    with pytest.raises(ValueError):
        mih.fit_iht(np.zeros(4), np.zeros((4, 3), dtype=np.uint8))      # raw SnpArray rejected


def test_fold_sharding_covers_every_combination(mih):
    from mendeliht_amd import dist as D
    q, npath = 5, 20
    for world in (1, 2, 3, 8):
        seen = sorted(i for r in range(world) for i in D.shard_combinations(q, npath, r, world))
        assert seen == list(range(q * npath))
    # the rule (mih_cv_assignment): round-robin over the combinations sorted by model size -- every rank gets a stratified
    # sample of the model sizes, 12 or 13 of the 100 fits at world = 8, and a single rank gets everything
    rank_of = mih.cv_assignment(range(1, 21), 5, 8)
    assert rank_of.shape == (5, 20)
    for r in range(8):
        fold, ik = np.nonzero(rank_of == r)
        assert fold.size in (12, 13)
        ks = np.sort(ik + 1)
        assert ks[0] <= 2 and ks[-1] >= 19 and np.max(np.diff(ks)) <= 3          # no rank collects only large or only small models
        assert len(set(fold)) == 5                                              # and every rank works on every fold
    assert np.all(mih.cv_assignment([5, 1, 9], 4, 1) == 0)
    # a path that is not sorted: the rule sorts by the model size, not by the position in the path
    a = mih.cv_assignment([3, 50, 7], 2, 2)
    assert sorted(a[:, 1]) == [0, 1] and sorted(a[:, 2]) == [0, 1] and sorted(a[:, 0]) == [0, 1]


def _build_harness(tmp_path):
    """tests/abi_harness.c compiled as plain C against include/mendeliht_hip.h (the symbol list is generated from the
    header's declarations, so the harness resolves exactly what the header promises)."""
    import subprocess
    header = open(os.path.join(ROOT, "include", "mendeliht_hip.h")).read()
    declared = sorted(set(re.findall(r"^int\s+(mih_\w+)\s*\(", header, flags=re.M)))
    (tmp_path / "abi_symbols.inc").write_text(",\n".join(f'"{s}"' for s in declared) + "\n")
    exe = tmp_path / "abi_harness"
    # the header alone must be strictly conforming C99; the harness itself converts dlsym's void* to function
    # pointers (POSIX, not ISO C), so it is built without -pedantic
    (tmp_path / "hdr_only.c").write_text('#include "mendeliht_hip.h"\nint main(void) { return (int)sizeof(mih_fit_params) == 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                           str(tmp_path / "hdr_only.c")])
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", str(tmp_path),
                           os.path.join(ROOT, "tests", "abi_harness.c"), "-o", str(exe), "-ldl", "-lm"])
    return exe


def test_c_harness_compiles_and_resolves_every_symbol(mih, tmp_path):
    """The header is consumable C (gcc -std=c99 -pedantic -Werror) and the library, dlopen'ed from plain C, exports every
    declared entry point with struct sizes equal to the C compiler's."""
    import subprocess
    exe = _build_harness(tmp_path)
    r = subprocess.run([str(exe), mih.library_path(), os.path.join(ROOT, "tests", "fixtures"), "symbols-only"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "entry points resolved" in r.stdout


def test_stand_in_rccl_matches_the_real_prototypes_and_codes(tmp_path):
    """tests/fake_rccl.c (the test-only stand-in the one-GPU box loads through MENDELIHT_RCCL_LIB so that csrc/comm.hip runs with
    more than one rank) includes the REAL <rccl/rccl.h>: it only compiles if its six definitions match the real prototypes, and
    the codes comm.hip declares by hand must be the header's (the stand-in rejects anything else at run time).  No GPU call."""
    import subprocess
    src = os.path.join(ROOT, "tests", "fake_rccl.c")
    lib = tmp_path / "libfake_rccl.so"
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src,
                           "-o", str(lib), "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread", "-ldl"])
    # ... and the code object of its one kernel (the device-side all-reduce) cross-compiles for gfx950; its argument block is the C file's
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--genco", "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "tests", "fake_rccl_kernels.hip"),
                           "-o", str(tmp_path / "fake_rccl_kernels.hsaco")])
    ksrc = open(os.path.join(ROOT, "tests", "fake_rccl_kernels.hip")).read()
    assert "struct FakePeers { const double *p[16]; int n; };" in ksrc and "typedef struct { const double *p[16]; int n; } fake_peers;" in open(src).read()
    out = subprocess.check_output(["nm", "-D", "--defined-only", str(lib)], text=True)
    have = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert {"ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllReduce", "ncclAllGather", "ncclGetErrorString"} <= have
    comm = open(os.path.join(ROOT, "mendeliht.jl_amd", "csrc", "comm.hip")).read()
    hdr = open("/opt/rocm/include/rccl/rccl.h").read()
    mine = dict(re.findall(r"(kNccl\w+)\s*=\s*(\d+)", comm))
    for ours, theirs in (("kNcclFloat64", "ncclFloat64"), ("kNcclSum", "ncclSum"), ("kNcclMax", "ncclMax"), ("kNcclSuccess", "ncclSuccess")):
        real = re.search(rf"\b{theirs}\s*=\s*(\d+)", hdr)
        assert real and int(mine[ours]) == int(real.group(1)), (ours, theirs)
    assert "NCCL_UNIQUE_ID_BYTES 128" in hdr


def test_every_python_file_of_the_tree_compiles():
    """tools/ holds seventy measurement scripts that only ever run on the GPU box: a syntax error in one of them would surface in
    the middle of a profiling call.  Byte-compile every .py of the tree (nothing is imported or executed)."""
    import glob
    import py_compile
    import tempfile
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        for pat in ("*.py", "tools/*.py", "tests/*.py", "tests/golden/*.py", "oracle/*.py", "mendeliht.jl_amd/*.py"):
            for f in sorted(glob.glob(os.path.join(ROOT, pat))):
                try:
                    py_compile.compile(f, cfile=os.path.join(tmp, "x.pyc"), doraise=True)
                except py_compile.PyCompileError as e:
                    bad.append((os.path.relpath(f, ROOT), str(e).splitlines()[-1]))
    assert not bad, bad

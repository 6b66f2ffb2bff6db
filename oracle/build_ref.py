"""Builds oracle/_ref: the reference's own SpMV templates, compiled in place.

TEST INFRASTRUCTURE ONLY (see csx_oracle.c).  Usable only where the reference
tree is mounted (default /root/reference); on the GPU box only the prebuilt
.so files under oracle/_ref/ are used.

The reference never compiles src/templates/*.c ahead of time: at tune time
CsxJit reads the templates, substitutes ${...} hooks for the patterns present
in a partition and hands the text to clang (include/sparsex/internals/
CsxJit.hpp:281-357, 359-415, 419-673).  This script performs that same text
substitution for a given id_map and compiles the result with gcc.  No
reference source is copied into the repository: template and header text is
read from the reference tree at build time and the outputs (generated C and
.so) stay under oracle/_ref/, which is git-ignored.  The only derived header
is sparsex/config.h, produced from the reference's config.h.in by the same
two-macro substitution its configure script performs (int / double).
"""
import ctypes as C
import hashlib
import os
import re
import subprocess

REF_ROOT = os.environ.get("SPX_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")

T_NAMES = {1: "horiz", 2: "vert", 3: "diag", 4: "rdiag"}


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "src", "templates"))


def _read(rel):
    with open(os.path.join(REF_ROOT, rel)) as f:
        return f.read()


def _gen_headers():
    gen = os.path.join(REF_DIR, "gen", "sparsex")
    os.makedirs(os.path.join(gen, "internals"), exist_ok=True)
    cfg = _read("include/sparsex/config.h.in")
    cfg = cfg.replace("@SPX_INDEX_TYPE@", "int").replace("@SPX_VALUE_TYPE@", "double")
    with open(os.path.join(gen, "config.h"), "w") as f:
        f.write(cfg)
    chpp = _read("include/sparsex/internals/Config.hpp.in")
    chpp = chpp.replace("@SPX_USE_NUMA@", "0")
    chpp = re.sub(r"@[A-Za-z_]+@", "", chpp)
    with open(os.path.join(gen, "internals", "Config.hpp"), "w") as f:
        f.write(chpp)
    return os.path.join(REF_DIR, "gen")


def _subst(text, mapping):
    for k, v in mapping.items():
        text = text.replace("${%s}" % k, v)
    return text



def row_and_column_hooks(symmetric, row_jumps, full_colind):
    """The text CsxJit substitutes for ${new_row_hook} and ${next_x}, and the argument list of a
    unit routine's call -- typed here from include/sparsex/internals/CsxJit.hpp:359-413 (DoNewRowHook)
    and :640-672; tests/test_oracle_golden.py::test_hook_text_equals_the_reference extracts the
    literals from that file and compares (in the container that holds the reference)."""
    hooks = {}
    if not symmetric:
        if row_jumps:
            hooks["new_row_hook"] = ("if (test_bit(&flags, CTL_RJMP_BIT))\n"
                                     "\t\t\t\ty_curr += ul_get(&ctl);\n"
                                     "\t\t\telse\n\t\t\t\ty_curr++;")
        else:
            hooks["new_row_hook"] = "y_curr++;"
        hooks["next_x"] = ("x_curr = x + u32_get(&ctl);" if full_colind
                           else "x_curr += ul_get(&ctl);")
        call = "(&ctl, size, &v, &x_curr, &y_curr, scale_f);"
    else:
        if row_jumps:
            hooks["new_row_hook"] = (
                "if (test_bit(&flags, CTL_RJMP_BIT)) {\n"
                "\t\t\t\tint jmp = ul_get(&ctl);\n"
                "\t\t\t\tfor (i = 0; i < jmp; i++) {\n"
                "\t\t\t\t\ty[y_indx] += x[y_indx] * (*dv) * scale_f;\n"
                "\t\t\t\t\ty_indx++;\n\t\t\t\t\tdv++;\n\t\t\t\t}\n"
                "\t\t\t} else {\n"
                "\t\t\t\ty[y_indx] += x[y_indx] * (*dv) * scale_f;\n"
                "\t\t\t\ty_indx++;\n\t\t\t\tdv++;\n\t\t\t}\n")
        else:
            hooks["new_row_hook"] = ("y[y_indx] += x[y_indx] * (*dv) * scale_f;\n"
                                     "\t\t\ty_indx++;\n\t\t\tdv++;\n")
        hooks["next_x"] = ("x_indx = u32_get(&ctl);" if full_colind
                           else "x_indx += ul_get(&ctl);")
        call = "(&ctl, size, &v, x, y, cur, &x_indx, &y_indx, scale_f);"
    return hooks, call

def generate_source(id_map, symmetric, row_jumps, full_colind):
    """The C text CsxJit would hand to its compiler for this partition."""
    sfx = "_sym" if symmetric else ""
    defs = []
    entries = {}
    for slot, pid in enumerate(id_map):
        if pid < 0:
            break
        t, delta = pid // 10000, pid % 10000
        if t == 0:
            body = _subst(_read("src/templates/delta%s_tmpl.c" % sfx),
                          {"bits": str(delta), "align_ctl": ""})
            name = "delta%d_case" % delta
        elif t in T_NAMES:
            body = _subst(_read("src/templates/%s%s_tmpl.c" % (T_NAMES[t], sfx)),
                          {"delta": str(delta)})
            name = "%s%d_case" % (T_NAMES[t], delta)
        elif 5 <= t <= 12:
            r, c = t - 4, delta
            if symmetric:
                tmpl = "block_row_sym_tmpl.c"
            else:
                tmpl = "block_row_one_tmpl.c" if t == 5 else "block_row_tmpl.c"
            body = _subst(_read("src/templates/" + tmpl), {"r": str(r), "c": str(c)})
            name = "block_row_%dx%d_case" % (r, c)
        elif 13 <= t <= 20:
            r, c = delta, t - 12
            if symmetric:
                tmpl = "block_col_sym_tmpl.c"
            else:
                tmpl = "block_col_one_tmpl.c" if t == 13 else "block_col_tmpl.c"
            body = _subst(_read("src/templates/" + tmpl), {"r": str(r), "c": str(c)})
            name = "block_col_%dx%d_case" % (r, c)
        else:
            raise ValueError("unknown pattern id %d" % pid)
        defs.append(body)
        entries[slot] = name

    hooks = {"spmv_func_definitions": "\n".join(defs)}
    fixed, call = row_and_column_hooks(symmetric, row_jumps, full_colind)
    hooks.update(fixed)
    hooks["body_hook"] = body_hook_text(entries, call)
    main = _read("src/templates/csx%s_spmv_tmpl.c" % sfx)
    return _subst(main, hooks)


def body_hook_text(entries, call):
    """${body_hook} as CsxJit::DoHook writes it (include/sparsex/internals/CsxJit.hpp:637-672): one routine is
    called directly, several through a switch over the pattern slot.  `entries`: {slot: routine name}; `call`: the
    argument list with its semicolon.  tests/test_oracle_golden.py rebuilds the same text from the expression in
    the header itself and compares."""
    if len(entries) == 1:
        return "yr += " + list(entries.values())[0] + call
    body = "switch (patt_id) {\n"
    for slot in sorted(entries):
        body += "\t\tcase %d:\n\t\t\tyr += %s%s\n\t\t\tbreak;\n" % (slot, entries[slot], call)
    body += ("\t\tdefault:\n\t\t\tfprintf(stderr, \"[BUG] unknown pattern\\n\");\n"
             "\t\t\texit(1);\n\t\t};")
    return body


def build(id_map, symmetric=False, row_jumps=False, full_colind=False, opt="-O2"):
    """Returns the path of the .so holding the generated multiply routine."""
    ids = [int(i) for i in id_map if int(i) >= 0]
    if not ids:
        return None
    key = "v2|%s|%d|%d|%d|%s" % (",".join(map(str, ids)), symmetric, row_jumps, full_colind, opt)
    tag = hashlib.sha1(key.encode()).hexdigest()[:16]
    so = os.path.join(REF_DIR, "csxref_%s.so" % tag)
    if os.path.exists(so):
        return so
    if not reference_available():
        return None
    inc = _gen_headers()
    src = os.path.join(REF_DIR, "csxref_%s.c" % tag)
    with open(src, "w") as f:
        f.write(generate_source(ids, symmetric, row_jumps, full_colind))
    # CtlUtil.hpp's u16_get/u32_get advance the byte cursor through a
    # (uint32_t **) cast of it; under gcc's strict-aliasing rules that is
    # undefined (seen to break the full_colind stream at -O2), so the templates
    # are built with the aliasing assumption off
    cmd = ["gcc", "-std=gnu99", opt, "-fno-strict-aliasing", "-fPIC", "-shared", "-w", "-I" + inc,
           "-I" + os.path.join(REF_ROOT, "include"), src, "-o", so]
    subprocess.check_call(cmd)
    return so


def lookup(id_map, symmetric=False, row_jumps=False, full_colind=False, opt="-O2"):
    """Path of a previously built .so for this configuration, or None."""
    ids = [int(i) for i in id_map if int(i) >= 0]
    key = "v2|%s|%d|%d|%d|%s" % (",".join(map(str, ids)), symmetric, row_jumps, full_colind, opt)
    so = os.path.join(REF_DIR, "csxref_%s.so" % hashlib.sha1(key.encode()).hexdigest()[:16])
    return so if os.path.exists(so) else None


# ---- ctypes view of the reference's structs (Csx.hpp:29-53, Vector.hpp:30-35) ----

class RefVector(C.Structure):
    _fields_ = [("elements", C.POINTER(C.c_double)), ("size", C.c_size_t),
                ("alloc_type", C.c_int), ("vec_mode", C.c_int)]


class RefCsxMatrix(C.Structure):
    _fields_ = [("values", C.POINTER(C.c_double)), ("ctl", C.POINTER(C.c_uint8)),
                ("nnz", C.c_int), ("ncols", C.c_int), ("nrows", C.c_int),
                ("ctl_size", C.c_int), ("row_start", C.c_int), ("row_jumps", C.c_uint8),
                ("id_map", C.c_long * 63), ("rows_info", C.c_void_p)]


class RefCsxSymMatrix(C.Structure):
    _fields_ = [("lower_matrix", C.POINTER(RefCsxMatrix)), ("dvalues", C.POINTER(C.c_double))]


def build_test_client():
    """The reference's own test client (test/src/sparsex_test.c), compiled unmodified from where it
    lies and linked against THIS repository's libsparsex.so -> oracle/_ref/sparsex_test.  Its one
    non-API dependency, check_result(), comes from tests/ref_client/ (the reference's version sits on
    its Boost-based internals).  The source is fed through stdin so that `#include "CsxCheck.hpp"`
    resolves to tests/ref_client/CsxCheck.hpp, not to the file next to it."""
    if not reference_available():
        return None
    root = os.path.dirname(HERE)
    os.makedirs(REF_DIR, exist_ok=True)
    exe = os.path.join(REF_DIR, "sparsex_test")
    src = os.path.join(REF_ROOT, "test", "src", "sparsex_test.c")
    client = os.path.join(root, "tests", "ref_client")
    with open(src, "rb") as f:
        text = f.read()
    cmd = ["gcc", "-std=gnu99", "-O1", "-x", "c", "-", os.path.join(client, "check_result.c"),
           "-I" + client, "-I" + os.path.join(root, "include"),
           "-L" + os.path.join(root, "sparsex_amd", "lib"), "-lsparsex", "-lm",
           "-Wl,-rpath,$ORIGIN/../../sparsex_amd/lib", "-o", exe]
    subprocess.run(cmd, input=text, check=True)
    return exe


EXAMPLES = ["csr_example", "mmf_example", "advanced_example", "matrix_caching_example_p1",
            "matrix_caching_example_p2", "reordering_example"]


def build_examples():
    """The reference's six example clients (src/examples/*.c), compiled unmodified from where
    they lie against THIS repository's headers and libsparsex.so -> oracle/_ref/examples/.
    tests/test_gpu_reference_client.py runs them on the GPU."""
    if not reference_available():
        return None
    root = os.path.dirname(HERE)
    out = os.path.join(REF_DIR, "examples")
    os.makedirs(out, exist_ok=True)
    for name in EXAMPLES:
        cmd = ["gcc", "-std=gnu99", "-O1", "-w", os.path.join(REF_ROOT, "src", "examples", name + ".c"),
               "-I" + os.path.join(root, "include"), "-L" + os.path.join(root, "sparsex_amd", "lib"),
               "-lsparsex", "-lm", "-Wl,-rpath,$ORIGIN/../../../sparsex_amd/lib",
               "-o", os.path.join(out, name)]
        subprocess.check_call(cmd)
    return out

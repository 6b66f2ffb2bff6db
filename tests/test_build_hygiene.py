"""Experiment switches stay out of the product's way.

* Every `SPX_ABL_*` macro lives in sparsex_amd/csrc/spx_abl.hpp and nowhere else: the kernels read constexpr
  switches (`abl::sym_no_x` ...) in ordinary `if`s, so every guarded branch is compiled and type-checked in the
  DEFAULT build -- a variant cannot rot.  No other `#if` of the kernel sources depends on an experiment macro.
* The default build leaves every switch off; each macro turns its switch on (and the two umbrella macros theirs).
* The product sources read no experiment environment variables: the ones that remain are listed here with
  their reason.
"""
import glob
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sparsex_amd", "csrc")

SWITCHES = {"SPX_ABL_SYM_NOSLOTADD": ["sym_no_slot_add"], "SPX_ABL_SYM_ONEADD": ["sym_one_add"],
            "SPX_ABL_SYM_NOX": ["sym_no_x"], "SPX_ABL_SYM_NOHANDOVER": ["sym_no_handover"],
            "SPX_ABL_SYM_NOOWN": ["sym_no_own"], "SPX_ABL_SYM_NOINIT": ["sym_no_init"],
            "SPX_ABL_SYM_NOPRIVATE": ["sym_no_private"], "SPX_ABL_SYM_NOTILERUN": ["sym_no_tile_run"], "SPX_ABL_SYM_NOMIXED": ["sym_no_mixed"],
            "SPX_ABL_SYM_STREAM": ["sym_no_slot_add", "sym_no_x", "sym_no_handover"],
            "SPX_ABL_SYM_NOWRITES": ["sym_no_init", "sym_no_own", "sym_no_handover"]}
ALL = sorted({s for v in SWITCHES.values() for s in v})

# environment variables the product sources may read, and why
ALLOWED_ENV = {
    "SPX_HOST_PARTS_MIN_BYTES": "tests: the part-by-part return of y on small matrices",
    "SPX_HOST_XPIECE_BYTES": "tests: x on its way up in many pieces on small matrices",
    "SPX_NO_CSR_FAST_PATH": "tests: the CSR fast path of the partition builder against the general one",
    "SPX_NO_HUGE_PAGES": "operations: hosts where transparent huge pages must not be asked for",
    "SPX_VEC_DEVICE": "spx_options_set_from_env: an unchanged client opts in to resident vectors",
    # the reference's own (include/sparsex/internals/Runtime.cpp / matvec.h: spx_options_set_from_env)
    "SYMMETRIC": "reference", "CPU_AFFINITY": "reference", "NUM_THREADS": "reference", "XFORM_CONF": "reference",
    "WINDOW_SIZE": "reference", "SAMPLES": "reference", "SAMPLING_PORTION": "reference", "SAMPLING": "reference",
}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")) +
                  glob.glob(os.path.join(CSRC, "*.cpp")) + glob.glob(os.path.join(CSRC, "*.h")))


def test_experiment_macros_only_in_spx_abl():
    for f in sources():
        text = open(f).read()
        if os.path.basename(f) == "spx_abl.hpp":
            continue
        assert "SPX_ABL_" not in text and "SPX_EXPERIMENT" not in text and "SPX_BISECT" not in text, f
        # no conditional compilation on anything but include guards / compiler facts
        for m in re.finditer(r"^\s*#\s*(?:if|ifdef|ifndef|elif)\s+(.*)$", text, re.M):
            cond = m.group(1)
            assert re.search(r"_H\b|_HPP\b|__cplusplus|__linux__|MADV_HUGEPAGE|NDEBUG", cond), \
                "%s: conditional on `%s`" % (os.path.basename(f), cond)


def test_every_switch_named_in_the_header_is_used_and_known():
    hdr = open(os.path.join(CSRC, "spx_abl.hpp")).read()
    declared = set(re.findall(r"constexpr bool (\w+) = true;", hdr))
    assert declared == set(ALL), declared ^ set(ALL)
    assert set(re.findall(r"#ifdef (SPX_ABL_\w+)", hdr)) == set(SWITCHES)
    used = set()
    for f in sources():
        used |= set(re.findall(r"abl::(\w+)", open(f).read()))
    assert used == set(ALL), "declared but unused, or used but undeclared: %s" % (used ^ set(ALL))


def _switch_values(defs):
    src = '#include "spx_abl.hpp"\n#include <cstdio>\nint main() { std::printf("' + " ".join("%d" for _ in ALL) + \
          '\\n", ' + ", ".join("(int) spx::abl::" + s for s in ALL) + "); return 0; }\n"
    exe = "/tmp/spx_abl_probe_%d" % os.getpid()
    subprocess.run(["g++", "-std=c++17", "-I" + CSRC, "-x", "c++", "-", "-o", exe] + ["-D" + d for d in defs],
                   input=src.encode(), check=True)
    out = subprocess.check_output([exe]).decode().split()
    os.unlink(exe)
    return {s: int(v) for s, v in zip(ALL, out)}


def test_default_build_leaves_every_switch_off():
    assert not any(_switch_values([]).values())


@pytest.mark.parametrize("macro", sorted(SWITCHES))
def test_each_macro_turns_on_its_switches_and_nothing_else(macro):
    got = _switch_values([macro])
    assert {s for s, v in got.items() if v} == set(SWITCHES[macro])


def test_product_sources_read_no_experiment_environment():
    seen = {}
    for f in sources():
        for name in re.findall(r'getenv\("(\w+)"\)', open(f).read()):
            seen.setdefault(name, os.path.basename(f))
    extra = {k: v for k, v in seen.items() if k not in ALLOWED_ENV}
    assert not extra, "environment hooks in the product sources: %s" % extra


def test_the_round_scripts_only_name_variants_that_exist():
    """tools/r06/*.sh load variant libraries by name: every name must be one of spx_abl.hpp's macros (with or
    without its SPX_ABL_ prefix), or a combination the build script spells out."""
    known = {m[len("SPX_ABL_"):] for m in SWITCHES} | {"SYM_STREAM_NOWRITES"}
    for f in glob.glob(os.path.join(ROOT, "tools", "r06", "*.sh")):
        for name in re.findall(r"libsparsex_(\w+)\.so", open(f).read()):
            if name.startswith("$"):
                continue
            assert name in known, "%s loads an unknown variant %s" % (os.path.basename(f), name)
        for lst in re.findall(r"VARIANTS:-([A-Z_ ]+)\}", open(f).read()):
            for name in lst.split():
                assert name in known, (f, name)

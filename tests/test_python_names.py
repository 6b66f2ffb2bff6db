"""Every global name a function of bench.py, __graft_entry__.py, the package, the oracle scripts and the tools refers to
exists in its module (or is a builtin): bench.py runs its legs only on a GPU box, so a name that is missing in one of
them (round 5: `sx` inside host_api_rate) would otherwise show up only there."""
import builtins
import glob
import os
import symtable

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = sorted([os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")] +
               glob.glob(os.path.join(ROOT, "sparsex_amd", "*.py")) + glob.glob(os.path.join(ROOT, "oracle", "*.py")) +
               glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tests", "*.py")) +
               glob.glob(os.path.join(ROOT, "examples", "*.py")))
MODULE_DUNDERS = {"__file__", "__name__", "__doc__", "__package__", "__spec__", "__builtins__"}


def undefined_globals(path):
    src = open(path).read()
    mod = symtable.symtable(src, path, "exec")
    top = set(mod.get_identifiers()) | MODULE_DUNDERS
    bad = []

    def walk(tab):
        for child in tab.get_children():
            for s in child.get_symbols():
                n = s.get_name()
                if s.is_global() and s.is_referenced() and n not in top and not hasattr(builtins, n):
                    bad.append("%s (line %d): %s" % (child.get_name(), child.get_lineno(), n))
            walk(child)
    walk(mod)
    return bad


@pytest.mark.parametrize("path", FILES, ids=[os.path.relpath(f, ROOT) for f in FILES])
def test_no_undefined_global_names(path):
    assert undefined_globals(path) == []

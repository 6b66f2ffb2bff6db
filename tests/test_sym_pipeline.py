"""The device-side pass headers of the pipelined read-once kernel (sparsex_amd/csrc/sxplan.hpp) on the CPU: an SX
header spells a pass' geometry out for lane 0 (row, first column, slot, steps per lane); decoded here the way
csx_spmv_sx_kernel decodes it, every lane must name exactly the row, columns and slot the stream's own descriptors
name.  Also: what the emitter's passes of their own (spx.gpu.sym_pure_passes) do to the stream -- the same nonzeros,
more single-descriptor passes -- and that the plan only ever takes leading read-once passes of width <= 4."""
import numpy as np
import pytest

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import tune
from stream_decode import Stream, PASS, KIND_BLOCK, KIND_HORIZ, KIND_VERT, KIND_DIAG, KIND_ADIAG

SX, INLINE, SYMSEG, NO_SLOT = 4, 1, 5, 0xFFFFFFFF
SYM = {"spx.matrix.symmetric": "true", "spx.gpu.sym_segments": "true", "spx.preproc.sampling": "none"}

CASES = [
    ("nlpkkt", lambda: synth.syn_nlpkkt(44), dict(SYM)),
    ("nlpkkt-threads", lambda: synth.syn_nlpkkt(44), dict(SYM, **{"spx.rt.nr_threads": "3"})),
    ("nlpkkt-wide2048", lambda: synth.syn_nlpkkt(44), dict(SYM, **{"spx.gpu.sym_wide_rows": "2048"})),
    ("nlpkkt-narrow", lambda: synth.syn_nlpkkt(44), dict(SYM, **{"spx.gpu.sym_wide_rows": "512", "spx.gpu.rowblock_rows": "96"})),
    ("kkt2f", lambda: synth.syn_kkt2f(12), dict(SYM)),
    ("cant", lambda: synth.syn_cant(0.05), dict(SYM)),
    ("cant-min4", lambda: synth.syn_cant(0.04), dict(SYM, **{"spx.gpu.sym_segment_min": "4"})),
    ("nd24k-segments-and-tiles", lambda: synth.syn_nd24k(0.03), dict(SYM)),
    ("no-inline", lambda: synth.syn_nlpkkt(44), dict(SYM, **{"spx.gpu.inline_desc": "false"})),
]


def lanes_of(rb, ps, descs):
    """(row, first column, slot or -1) of every lane of a read-once pass, from the stream's descriptors."""
    nseg, mask = int(ps["nseg"]), int(ps["mask"])
    if int(ps["flags"]) & INLINE:
        mask = 0
    starts = np.array([(mask >> l) & 1 for l in range(nseg)])
    rank = int(ps["rank0"]) + 2 * np.cumsum(starts)
    d = descs[int(rb["desc_off"]) + rank]
    slot0 = descs[int(rb["desc_off"]) + rank + 1]["col0"].astype(np.int64)
    bits = d["bits"].astype(np.int64)
    s = (int(ps["seg0"]) + np.arange(nseg) - ((bits >> 9) & 8191)) & 0xffff
    kind, step = (bits >> 22) & 7, bits >> 25
    drow = np.where(kind == KIND_BLOCK, 1, np.where(kind >= KIND_VERT, step, 0))
    dcol = np.where((kind == KIND_HORIZ) | (kind == KIND_DIAG), step, np.where(kind == KIND_ADIAG, -step, 0))
    row = int(ps["elem0"]) + (bits & 511) + s * drow
    col = d["col0"].astype(np.int64) + s * dcol
    slot = np.where(slot0 == NO_SLOT, -1, slot0 + s * dcol)
    return row, col, slot, len(set(rank.tolist()))


@pytest.mark.parametrize("name,gen,opts", CASES, ids=[c[0] for c in CASES])
def test_sx_headers_name_what_the_stream_names(tmp_path, name, gen, opts):
    A = tune(gen(), opts, host_only=True)
    f = str(tmp_path / "m.spx")
    A.save(f)
    s = Stream(f)
    words, n_sx, cnt = A.sym_pipeline()
    assert len(n_sx) == len(s.rbs) and words.shape == (len(s.passes), 6)
    orig = np.frombuffer(s.passes.tobytes(), dtype="<u4").reshape(-1, 6)
    seen_sx = seen_sym = elems_sx = elems_sym = 0
    for i, rb in enumerate(s.rbs):
        p0, k = int(rb["pass_off"]), int(n_sx[i])
        assert k <= int(rb["n_pass"])
        for t in range(int(rb["n_pass"])):
            ps = s.passes[p0 + t]
            w = words[p0 + t]
            is_sym = int(ps["kind"]) == SYMSEG
            elems = int(ps["nseg"]) * int(ps["width"])
            seen_sym += is_sym
            elems_sym += elems if is_sym else 0
            if t >= k:
                # everything behind the leading SX passes keeps its header, bit for bit
                assert (w == orig[p0 + t]).all() and not ((int(w[4]) >> 24) & SX)
                continue
            assert is_sym and 1 <= int(ps["width"]) <= 4 and int(ps["nseg"]) >= 1
            assert (int(w[4]) >> 24) & SX and (int(w[4]) & 0xffffff) == (int(orig[p0 + t][4]) & 0xffffff)
            assert w[2] == orig[p0 + t][2] and w[5] == orig[p0 + t][5]
            row, col, slot, n_units = lanes_of(rb, ps, s.descs)
            assert n_units == 1
            l = np.arange(int(ps["nseg"]))
            geo = int(w[1])
            drow, dcol = (geo >> 11) & 127, ((geo >> 18) & 255) - 128
            assert ((geo & 2047) + l * drow == row).all()
            assert (int(w[0]) + l * dcol == col).all()
            if int(w[3]) == NO_SLOT:
                assert (slot == -1).all()
            else:
                assert (int(w[3]) + l * dcol == slot).all()
                assert (slot >= 0).all() and (slot + int(ps["width"]) <= int(rb["n_slots"]) + int(rb["n_rows"])).all()
            # what the kernel's x loads rely on: a strictly lower segment never reaches x[row]
            assert (col + int(ps["width"]) - 1 < int(rb["row0"]) + row).all() and (col >= 0).all()
            seen_sx += 1
            elems_sx += elems
    assert cnt["sx_passes"] == seen_sx and cnt["sym_passes"] == seen_sym
    assert cnt["sx_elems"] == elems_sx and cnt["sym_elems"] == elems_sym
    assert cnt["rowblocks_with_sx"] == int((n_sx > 0).sum())
    if name.startswith("nlpkkt") or name == "no-inline":
        assert elems_sx >= 0.5 * elems_sym, "the stencil's long runs should carry their geometry in the header"


def test_general_stream_has_no_sx_passes(tmp_path):
    A = tune(synth.syn_nlpkkt(10), {"spx.preproc.sampling": "none"}, host_only=True)
    words, n_sx, cnt = A.sym_pipeline()
    assert not n_sx.any() and cnt["sx_passes"] == 0 and cnt["sym_passes"] == 0


@pytest.mark.parametrize("gen", [lambda: synth.syn_nlpkkt(44), lambda: synth.syn_kkt2f(12), lambda: synth.syn_cant(0.05)],
                         ids=["nlpkkt", "kkt2f", "cant"])
def test_passes_of_their_own_hold_the_same_nonzeros(tmp_path, gen):
    csr = gen()
    trip = {}
    stats = {}
    for pure in ("true", "false"):
        A = tune(csr, dict(SYM, **{"spx.gpu.sym_pure_passes": pure}), host_only=True)
        f = str(tmp_path / ("m_%s.spx" % pure))
        A.save(f)
        s = Stream(f)
        r, c, v, _ = s.triplets()
        o = np.lexsort((c, r))
        trip[pure] = (r[o], c[o], v[o])
        sym = s.passes[s.passes["kind"] == SYMSEG]
        single = ((sym["flags"] & INLINE) != 0) | (sym["mask"] == 0)
        stats[pure] = (len(sym), int(single.sum()), int((sym["nseg"].astype(np.int64) * sym["width"])[single].sum()))
    for a, b in zip(trip["true"], trip["false"]):
        assert np.array_equal(a, b)
    # more of the read-once nonzeros sit in single-descriptor passes, at the price of some more passes
    assert stats["true"][2] >= stats["false"][2]
    assert stats["true"][0] <= 1.7 * stats["false"][0] + 8

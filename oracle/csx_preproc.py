"""Pure-Python restatement of the reference's CSX preprocessor (small inputs).

TEST INFRASTRUCTURE ONLY (see oracle/csx_oracle.c).  An independent second
restatement -- written from the reference sources, not from the product's
C++ -- of: partitioning, substructure mining, selection, encoding and the ctl
emitter.  tests/test_preproc_oracle.py checks that the product's host
preprocessor produces the very same units and byte streams.

Pinning status: the reference pins none of this (its unit tests print without
asserting) and its encoder cannot be compiled here
(Boost), so agreement of two independent restatements plus the hand-derived
streams of SURVEY.md section 7.0 is what backs the encoder; y-parity is pinned
separately through the SpMV oracle.

Reference locations (paths inside the reference tree):
  partitioning      include/sparsex/internals/SparseInternal.hpp:117-152,
                    SparsePartition.hpp:508-563 (general), :1087-1129 (symmetric)
  transforms        include/sparsex/internals/Xform.hpp:37-248
  transform + sort  SparsePartition.hpp:680-744; windows :775-839; divide/merge :965-1074
  statistics        EncodingManager.hpp:622-813, 1322-1487; Statistics.hpp:36-822;
                    Statistics.cpp:28-87
  selection         EncodingManager.hpp:816-861
  encoding          EncodingManager.hpp:864-1319
  sampling windows  EncodingManager.hpp:561-619, 1489-1599
  ctl emitter       CsxManager.hpp:238-706, CtlBuilder.cpp:32-93, CsxUtil.hpp:58-74,
                    CsxUtil.cpp:30-33, Delta.hpp:35-48
"""
import math

H, V, D, AD = 1, 2, 3, 4
BR1, BR8, BC1, BC8 = 5, 12, 13, 20
NTYPES = 21
SHORT = {"none": 0, "h": H, "v": V, "d": D, "ad": AD, "br": "br", "bc": "bc", "all": "all"}
for _i in range(1, 9):
    SHORT["br%d" % _i] = BR1 + _i - 1
    SHORT["bc%d" % _i] = BC1 + _i - 1


def align_of(t):
    if BR1 <= t <= BR8:
        return t - BR1 + 1
    if BC1 <= t <= BC8:
        return t - BC1 + 1
    return 0


def expand(t):
    if t == "br":
        return list(range(BR1, BR8 + 1))
    if t == "bc":
        return list(range(BC1, BC8 + 1))
    if t == "all":
        return list(range(0, NTYPES))
    return [t]


# ---- Xform.hpp:37-248 ----------------------------------------------------------

def from_h(t, r, c, nr, nc):
    if t == H:
        return r, c
    if t == V:
        return c, r
    if t == D:
        return nr + c - r, min(r, c)
    if t == AD:
        n = r + c - 1
        return n, (r if n <= nc else nc - c + 1)
    a = align_of(t)
    if BR1 <= t <= BR8:
        return (r - 1) // a + 1, (r - 1) % a + a * (c - 1) + 1
    r, c = c, r
    return (r - 1) // a + 1, (r - 1) % a + a * (c - 1) + 1


def to_h(t, r, c, nr, nc):
    if t == H:
        return r, c
    if t == V:
        return c, r
    if t == D:
        return (nr + c - r, c) if r < nr else (c, r + c - nr)
    if t == AD:
        return (c, r - c + 1) if r <= nc else (r + c - nc, nc - c + 1)
    a = align_of(t)
    rr, cc = a * (r - 1) + (c - 1) % a + 1, (c - 1) // a + 1
    if BR1 <= t <= BR8:
        return rr, cc
    return cc, rr


class Elem:
    __slots__ = ("row", "col", "vals", "type", "delta")

    def __init__(self, row, col, vals, type_=0, delta=0):
        self.row, self.col, self.vals, self.type, self.delta = row, col, vals, type_, delta

    @property
    def size(self):
        return len(self.vals)

    def is_unit(self):
        return self.delta != 0


class Part:
    def __init__(self):
        self.elems = []
        self.rowptr = [0]
        self.type = H
        self.nr_rows = self.nr_cols = self.nnz = 0
        self.row_start = 0

    def set_rowptr(self):
        rp = [0]
        prev = 1
        for i, e in enumerate(self.elems):
            if e.row != prev:
                rp.extend([i] * (e.row - prev))
                prev = e.row
        if rp[-1] != len(self.elems):
            rp.append(len(self.elems))
        self.rowptr = rp

    def transform(self, t):
        if self.type == t:
            return
        for e in self.elems:
            r, c = e.row, e.col
            if self.type != H:
                r, c = to_h(self.type, r, c, self.nr_rows, self.nr_cols)
            if t != H:
                r, c = from_h(t, r, c, self.nr_rows, self.nr_cols)
            e.row, e.col = r, c
        self.elems.sort(key=lambda e: (e.row, e.col))
        if self.elems:
            self.set_rowptr()
        self.type = t

    def rows(self):
        for i in range(len(self.rowptr) - 1):
            yield self.elems[self.rowptr[i]:self.rowptr[i + 1]]


# ---- partitioning ----------------------------------------------------------------------

def build_partitions(triplets, nrows, ncols, nparts, symmetric=False):
    """triplets: row-major sorted list of (row, col, val), 1-based."""
    total = (len(triplets) + ncols) // 2 if symmetric else len(triplets)
    parts, pos, taken, row_start = [], 0, 0, 0
    for i in range(nparts):
        limit = (total - taken) // (nparts - i)
        p = Part()
        p.nr_cols, p.row_start = ncols, row_start
        diag = []
        row_prev, cnt = 1, 0
        while pos < len(triplets):
            r, c, v = triplets[pos]
            row = r - row_start
            if not symmetric:
                if row != row_prev:
                    if limit and cnt >= limit:
                        break
                    row_prev = row
                p.elems.append(Elem(row, c, [v]))
                cnt += 1
            elif r > c:
                if row != row_prev:
                    if limit and len(diag) + cnt >= limit and row_prev == row - 1:
                        break
                    row_prev = row
                p.elems.append(Elem(row, c, [v]))
                cnt += 1
            elif r == c:
                diag.append(v)
            pos += 1
        p.set_rowptr()
        p.nnz = cnt
        p.nr_rows = len(p.rowptr) - 1
        p.diag = diag
        got = cnt + (len(diag) if symmetric else 0)
        if symmetric:
            nrows_p = max(p.nr_rows, len(diag))     # see DESIGN.md, deviations
            p.nr_rows_sym = nrows_p
            row_start += nrows_p
        else:
            row_start += p.nr_rows
        taken += got
        parts.append(p)
    if taken != total:
        raise ValueError("matrix has less elements than claimed")
    return parts


# ---- statistics ------------------------------------------------------------------------------

class Stats:
    def __init__(self):
        self.types = {}       # type -> {"inst": {delta: [nnz, units, deltas]}, "tot": [..]}
        self.total = [0, 0, 0]

    def append(self, t, d, nnz, units, deltas=0):
        ts = self.types.setdefault(t, {"inst": {}, "tot": [0, 0, 0]})
        cur = ts["inst"].setdefault(d, [0, 0, 0])
        for k, x in enumerate((nnz, units, deltas)):
            cur[k] += x
            ts["tot"][k] += x
            self.total[k] += x

    @staticmethod
    def _recalc(ts):
        ts["tot"] = [sum(v[k] for v in ts["inst"].values()) for k in range(3)]

    def _finish(self, erase_inst, erase_types, last_recalc):
        for t, d in erase_inst:
            self.types[t]["inst"].pop(d, None)
        for t in erase_types:
            self.types.pop(t, None)
        if last_recalc:
            self.total = [sum(ts["tot"][k] for ts in self.types.values()) for k in range(3)]

    def scale(self, f):
        ei, et, last = [], [], 0
        for t in sorted(self.types):
            ts = self.types[t]
            rec = 0
            for d in sorted(ts["inst"]):
                ts["inst"][d] = [int(x * f) for x in ts["inst"][d]]
                rec += 1
                if ts["inst"][d] == [0, 0, 0]:
                    ei.append((t, d))
            if rec:
                self._recalc(ts)
            if ts["tot"] == [0, 0, 0]:
                et.append(t)
            last = rec
        self._finish(ei, et, last)

    def filter(self, nnz, min_cov, kept):
        ei, et, last = [], [], 0
        for t in sorted(self.types):
            ts = self.types[t]
            rec = 0
            for d in sorted(ts["inst"]):
                if ts["inst"][d][0] / float(nnz) < min_cov:
                    ts["inst"][d] = [0, 0, 0]
                    rec += 1
                else:
                    kept.add((t, d))
                if ts["inst"][d] == [0, 0, 0]:
                    ei.append((t, d))
            if rec:
                self._recalc(ts)
            if ts["tot"] == [0, 0, 0]:
                et.append(t)
            last = rec
        self._finish(ei, et, last)

    def split_blocks(self, max_unit, nnz, min_cov):
        ei, et, last = [], [], 0
        for t in sorted(self.types):
            ts = self.types[t]
            for d in sorted(ts["inst"]):
                if ts["inst"][d] == [0, 0, 0]:
                    ei.append((t, d))
            rec = 1 if _split_type(t, ts["inst"], max_unit, nnz, min_cov) else 0
            if rec:
                self._recalc(ts)
            if ts["tot"] == [0, 0, 0]:
                et.append(t)
            last = rec
        self._finish(ei, et, last)


def _split_data(fixed, var, maxvar, data, inst):
    chunks, rem = var // maxvar, var % maxvar
    maxblock = maxvar * fixed
    nmax = chunks * data[1]
    remnnz = data[0] - nmax * maxblock
    cur = inst.setdefault(maxvar, [0, 0, 0])
    cur[0] += nmax * maxblock
    cur[1] += nmax
    if rem >= 2:
        cur = inst.setdefault(rem, [0, 0, 0])
        cur[0] += remnnz
        cur[1] += data[1]


def _prev(inst, k):
    below = [x for x in inst if x < k]
    return max(below) if below else None


def _split_type(t, inst, max_unit, nnz, min_cov):
    """BlockSplitter::Manipulate (Statistics.cpp:51-87): the reference walks a
    std::map with reverse iterators while inserting smaller keys, i.e. it always
    steps to the largest key below the current one at the time of the step."""
    fixed = align_of(t)
    if not fixed:
        return 0
    maxdim = max_unit // fixed
    ret, erase = 0, []
    k = max(inst) if inst else None
    while k is not None and k * fixed > max_unit:
        _split_data(fixed, k, maxdim, list(inst[k]), inst)
        erase.append(k)
        ret += 1
        k = _prev(inst, k)
    for d in erase:
        inst.pop(d, None)
    erase = []
    ki = max(inst) if inst else None
    kj = ki
    while ki is not None:
        if not (inst[ki][0] / float(nnz) < min_cov):
            while kj is not None and kj >= ki and inst[kj][0] / float(nnz) < min_cov:
                _split_data(fixed, kj, ki, list(inst[kj]), inst)
                erase.append(kj)
                ret += 1
                kj = _prev(inst, kj)
        ki = _prev(inst, ki)
    for d in erase:
        inst.pop(d, None)
    return ret


def _rle(xs):
    """delta encoding (element 0 keeps its value) followed by run lengths"""
    ds = [xs[0]] + [xs[i] - xs[i - 1] for i in range(1, len(xs))]
    out = []
    for d in ds:
        if out and out[-1][1] == d:
            out[-1][0] += 1
        else:
            out.append([1, d])
    return out


def _delta_bytes(v):
    return 1 if v <= 0xFF else 2 if v <= 0xFFFF else 4 if v <= 0xFFFFFFFF else 8


class Encoder:
    def __init__(self, part, opts, nr_threads=1):
        self.p = part
        self.min_limit = int(opts.get("spx.matrix.min_unit_size", 4))
        self.max_limit = int(opts.get("spx.matrix.max_unit_size", 255))
        self.min_cov = float(opts.get("spx.matrix.min_coverage", 0.1))
        self.split = opts.get("spx.matrix.split_blocks", "true") == "true"
        self.cost = opts.get("spx.preproc.heuristic", "ratio") == "cost"
        self.ignore = set(range(NTYPES))
        self.inst = set()
        self.enc_total_deltas = 0
        self.enc_deltas = {}
        self.seq = []
        method = opts.get("spx.preproc.sampling", "portion")
        self.sampling = method != "none"
        if self.sampling:
            samples = int(opts.get("spx.preproc.sampling.nr_samples", 48))
            self.samples = int(math.ceil(float(samples) / nr_threads))
            if method == "portion":
                portion = float(opts.get("spx.preproc.sampling.portion", 0.01))
                self.window = int(portion * part.nnz / self.samples)
            else:
                self.window = int(opts.get("spx.preproc.sampling.window_size", 0))
            self._splits()
            self.samples = min(self.samples, len(self.splits))
            self._select()

    def _splits(self):
        rp = self.p.rowptr
        nr = len(rp) - 1
        self.splits, self.split_nz, cnt = [0], [], 0
        for i in range(nr):
            n = cnt + rp[i + 1] - rp[i]
            if n < self.window:
                cnt = n
            else:
                self.splits.append(i + 1)
                self.split_nz.append(n)
                cnt = 0
        if cnt:
            if not self.split_nz:
                self.split_nz.append(cnt)
                self.splits.append(nr)
                return
            self.split_nz[-1] += cnt
            if cnt > self.window // 2:
                self.splits.append(nr)
            else:
                self.splits[-1] = nr

    def _select(self):
        ns, nsm = len(self.splits), self.samples
        sel = [0] * nsm
        if nsm == ns:
            self.sel = list(range(ns))
            return
        if nsm > ns // 2:
            for i in range(ns // 2):
                sel[i] = i
            nsm -= ns // 2
            ns -= ns // 2
        skip = ns // (nsm + 1)
        for i in range(nsm):
            sel[i] = (i + 1) * skip
        self.sel = sel

    def remove_ignore(self, t):
        for x in expand(t):
            if x in (BR1, BC1):
                continue
            self.ignore.discard(x)

    # -- statistics ---------------------------------------------------------
    def _row_stats(self, part, cols, st):
        t = part.type
        a = align_of(t)
        rles = _rle(cols)
        if a:
            start = 0
            for freq, val in rles:
                start += val
                if val == 1:
                    if start == 1:
                        skip, n = 0, freq
                    else:
                        skip = (start - 2) % a
                        skip = a - skip if skip else 0
                        n = freq + 1
                    n = n - skip if n > skip else 0
                    other = n // a
                    if other >= 2:
                        st.append(t, other, other * a, 1)
                start += val * (freq - 1)
            return
        col, last_patt = 0, False
        for freq, val in rles:
            absorb = col != 0 and not last_patt
            limit = self.min_limit - 1 if absorb else self.min_limit
            if freq > 1 and freq >= limit:
                real = freq + 1 if absorb else freq
                rem = real % self.max_limit
                nnz, units = real, real // self.max_limit + (1 if rem else 0)
                if rem and rem < self.min_limit:
                    units -= 1
                    nnz -= rem
                st.append(t, val, nnz, units)
                last_patt = True
            else:
                last_patt = False
            col += val

    def _gen_stats(self, part, st):
        for row in part.rows():
            if row:
                self._row_stats(part, [e.col for e in row], st)

    def _delta_stats(self, part, st):
        for row in part.rows():
            if row:
                xs = [e.col for e in row]
                mx = max([xs[i] - xs[i - 1] for i in range(1, len(xs))] + [0])
                npatt = -(-len(xs) // self.max_limit)
                st.append(part.type, 0, 0, 0, npatt)

    def gen_all_stats(self):
        st = Stats()
        self.inst = set()
        p = self.p
        if self.sampling and len(p.rowptr) - 1 > self.samples:
            samples_nnz = 0
            p.transform(H)
            for i in range(self.samples):
                s = self.sel[i]
                if s + 1 >= len(self.splits) or s >= len(self.split_nz):
                    break
                ws, we = self.splits[s], self.splits[s + 1]
                if ws >= we - 1:
                    break
                length = min(we - ws, len(p.rowptr) - 1 - ws)
                if length <= 0:       # rows beyond the last represented one
                    break
                es, ee = p.rowptr[ws], p.rowptr[ws + length]
                if es == ee:
                    break
                w = Part()
                w.elems = p.elems[es:ee]
                for e in w.elems:
                    e.row -= ws
                w.set_rowptr()
                w.nr_rows, w.nr_cols, w.nnz, w.type = length, p.nr_cols, ee - es, H
                samples_nnz += self.split_nz[s]
                for t in range(H, NTYPES):
                    if t in self.ignore:
                        continue
                    w.transform(t)
                    self._gen_stats(w, st)
                w.transform(H)
                for e in w.elems:
                    e.row += ws
                p.elems[es:ee] = w.elems
            if samples_nnz:
                st.scale(p.nnz / float(samples_nnz))
            if self.split:
                st.split_blocks(self.max_limit, p.nnz, self.min_cov)
            st.filter(p.nnz, self.min_cov, self.inst)
        else:
            if self.cost:
                self._delta_stats(p, st)
            for t in range(H, NTYPES):
                if t in self.ignore:
                    continue
                p.transform(t)
                self._gen_stats(p, st)
                if align_of(t) and self.split:
                    st.split_blocks(self.max_limit, p.nnz, self.min_cov)
                st.filter(p.nnz, self.min_cov, self.inst)
                if self.cost:
                    self._delta_stats(p, st)
        return st

    def choose(self, st):
        best, best_score = 0, 0
        for t in sorted(st.types):
            nnz, units, deltas = st.types[t]["tot"]
            if self.cost:
                nd = self.enc_total_deltas + deltas
                sw = units if t == 0 else units + nd
                score = 0 if nnz < units + sw else nnz - units - sw
            else:
                score = nnz - units
            if score == 0:
                self.ignore.add(t)
            elif score > best_score:
                best, best_score = t, score
        return best

    # -- encoding ---------------------------------------------------------------
    def _encode_run(self, row_no, cols, vals, out):
        t = self.p.type
        a = align_of(t)
        rles = _rle(cols)
        vi = 0
        col = 0
        if not a:
            for freq, val in rles:
                left = freq
                if freq != 1 and (t, val) in self.inst:
                    col += val
                    start, left = col, freq
                    if col != val and not out[-1].is_unit():
                        start -= val
                        left += 1
                        out.pop()
                        vi -= 1
                    while left >= self.min_limit:
                        n = min(self.max_limit, left)
                        self._emit(out, row_no, start, vals[vi:vi + n], t, val)
                        vi += n
                        start += val * n
                        left -= n
                    col = start - val
                for _ in range(left):
                    col += val
                    out.append(Elem(row_no, col, [vals[vi]]))
                    vi += 1
            return
        for freq, val in rles:
            col += val
            if col == 1:
                skip_f, n = 0, freq
            else:
                skip_f = (col - 2) % a
                skip_f = a - skip_f if skip_f else 0
                n = freq + 1
            n = n - skip_f if n > skip_f else 0
            skip_b = n % a
            n -= skip_b
            if self.split:
                ok = val == 1 and n >= 2 * a
            else:
                ok = val == 1 and (t, n // a) in self.inst and n >= 2 * a
            if ok:
                if col != 1:
                    start = col - 1
                    out.pop()
                    vi -= 1
                else:
                    start = col
                for _ in range(skip_f):
                    out.append(Elem(row_no, start, [vals[vi]]))
                    start += 1
                    vi += 1
                if self.split:
                    other = n // a
                    for (tt, dd) in sorted(self.inst, reverse=True):
                        if tt != t:
                            continue
                        while other >= dd:
                            nb = a * dd
                            self._emit(out, row_no, start, vals[vi:vi + nb], t, dd)
                            start += nb
                            vi += nb
                            n -= nb
                            other -= dd
                    skip_b += n
                else:
                    maxl = self.max_limit // a * a
                    nblocks = n // maxl
                    nb = min(maxl, n)
                    if nblocks == 0:
                        nblocks = 1
                    else:
                        skip_b += n - nb * nblocks
                    for _ in range(nblocks):
                        self._emit(out, row_no, start, vals[vi:vi + nb], t, nb // a)
                        start += nb
                        vi += nb
                for _ in range(skip_b):
                    out.append(Elem(row_no, start, [vals[vi]]))
                    start += 1
                    vi += 1
            else:
                for i in range(freq):
                    out.append(Elem(row_no, col + i * val, [vals[vi]]))
                    vi += 1
            col += val * (freq - 1)

    @staticmethod
    def _emit(out, row, col, vals, t, d):
        if len(vals) == 1:
            out.append(Elem(row, col, list(vals)))
        else:
            out.append(Elem(row, col, list(vals), t, d))

    def encode(self, t):
        if t == 0:
            return
        p = self.p
        p.transform(t)
        out = []
        for row in p.rows():
            if not row:
                continue
            row_no = row[0].row
            cols, vals = [], []
            for e in row:
                if not e.is_unit():
                    cols.append(e.col)
                    vals.append(e.vals[0])
                    continue
                if cols:
                    self._encode_run(row_no, cols, vals, out)
                    cols, vals = [], []
                out.append(e)
            if cols:
                self._encode_run(row_no, cols, vals, out)
        p.elems = out
        p.set_rowptr()
        self.ignore.add(t)

    def encode_all(self):
        if not self.p.nnz:
            return
        while True:
            st = self.gen_all_stats()
            t = self.choose(st)
            if t == 0:
                break
            if t in st.types:
                self.enc_deltas[t] = st.types[t]["tot"][2]
            self.enc_total_deltas = sum(self.enc_deltas.values())
            self.encode(t)
            self.seq.append(t)
        self.p.transform(H)

    def encode_serial(self, seq):
        if not self.p.nnz:
            return
        self.ignore = set(range(NTYPES))
        for t, deltas in seq:
            self.remove_ignore(t)
            for d in deltas:
                self.inst.add((t, d))
            self.encode(t)
            self.ignore.add(t)
        self.p.transform(H)


def parse_xform(s):
    import re
    seq, explicit = [], False
    for m in re.finditer(r"([a-z]+([0-9]*))(\{([0-9]+(,[0-9]+)*)\})?", s):
        deltas = [int(x) for x in m.group(4).split(",")] if m.group(4) else []
        explicit = explicit or bool(deltas)
        seq.append((SHORT[m.group(1)], deltas))
    return seq, explicit


def preprocess(triplets, nrows, ncols, opts):
    """Returns the encoded partitions (horizontal order) of the general path."""
    nparts = int(opts.get("spx.rt.nr_threads", 1))
    sym = opts.get("spx.matrix.symmetric", "false") == "true"
    seq, explicit = parse_xform(opts.get("spx.preproc.xform", "all"))
    parts = build_partitions(triplets, nrows, ncols, nparts, sym)

    def run(p):
        enc = Encoder(p, opts, nparts)
        if explicit:
            enc.encode_serial(seq)
        else:
            for t, _ in seq:
                enc.remove_ignore(t)
            enc.encode_all()

    if not sym:
        for p in parts:
            run(p)
        return parts
    for pid, p in enumerate(parts):
        m1, m2 = Part(), Part()
        for m in (m1, m2):
            m.nr_cols, m.row_start = p.nr_cols, p.row_start
        for e in p.elems:
            (m1 if e.col < p.row_start + 1 else m2).elems.append(Elem(e.row, e.col, list(e.vals)))
        for m in (m1, m2):
            m.set_rowptr()
            m.nnz = len(m.elems)
            m.nr_rows = len(m.rowptr) - 1
        if explicit:
            run(m1)
            run(m2)
        else:
            if pid:
                run(m1)
            else:
                Encoder(m1, opts, nparts)
            run(m2)
        merged = []
        for i in range(len(p.rowptr) - 1):
            for m in (m1, m2):
                if len(m.rowptr) - 1 > i:
                    merged += m.elems[m.rowptr[i]:m.rowptr[i + 1]]
        p.elems = merged
        p.set_rowptr()
        p.nr_rows = getattr(p, "nr_rows_sym", p.nr_rows)
    return parts


def units_of(part):
    return [(e.type, e.delta, e.size, e.row, e.col) for e in part.elems]


# ---- ctl emitter (CsxManager.hpp:301-706) ---------------------------------------------------

def emit_ctl(part, symmetric=False, full_colind=False):
    ctl, values, slots = bytearray(), [], {}
    state = {"new_row": False, "empty": 0, "last_col": 1, "row_jumps": False}

    def varint(v):
        v &= (1 << 64) - 1
        while True:
            b = v & 0x7F
            if v < 0x80:
                ctl.append(b)
                return
            ctl.append(b | 0x80)
            v >>= 7

    def head(pid, size, ucol):
        flag = slots.setdefault(pid, len(slots))
        rowjmp = 0
        if state["new_row"]:
            flag |= 0x80
            state["new_row"] = False
            if state["empty"]:
                rowjmp = state["empty"] + 1
                state["empty"] = 0
                state["row_jumps"] = True
                flag |= 0x40
        ctl.append(flag)
        ctl.append(size)
        if rowjmp:
            varint(rowjmp)
        if full_colind:
            ctl.extend(int(ucol & 0xFFFFFFFF).to_bytes(4, "little"))
        else:
            varint(ucol)

    def add_cols(cols):
        deltas = [cols[0] - state["last_col"]] + [cols[i] - cols[i - 1] for i in range(1, len(cols))]
        state["last_col"] = cols[-1]
        nb = _delta_bytes(max(deltas[1:] + [0]))
        head(nb << 3, len(cols), (cols[0] - 1) if full_colind else deltas[0])
        for d in deltas[1:]:
            ctl.extend(int(d).to_bytes(nb, "little"))

    def add_unit(e):
        a = align_of(e.type)
        pid = e.type * 10000 + (e.size // a if a else e.delta)
        head(pid, e.size, (e.col - 1) if full_colind else e.col - state["last_col"])
        state["last_col"] = e.col + ((e.size - 1) * e.delta if e.type == H else 0)

    nr = len(part.rowptr) - 1
    for i in range(nr):
        row = part.elems[part.rowptr[i]:part.rowptr[i + 1]]
        if not row:
            if not state["new_row"]:
                state["new_row"] = True
            else:
                state["empty"] += 1
            continue
        state["last_col"] = 1
        cols = []
        j = 0
        for phase in range(2 if symmetric else 1):
            while j < len(row):
                e = row[j]
                if symmetric and phase == 0 and not (e.col < part.row_start + 1):
                    break
                if e.is_unit():
                    if cols:
                        add_cols(cols)
                        cols = []
                    add_unit(e)
                    values.extend(e.vals)
                else:
                    if len(cols) == 255:
                        add_cols(cols)
                        cols = []
                    cols.append(e.col)
                    values.append(e.vals[0])
                j += 1
            if cols:
                add_cols(cols)
                cols = []
        state["new_row"] = True
    id_map = [pid for pid, _ in sorted(slots.items(), key=lambda kv: kv[1])]
    return bytes(ctl), values, id_map, state["row_jumps"]

"""Independent numpy decoder of the row-block descriptor stream (the file
``spx_mat_save`` writes; layout: sparsex_amd/csrc/gpu_format.h and the
SavedHeader of api.cpp).  Test infrastructure: it lets the CPU suite check the
HBM layout the kernel walks -- every lane's row, column and value -- without a
GPU.  It shares no code with the emitter or the kernel."""
import struct

import numpy as np

RB = np.dtype([("val_off", "<u8"), ("pass_off", "<u4"), ("desc_off", "<u4"), ("cidx_off", "<u4"),
               ("seg_off", "<u4"), ("cbase", "<u4"), ("row0", "<u4"), ("n_rows", "<u2"),
               ("n_pass", "<u2"), ("cidx_width", "u1"), ("flags", "u1"), ("n_slots", "<u2"),
               ("carry_slot", "<u4"), ("spill_off", "<u4"), ("xwin_base", "<u4"), ("xwin_len", "<u2"),
               ("near_off", "<u2"), ("hi_off", "<u4"), ("pad2", "<u4")])
PASS = np.dtype([("mask", "<u8"), ("val_off", "<u4"), ("rank0", "<u2"), ("seg0", "<u2"),
                 ("nseg", "u1"), ("width", "u1"), ("kind", "u1"), ("flags", "u1"), ("elem0", "<u4")])
DESC = np.dtype([("col0", "<u4"), ("bits", "<u4")])
SHARED = np.dtype([("row", "<u4"), ("first_slot", "<u4"), ("n_slots", "<u4")])
assert RB.itemsize == 64 and PASS.itemsize == 24 and DESC.itemsize == 8

KIND_BLOCK, KIND_HORIZ, KIND_VERT, KIND_DIAG, KIND_ADIAG = range(5)


class Stream:
    def __init__(self, path):
        with open(path, "rb") as f:
            buf = f.read()
        assert buf[:8] == b"SPXHIP14", buf[:8]
        hdr = struct.unpack_from("<4i3Q2i4Q8IQ", buf, 8)
        (self.nrows, self.ncols, self.nnz, self.symmetric, self.nr_partitions, self.first_part,
         self.last_part, self.own_lo, self.own_hi, self.nnz_stored, self.n_unit_elems,
         self.n_delta_elems, self.n_units, self.n_carry, flags, self.n_spill, self.lds_doubles,
         self.waves, self.n_encoded, self.sym_atomic, _, self.checksum) = hdr
        self.sym_fused = bool(flags & 1)
        self.pass_stride = flags >> 1
        self.off = 8 + struct.calcsize("<4i3Q2i4Q8IQ")
        self.buf = buf

        def vec(dt):
            (n,) = struct.unpack_from("<Q", self.buf, self.off)
            self.off += 8
            a = np.frombuffer(self.buf, dtype=dt, count=n, offset=self.off)
            self.off += n * np.dtype(dt).itemsize
            return a
        self.bounds = vec("<i4").reshape(-1, 3)
        self.rbs = vec(RB)
        self.passes = vec(PASS)
        self.descs = vec(DESC)
        self.cidx = vec("u1")
        self.segrows = vec("<u2")
        self.shared = vec(SHARED)
        self.dvalues = vec("<f8")
        self.values = vec("<f8")
        self.fix_ptr = vec("<u4")
        self.fix_idx = vec("<u4")
        self.slot_group_col = vec("<u4")
        self.mirror_rows, self.mirror_ptr = vec("<u4"), vec("<u4")     # symmetric slice: thin mirror image, per row
        self.mirror_col, self.mirror_val = vec("<u4"), vec("<f8")
        self.perm = vec("<i4")
        # the encoded partitions (kept for exports of a restored matrix) follow
        self.encoded = []
        for _ in range(self.n_encoded):
            ph = struct.unpack_from("<4Q2i", self.buf, self.off)
            self.off += struct.calcsize("<4Q2i")
            elems, rowptr, pool, diag = vec("V32"), vec("<i4"), vec("<f8"), vec("<f8")
            assert len(elems) == ph[3]
            self.encoded.append((ph, elems, rowptr, pool, diag))
        assert self.off == len(buf)

    def triplets(self):
        """(row, col, value, rowblock) of every stored nonzero, as the kernel's lanes see them."""
        R, Cc, V, B = [], [], [], []
        self._tile_adds = []         # (row-block, slot of the column, value, global row) per tile nonzero
        for bi, rb in enumerate(self.rbs):
            vbase = int(rb["val_off"])
            for ps in self.passes[int(rb["pass_off"]):int(rb["pass_off"]) + int(rb["n_pass"])]:
                pv = vbase + int(ps["val_off"])
                assert pv % 2 == 0                    # 16-byte aligned lane loads
                if ps["kind"] in (0, 5):
                    # unit pass; kind 5: read-once row segments of a symmetric matrix (two descriptor
                    # entries per unit, the second holds the slot; every value counts twice)
                    sym = ps["kind"] == 5
                    nseg, W, mask = int(ps["nseg"]), int(ps["width"]), int(ps["mask"])
                    inline = bool(int(ps["flags"]) & 1)   # the pass' only descriptor sits in its header (`mask`)
                    if inline:
                        one = np.zeros(1, dtype=DESC)
                        one["col0"], one["bits"] = mask & 0xffffffff, mask >> 32
                        assert one[0] == self.descs[int(rb["desc_off"]) + int(ps["rank0"])], "inline descriptor differs"
                        mask = 0
                    assert 1 <= nseg <= 64 and 1 <= W <= 8 and not (mask & 1)
                    lanes = np.arange(nseg)
                    starts = np.array([(mask >> l) & 1 for l in range(nseg)])
                    rank = int(ps["rank0"]) + (2 if sym else 1) * np.cumsum(starts)
                    d = np.repeat(one, nseg) if inline else self.descs[int(rb["desc_off"]) + rank]
                    bits = d["bits"].astype(np.int64)
                    s = (int(ps["seg0"]) + lanes - ((bits >> 9) & 8191)) & 0xffff
                    kind, step = (bits >> 22) & 7, bits >> 25
                    drow = np.where(kind == KIND_BLOCK, 1, np.where(kind >= KIND_VERT, step, 0))
                    dcol = np.where((kind == KIND_HORIZ) | (kind == KIND_DIAG), step,
                                    np.where(kind == KIND_ADIAG, -step, 0))
                    row = int(ps["elem0"]) + (bits & 511) + s * drow       # (elem0: first row of the pass's part)
                    col = d["col0"].astype(np.int64) + s * dcol
                    assert (row < int(rb["n_rows"])).all(), "segment leaves its row-block"
                    if sym:
                        assert self.symmetric and self.sym_atomic
                        slot0 = self.descs[int(rb["desc_off"]) + rank + 1]["col0"].astype(np.int64)
                        has = slot0 != 0xFFFFFFFF
                        slot = np.where(has, slot0 + s * dcol, -1)
                        assert (col + W - 1 < row + int(rb["row0"])).all()       # strictly below the diagonal
                        assert (slot[has] + W <= int(rb["n_slots"]) + int(rb["n_rows"])).all()
                    for w in range(W):
                        pair = w >> 1
                        if (W & 1) and w == W - 1:
                            idx = pair * 2 * nseg + lanes
                        else:
                            idx = pair * 2 * nseg + lanes * 2 + (w & 1)
                        R.append(row + int(rb["row0"])); Cc.append(col + w)
                        V.append(self.values[pv + idx]); B.append(np.full(nseg, bi))
                        if sym:
                            v = self.values[pv + idx]
                            grow = row + int(rb["row0"])
                            # slotted lanes hand their transposed products to the slots, the others add to y
                            R.append((col + w)[has]); Cc.append(grow[has]); V.append(v[has]); B.append(np.full(int(has.sum()), -1 - bi))
                            self._tile_adds.append((bi, (slot + w)[has], v[has], grow[has]))
                            R.append((col + w)[~has]); Cc.append(grow[~has]); V.append(v[~has]); B.append(np.full(int((~has).sum()), bi))
                elif ps["kind"] in (2, 4):
                    # gather pass: lane l owns W leftover nonzeros of one row; kind 4: their
                    # columns lie in the row-block's x window (u16 offsets from xwin_base)
                    nseg, W = int(ps["nseg"]), int(ps["width"])
                    assert 1 <= nseg <= 64 and 1 <= W <= 8
                    lanes = np.arange(nseg)
                    sr = self.segrows[int(rb["seg_off"]) + int(ps["seg0"]) + lanes].astype(np.int64)
                    row, plen = sr & 2047, ((sr >> 11) & 7) + 1        # a pass is as wide as its longest piece
                    assert (row < int(rb["n_rows"])).all() and (plen <= W).all() and plen.max() == W
                    near = ps["kind"] == 4
                    cw = 2 if near else int(rb["cidx_width"])
                    cbase = int(rb["xwin_base"]) if near else int(rb["cbase"])
                    area = (int(rb["cidx_off"]) + (int(rb["near_off"]) if near else 0)) * 16
                    e0 = int(ps["elem0"])
                    for w in range(W):
                        pair = w >> 1
                        if (W & 1) and w == W - 1:
                            idx = pair * 2 * nseg + lanes
                        else:
                            idx = pair * 2 * nseg + lanes * 2 + (w & 1)
                        if cw == 3:         # u16 low halves, then (at hi_off) a u8 array of bits 16-23
                            o = area + (e0 + w * nseg + lanes) * 2
                            off = self.cidx[o].astype(np.int64) | (self.cidx[o + 1].astype(np.int64) << 8)
                            off |= self.cidx[area + int(rb["hi_off"]) * 16 + e0 + w * nseg + lanes].astype(np.int64) << 16
                        else:
                            o = area + (e0 + w * nseg + lanes) * cw
                            off = sum(self.cidx[o + b].astype(np.int64) << (8 * b) for b in range(cw))
                        if near:
                            assert (off < int(rb["xwin_len"])).all() and int(rb["xwin_len"]) <= 4096
                        have = plen > w                            # (padding: zero value, offset 0)
                        assert (self.values[pv + idx][~have] == 0).all() and (off[~have] == 0).all()
                        R.append((row + int(rb["row0"]))[have]); Cc.append((off + cbase)[have])
                        V.append(self.values[pv + idx][have]); B.append(np.full(int(have.sum()), bi))
                elif ps["kind"] == 3:
                    # symmetric tiles: lanes 8t..8t+7 = rows of tile t; every value counts twice
                    nseg = int(ps["nseg"])
                    assert nseg % 8 == 0 and 8 <= nseg <= 64 and int(ps["width"]) == 8 and self.symmetric
                    lanes = np.arange(nseg)
                    d = self.descs[int(rb["desc_off"]) + int(ps["rank0"]) + (lanes >> 3)]
                    bits = d["bits"].astype(np.int64)
                    row = int(ps["elem0"]) + (bits & 511) + (lanes & 7)
                    slot = bits >> 9
                    assert (row < int(rb["n_rows"])).all()
                    assert (slot + 8 <= int(rb["n_slots"]) + int(rb["n_rows"])).all()
                    col0 = d["col0"].astype(np.int64)
                    assert (col0 + 7 < row + int(rb["row0"])).all()          # strictly below the diagonal
                    for w in range(8):
                        idx = (w >> 1) * 2 * nseg + lanes * 2 + (w & 1)
                        v = self.values[pv + idx]
                        R.append(row + int(rb["row0"])); Cc.append(col0 + w); V.append(v); B.append(np.full(nseg, bi))
                        R.append(col0 + w); Cc.append(row + int(rb["row0"])); V.append(v); B.append(np.full(nseg, -1 - bi))
                        self._tile_adds.append((bi, slot + w, v, row + int(rb["row0"])))
                else:
                    raise AssertionError("unknown pass kind %d" % int(ps["kind"]))
        if len(self.mirror_rows):
            assert self.mirror_ptr.size == self.mirror_rows.size + 1 and int(self.mirror_ptr[-1]) == self.mirror_col.size
            assert (np.diff(self.mirror_rows.astype(np.int64)) > 0).all() and self.mirror_rows.max() < self.own_lo
            cnt = np.diff(self.mirror_ptr.astype(np.int64))
            R.append(np.repeat(self.mirror_rows.astype(np.int64), cnt)); Cc.append(self.mirror_col.astype(np.int64))
            V.append(self.mirror_val); B.append(np.full(self.mirror_col.size, len(self.rbs)))
        if not R:
            z = np.zeros(0, dtype=np.int64)
            return z, z, np.zeros(0), z
        return np.concatenate(R), np.concatenate(Cc), np.concatenate(V), np.concatenate(B)

    def matvec(self, x):
        """y = A x the way the kernels compute it: nonzeros into the rows of their row-block,
        the transposed contributions of the symmetric tiles through their slots -- own rows
        directly, the others through spill[] and the per-row fix lists -- and the diagonal from
        dvalues."""
        r, c, v, b = self.triplets()
        y = np.zeros(self.nrows)
        direct = b >= 0                     # (mirrored tile entries are routed through the slots below)
        np.add.at(y, r[direct], v[direct] * x[c[direct]])
        spill = np.zeros(self.n_spill)
        for bi, slot, val, grow in self._tile_adds:
            rb = self.rbs[bi]
            ns = int(rb["n_slots"])
            contrib = val * x[grow]
            own = slot >= ns
            np.add.at(y, int(rb["row0"]) + slot[own] - ns, contrib[own])
            np.add.at(spill, int(rb["spill_off"]) + slot[~own], contrib[~own])
        if self.n_spill:
            assert self.fix_ptr.size == self.nrows + 1 and self.fix_idx.size == self.n_spill
            assert np.array_equal(np.sort(self.fix_idx), np.arange(self.n_spill))
            sums = np.add.reduceat(np.append(spill[self.fix_idx], 0.0), self.fix_ptr[:-1].astype(np.int64))
            sums[np.diff(self.fix_ptr.astype(np.int64)) == 0] = 0.0
            y += sums
        if self.symmetric:
            y[self.own_lo:self.own_hi] += (self.dvalues * x[:self.dvalues.size])[self.own_lo:self.own_hi]
        return y

    def check_ownership(self):
        """Unshared row-blocks own disjoint row ranges; shared ones own single rows listed in `shared`."""
        owner = np.zeros(self.nrows + 1, dtype=np.int64)
        for rb in self.rbs:
            if rb["flags"] & 1:
                assert rb["n_rows"] == 1
                continue
            owner[int(rb["row0"]):int(rb["row0"]) + int(rb["n_rows"])] += 1
        assert owner.max(initial=0) <= 1
        srows = set(int(s["row"]) for s in self.shared)
        for rb in self.rbs:
            if rb["flags"] & 1:
                assert int(rb["row0"]) in srows and owner[int(rb["row0"])] == 0
        return owner

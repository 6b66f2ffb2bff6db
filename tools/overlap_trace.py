#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace of ONE rank of a multi-rank bench run and shows, for the last few
steps, how the parts of the product (csx_spmv_kernel on the launch stream) and the rounds of the halo
exchange (dist_pack_pos_kernel / dist_scatter_pos_kernel on the second stream) lie in time.
usage: tools/overlap_trace.py <dir with *kernel_trace.csv> [steps]"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            kind = ("part" if ("csx_spmv_kernel" in n or "csx_spmv_det_kernel" in n) else "pack" if "dist_pack_pos_kernel" in n else
                    "scatter" if "dist_scatter_pos_kernel" in n else None)
            if kind:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, r.get("Stream_Id", r.get("Queue_Id", "?"))))
    rows.sort()
    packs = [i for i, r in enumerate(rows) if r[2] == "pack"]
    if not packs:
        print("no pack kernels in the trace")
        return
    # the last `steps` overlapped steps: from a few parts in front of the (steps * 4)-th last pack kernel
    # to the last scatter kernel (later phases of the bench -- plain halo step, kernels only -- follow)
    scat = [i for i, r in enumerate(rows) if r[2] == "scatter"]
    first = packs[max(0, len(packs) - steps * 4)]
    while first > 0 and rows[first - 1][2] == "part" and rows[first][0] - rows[first - 1][0] < 2_000_000:
        first -= 1
    tail = rows[first:scat[-1] + 1]
    t0 = tail[0][0]
    print("| kernel | stream / queue | start us | end us | overlaps a part of the product |")
    print("|---|---|---|---|---|")
    parts = [r for r in tail if r[2] == "part"]
    hidden = total = 0
    for s, e, kind, q in tail:
        ov = ""
        if kind != "part":
            o = sum(max(0, min(e, pe) - max(s, ps)) for ps, pe, _, _ in parts)
            ov = "%.0f %% of its %.1f us" % (100.0 * o / max(e - s, 1), (e - s) / 1e3)
            hidden += o
            total += e - s
        print("| %s | %s | %.1f | %.1f | %s |" % (kind, q, (s - t0) / 1e3, (e - t0) / 1e3, ov))
    print("\npack + scatter time running while a part of the product runs: %.0f %%" % (100.0 * hidden / max(total, 1)))


if __name__ == "__main__":
    main()

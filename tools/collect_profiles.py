#!/usr/bin/env python3
"""Copies the summaries tools/refresh_profiles.sh left under gpurun_out/<round>_<w>/
into profiles/<round>/ and regenerates profiles/traffic.json from the PMC passes.
usage: tools/collect_profiles.py r02"""
import json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
traffic = {}
# gpurun_out tag -> key bench.py looks up (workload[-e<edge>][-sym])
KEYS = {"nlpkkt": None, "nlpkkt_sym": None, "cant": "syn-cant", "nd24k_sym": "syn-nd24k-sym", "webbase": "syn-webbase"}
for w, key in KEYS.items():
    src = os.path.join(ROOT, "gpurun_out", "%s_%s" % (rnd, w))
    if not os.path.isdir(src):
        continue
    for a, b in (("kernel_stats.csv", "%s_kernel_stats.csv"), ("bench_line.json", "%s_bench_line_under_rocprof.json"),
                 ("pmc_FETCH_SIZE.txt", "%s_pmc_FETCH_SIZE.txt"), ("pmc_WRITE_SIZE.txt", "%s_pmc_WRITE_SIZE.txt"),
                 ("bench_plain.json", "bench_%s.json"), ("bench_plain_sym.json", "bench_%s_symmetric.json"),
                 ("bench_plain_general.json", "bench_%s_general_path.json")):
        p = os.path.join(src, a)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            shutil.copy(p, os.path.join(dst, b % w))
    if key is None:
        try:
            line = json.load(open(os.path.join(src, "bench_line.json")))
            edge = re.search(r"grid edge (\d+)", line["config"]["workload"]).group(1)
            key = "syn-nlpkkt-e%s%s" % (edge, "-sym" if w.endswith("_sym") else "")
        except Exception:
            continue
    # The launch autotuner inside spx_mat_tune runs every variant (2/4/8 wavefronts, both
    # hand-over modes of the symmetric tiles) a few hundred times, also under the profiler,
    # whose serialised launches can tip its choice.  The variant that counts is the one the
    # un-profiled bench line names; its per-launch counters are in the PMC files either way.
    want = None
    try:
        plain = json.load(open(os.path.join(ROOT, "gpurun_out", "%s_nlpkkt" % rnd, "bench_plain.json")))
        cfg = {"cant": "syn-cant", "nd24k_sym": "syn-nd24k --symmetric", "webbase": "syn-webbase",
               "nlpkkt_sym": "syn-nlpkkt --symmetric (the bench matrix)"}.get(w)
        name = plain["configs"][cfg]["roofline"]["kernel"] if cfg else plain["roofline"]["kernel"]
        want = re.search(r"csx_spmv[a-z_]*kernel<\d+(?:, \d+)*>", name).group(0)
    except Exception:
        pass
    # per counter: {kernel (as the profiler names it): (launches, KiB per launch)}; ONE kernel is chosen for both
    # counters -- the one the un-profiled line names where the profiled run launched it, else the SpMV kernel
    # with the most launches in both files -- and the entry is labelled with the kernel whose numbers it holds
    seen = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        p = os.path.join(src, "pmc_%s.txt" % c)
        seen[c] = {}
        if not os.path.exists(p):
            continue
        for line in open(p):
            if "csx_spmv" not in line:
                continue
            name = line.split(" launches=")[0].strip()
            seen[c][name] = (int(re.search(r"launches=([0-9]+)", line).group(1)),
                             float(re.search(r"per_launch=([0-9.]+)", line).group(1)))
    both = [k for k in seen["FETCH_SIZE"] if k in seen["WRITE_SIZE"]]
    kib, kern = {}, None
    fam = {c: [k for k in seen[c] if want is not None and ("::" + want.split("<")[0] + "<") in k] for c in seen}
    if want is not None and not any(("::" + want.split("<")[0] + "<") in k for k in both) and all(fam.values()):
        # the two passes settled on different wavefront counts of the bench loop's kernel (a small matrix whose
        # variants tie): each counter from its own pass, the entry names both
        pick = {c: max(fam[c], key=lambda k: seen[c][k][0]) for c in fam}
        kib = {c: seen[c][pick[c]][1] for c in pick}
        kern = "%s [FETCH_SIZE pass] / %s [WRITE_SIZE pass]" % (pick["FETCH_SIZE"], pick["WRITE_SIZE"])
    elif both:
        named = [k for k in both if want is not None and ("::" + want + "(") in k]
        # (the launch tuner's trials are in the files too: where the profiled run settled on another wavefront count
        # than the plain line, the same kernel with another count stands in -- never a kernel of another family)
        family = [k for k in both if want is not None and ("::" + want.split("<")[0] + "<") in k]
        pool = named or family or both
        kern = max(pool, key=lambda k: seen["FETCH_SIZE"][k][0])
        kib = {c: seen[c][kern][1] for c in ("FETCH_SIZE", "WRITE_SIZE")}
    if len(kib) == 2:
        traffic[key] = {
            "kernel": kern,
            "kernel_is_the_plain_line_s": bool(want is not None and ("::" + want + "(") in kern and " / " not in kern),
            "fetch_size_kib_per_launch": kib["FETCH_SIZE"],
            "write_size_kib_per_launch": kib["WRITE_SIZE"],
            "hbm_bytes_per_launch": int((2 * kib["FETCH_SIZE"] + kib["WRITE_SIZE"]) * 1024),
            "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over the bench command, the bench "
                    "loop's SpMV kernel only; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md "
                    "(calibrated for wide coalesced reads)",
        }
if traffic:
    path = os.path.join(ROOT, "profiles", "traffic.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    old.update(traffic)
    json.dump(old, open(path, "w"), indent=1)
print(json.dumps(traffic, indent=1))

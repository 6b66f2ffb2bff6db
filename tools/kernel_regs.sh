#!/bin/bash
# Registers, scratch and LDS of every kernel of spmv_kernels.hip (cross-compiled, no GPU needed):
#   tools/kernel_regs.sh [pattern] [extra hipcc flags]
# leaves the assembly in /tmp/spx_asm/spmv.s
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/spx_asm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -munsafe-fp-atomics -Iinclude -Isparsex_amd/csrc \
    ${2:-} -S --cuda-device-only -o /tmp/spx_asm/spmv.s sparsex_amd/csrc/${SPX_TU:-spmv_kernels}.hip 2>/dev/null
python3 - "${1:-.}" <<'PY'
import re, sys
pat = re.compile(sys.argv[1])
cur = {}
for line in open("/tmp/spx_asm/spmv.s"):
    m = re.match(r"\s+\.(name|sgpr_count|vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|agpr_count):\s+(\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "name":
        cur = {"name": v}
    cur[k] = v
    if k == "vgpr_spill_count" and pat.search(cur["name"]):
        name = re.sub(r"^_ZN3spx\d+", "", cur["name"])[:48]
        print("%-48s vgpr %3s sgpr %3s scratch %4s vspill %3s sspill %3s" % (
            name, cur.get("vgpr_count"), cur.get("sgpr_count"), cur.get("private_segment_fixed_size", "?"),
            cur.get("vgpr_spill_count"), cur.get("sgpr_spill_count", "?")))
PY

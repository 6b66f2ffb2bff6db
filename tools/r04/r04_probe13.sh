#!/bin/bash
# round-4 probe 13: where spx_mat_tune spends its time on the contract matrix (INFO log), with the CSR partition fast path
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04m; mkdir -p $OUT; cd $ROOT
python3 - > $OUT/tune_phases.txt 2>&1 <<'PY'
import sys, time, os
sys.path.insert(0, ".")
import torch
import sparsex_amd as sx
from sparsex_amd import synth
import bench
torch.cuda.set_device(0)
csr = synth._rows("nlpkkt", 240, 0, None, None, synth.SEED_BASE + 4)
for sym in ("false", "true"):
    for walk in (False, True):
        if walk: os.environ["SPX_NO_CSR_FAST_PATH"] = "1"
        else: os.environ.pop("SPX_NO_CSR_FAST_PATH", None)
        sx.lib().spx_log_info_console()
        t = time.time()
        A = bench.tune(csr, {"spx.rt.nr_threads": 32, "spx.rt.keep_encoded": "false", "spx.matrix.symmetric": sym})
        i = A.info()
        print("== symmetric %s, %s: tune %.2f s, emit + upload + launch autotune %.2f s, wall %.2f s" % (
            sym, "element walk" if walk else "CSR fast path", i.tune_seconds, i.emit_seconds, time.time() - t), flush=True)
        A.destroy()
PY
grep -v "^\[INFO\]: \(Format\|launch\)" $OUT/tune_phases.txt | grep "==\|partitions\|descriptor stream" | cut -c1-200

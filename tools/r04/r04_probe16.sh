#!/bin/bash
# round-4 probe 16: spx_mat_tune phases on the contract matrix after the counting-pass transform (INFO log), both paths;
# then the stream hashes on the box (must equal the ones taken before the change) and the parity file of the suite
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04p; mkdir -p $OUT; cd $ROOT
echo "transparent_hugepage: $(cat /sys/kernel/mm/transparent_hugepage/enabled) defrag: $(cat /sys/kernel/mm/transparent_hugepage/defrag)" > $OUT/thp.txt; cat $OUT/thp.txt
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
    sx.lib().spx_log_info_console()
    t = time.time()
    A = bench.tune(csr, {"spx.rt.nr_threads": 32, "spx.rt.keep_encoded": "false", "spx.matrix.symmetric": sym})
    i = A.info()
    print("== symmetric %s: tune %.2f s, emit + upload + launch autotune %.2f s, wall %.2f s" % (
        sym, i.tune_seconds, i.emit_seconds, time.time() - t), flush=True)
    A.destroy()
PY
grep -v "^\[INFO\]: \(Format\|launch\)" $OUT/tune_phases.txt | grep "==\|partitions\|descriptor stream\|ranges\|released" | cut -c1-200
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_client.py -x -q -m gpu --durations=6 ) 2>&1 | tail -14

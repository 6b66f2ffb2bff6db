#!/bin/bash
# round-3 probe 9: launch order by strips across the planes (spx.gpu.band_order), A/B inside one process each
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03j; mkdir -p $OUT; cd $ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_save_restore.py tests/test_get_set_entry.py -x -q -n 4 > $OUT/pytest_parity.log 2>&1; tail -n 3 $OUT/pytest_parity.log
S=$OUT/probe9.md
python tools/abl.py syn-nlpkkt --edge 120 --header band: plain:spx.gpu.band_order=false band2: plain2:spx.gpu.band_order=false > $S 2>$OUT/err.txt
python tools/abl.py syn-nlpkkt --edge 120 --symmetric band: plain:spx.gpu.band_order=false >> $S 2>>$OUT/err.txt
python tools/abl.py syn-nlpkkt --edge 240 --steps 30 band: plain:spx.gpu.band_order=false band2: plain2:spx.gpu.band_order=false >> $S 2>>$OUT/err.txt
python tools/abl.py syn-nlpkkt --edge 240 --steps 30 --symmetric band: plain:spx.gpu.band_order=false >> $S 2>>$OUT/err.txt
python tools/abl.py syn-kkt2f --edge 120 band: plain:spx.gpu.band_order=false >> $S 2>>$OUT/err.txt
python tools/abl.py syn-kkt2f --edge 120 --symmetric band: plain:spx.gpu.band_order=false >> $S 2>>$OUT/err.txt
cat $S; tail -n 3 $OUT/err.txt

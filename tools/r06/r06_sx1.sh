#!/bin/bash
# round 6: first A/B of the pipelined read-once kernel (csx_spmv_sx_kernel) -- parity on the symmetric GPU tests,
# then e240 symmetric: plain kernel on the former emission / on passes of their own / pipelined, 4 and 8 wavefronts,
# one and two passes per round
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06sx1; mkdir -p $OUT; cd $ROOT
timeout 900 python3 -m pytest tests -m gpu -x -q -k "sym" > $OUT/pytest_sym.txt 2>&1; tail -3 $OUT/pytest_sym.txt
R=$OUT/sx_first_ab.md; : > $R
SETS="oldemit_plain:spx.gpu.sym_pure_passes=false,spx.gpu.sym_pipeline=false plain:spx.gpu.sym_pipeline=false sx:spx.gpu.sym_pipeline=true sx_w4:spx.gpu.sym_pipeline=true,spx.gpu.waves=4 sx_w8:spx.gpu.sym_pipeline=true,spx.gpu.waves=8 auto: plain2:spx.gpu.sym_pipeline=false sx2:spx.gpu.sym_pipeline=true"
echo "two passes per round" >> $R
timeout 1200 python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 50 --header $SETS 2>$OUT/abl_b2.err | tee -a $R
echo "one pass per round (SPX_SX_PASSES_PER_ROUND=1)" >> $R
SPX_SX_PASSES_PER_ROUND=1 timeout 1200 python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 50 sx:spx.gpu.sym_pipeline=true sx_w4:spx.gpu.sym_pipeline=true,spx.gpu.waves=4 sx_w8:spx.gpu.sym_pipeline=true,spx.gpu.waves=8 2>$OUT/abl_b1.err | tee -a $R
grep -h "read-once pipeline" $OUT/*.err | sort | uniq -c | head

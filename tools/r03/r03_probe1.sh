#!/bin/bash
# round-3 probe: host topology, mining ablation (default vs xform=none), symmetric row-block widths
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03b; mkdir -p $OUT; cd $ROOT
{ lscpu | head -30; echo; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; numactl -H 2>/dev/null | head -20; python -c "import os;print(sorted(os.sched_getaffinity(0))[:8], len(os.sched_getaffinity(0)))"; } > $OUT/host.txt 2>&1
M=$OUT/mining.md
python tools/abl.py syn-cant --header default: none:spx.preproc.xform=none > $M 2>$OUT/mining.err
python tools/abl.py syn-nd24k --symmetric default: none:spx.preproc.xform=none >> $M 2>>$OUT/mining.err
python tools/abl.py syn-nd24k default: none:spx.preproc.xform=none >> $M 2>>$OUT/mining.err
python tools/abl.py syn-webbase default: none:spx.preproc.xform=none >> $M 2>>$OUT/mining.err
python tools/abl.py syn-nlpkkt --edge 120 default: none:spx.preproc.xform=none norecut:spx.gpu.recut_linear=false >> $M 2>>$OUT/mining.err
python tools/abl.py syn-nlpkkt --edge 120 --symmetric default: none:spx.preproc.xform=none >> $M 2>>$OUT/mining.err
python tools/abl.py syn-kkt2f --edge 100 default: none:spx.preproc.xform=none >> $M 2>>$OUT/mining.err
S=$OUT/sym_widths.md
python tools/abl.py syn-nlpkkt --edge 120 --symmetric --header w512:spx.gpu.sym_wide_rows=512 w1024:spx.gpu.sym_wide_rows=1024 w2048:spx.gpu.sym_wide_rows=2048 nosegs:spx.gpu.sym_segments=false w512x8:spx.gpu.sym_wide_rows=512,spx.gpu.waves=8 w512x4:spx.gpu.sym_wide_rows=512,spx.gpu.waves=4 > $S 2>$OUT/sym.err
python tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 30 w512:spx.gpu.sym_wide_rows=512 >> $S 2>>$OUT/sym.err
cat $M $S; tail -3 $OUT/*.err

#!/bin/bash
# probe 20: the next pair of pass headers fetched one round ahead in every kernel (-DSPX_PASS_PREFETCH)
# against the same sources without it; two libraries built side by side in exp_libs/, one process each
O=gpurun_out/r03v; mkdir -p $O
run() { # lib label args...
  local lib=$1; shift
  SPX_LIB_PATH=$PWD/exp_libs/libsparsex_$lib.so timeout 400 python3 tools/abl.py "$@" 2>>$O/err.txt | sed "s/^| /| $lib | /" >> $O/pass_prefetch.md
}
run base syn-nlpkkt --edge 240 --steps 100 default:
run pf   syn-nlpkkt --edge 240 --steps 100 default:
run base syn-nlpkkt --edge 240 --steps 100 default:
run pf   syn-nlpkkt --edge 240 --steps 100 --symmetric default:
run base syn-nlpkkt --edge 240 --steps 100 --symmetric default:
for w in syn-cant syn-webbase; do
  run base $w --steps 400 default:; run pf $w --steps 400 default:
done
run base syn-nd24k --symmetric --steps 400 default:; run pf syn-nd24k --symmetric --steps 400 default:
cat $O/pass_prefetch.md

#!/bin/bash
# Round 4, 25th GPU call (the budget's last two minutes): the literal leg with samblaster's two stages, its stage log.
out=$PWD/gpurun_out; mkdir -p $out
SSG_SBL_LOG=1 timeout 130 python tools/dbg/literal_ab.py --out $out/r04y_literal.json t16:t=16:SSG_SBL_LOG=1 > $out/r04y_literal.log 2>&1
grep -E "config|samblaster|\[bwa\]|16075" $out/r04y_literal.log | cut -c1-330

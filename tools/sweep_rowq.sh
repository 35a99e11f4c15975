#!/bin/bash
# A/B of the quantiser kernel families inside the real sample-step and stand-alone (tools only; run on the GPU box)
out=${1:-gpurun_out/rowq_sweep}
mkdir -p $out
python tools/microbench_rowq.py > $out/mb_default.txt 2>&1
OQ_ROWQ_FWD_NW=2 OQ_ROWQ_BWD_NW=2 python tools/microbench_rowq.py > $out/mb_nw2.txt 2>&1
B="python bench.py --steps 256 --warmup 32 --no-cpu-baseline --no-end-to-end"
OQ_ROWQ=0 $B > $out/b_old.json 2>/dev/null
$B > $out/b_all.json 2>/dev/null
OQ_ROWQ_FWD_LET=0 OQ_ROWQ_BWD_LET=0 $B > $out/b_nolet.json 2>/dev/null
OQ_ROWQ_BWD_LET=0 $B > $out/b_nobwdlet.json 2>/dev/null
OQ_ROWQ_BWD_LET=0 OQ_ROWQ_FWD_NW=2 $B > $out/b_nobwdlet_fnw2.json 2>/dev/null
OQ_ROWQ_BWD_LET=0 OQ_ROWQ_FWD_LET=0 OQ_ROWQ_FWD_NW=2 OQ_ROWQ_BWD_NW=2 $B > $out/b_nolet_nw2.json 2>/dev/null
for f in $out/b_*.json; do echo "$f $(python -c "import json;d=json.load(open('$f'));print(round(d['value'],1), round(d['ms_per_step'],3))")"; done
grep -h "fwd\|per 7B" $out/mb_default.txt; echo NW2; grep -h "fwd\|per 7B" $out/mb_nw2.txt

#!/bin/bash
# A/B of the quantiser kernel families inside the real sample-step and stand-alone (tools only; run on the GPU box)
out=${1:-gpurun_out/rowq_sweep}
mkdir -p $out
for nw in 0 2 4 8; do OQ_ROWQ_FWD_NW=$nw OQ_ROWQ_BWD_NW=$nw python tools/microbench_rowq.py > $out/mb_nw$nw.txt 2>&1; done
B="python bench.py --steps 256 --warmup 32 --no-cpu-baseline --no-end-to-end"
OQ_ROWQ=0 $B > $out/b_old.json 2>/dev/null
for nw in 0 2 4; do OQ_ROWQ_FWD_NW=$nw OQ_ROWQ_BWD_NW=$nw $B > $out/b_nw$nw.json 2>/dev/null; done
OQ_ROWQ=0 $B > $out/b_old2.json 2>/dev/null
OQ_ROWQ_FWD_NW=2 OQ_ROWQ_BWD_NW=2 $B > $out/b_nw2b.json 2>/dev/null
for f in $out/b_*.json; do echo "$f $(python -c "import json;d=json.load(open('$f'));print(round(d['value'],1), round(d['ms_per_step'],3))")"; done
for nw in 0 2 4 8; do echo NW$nw; grep -h "fwd\|per 7B" $out/mb_nw$nw.txt; done

#!/bin/bash
# Same-box A/B of an environment switch on one bench config: tools/ab_cfg.sh CONFIG NAME VAL_A VAL_B [rounds]
cfg=$1; name=$2; a=$3; b=$4; rounds=${5:-2}
for r in $(seq $rounds); do
  for v in $a $b; do
    val=$(env $name=$v python bench.py --config $cfg --steps 512 --warmup 128 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "import sys,json; print(round(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['value'],1))")
    echo "$cfg $name=$v: $val"
  done
done

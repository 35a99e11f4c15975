#!/bin/bash
# One hipGraph-replayed sample-step cut out of a rocprofv3 kernel trace of bench.py (run on the GPU box).
# usage: bash tools/step_breakdown.sh [out_dir] [extra bench args...]
set -o pipefail
out=${1:-gpurun_out/sb}; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-end-to-end "$@" > $out/prof.log 2>&1
python tools/step_profile.py $(find $out/prof -name "*kernel_trace.csv" | head -1) 80 > $out/step_breakdown.txt
cut -c1-150 $out/step_breakdown.txt

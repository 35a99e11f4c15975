#!/bin/bash
# Collects every number committed under profiles/ for this round (run on the GPU box: gpurun -- bash tools/collect_profiles.sh HEAD).
# rocprofv3 needs TMPDIR on a local disk; counters are collected in separate passes, never together with HIP/HSA traces.
set -o pipefail
HEAD=${1:-unknown}
out=gpurun_out/final
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > $out/r3_bench_default.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-end-to-end > $out/prof.log 2>&1
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/r3_bench_kernel_stats.csv
python tools/step_profile.py $(find $out/prof -name "*kernel_trace.csv" | head -1) 80 > $out/r3_step_breakdown.txt
rm -rf $out/prof
for c in llama-7b-w3a16g128 llama-2-13b-w4a4 llama-2-70b-w2a16g64; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$c -o bench -- python bench.py --config $c --steps 32 --warmup 8 --no-cpu-baseline --no-end-to-end > $out/prof_$c.log 2>&1
  python tools/step_profile.py $(find $out/prof_$c -name "*kernel_trace.csv" | head -1) 80 > $out/r3_step_breakdown_$c.txt
  rm -rf $out/prof_$c
done
P="python bench.py --steps 4 --warmup 1 --nsamples 4 --no-cpu-baseline --no-end-to-end"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/mfma -o p -- $P > $out/mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o p -- $P > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o p -- $P > $out/write.log 2>&1
M=$(find $out/mfma -name "*counter_collection.csv" | head -1); F=$(find $out/fetch -name "*counter_collection.csv" | head -1); W=$(find $out/write -name "*counter_collection.csv" | head -1)
python tools/gemm_counters.py llama-7b-w4a4 4 $HEAD $M $F $W > $out/r3_gemm_counters.json
python tools/pmc_traffic.py $F FETCH_SIZE 50 > $out/r3_pmc_fetch_by_kernel.txt
python tools/pmc_traffic.py $W WRITE_SIZE 50 > $out/r3_pmc_write_by_kernel.txt
python tools/pmc_summary.py $M "gemm_bf16|gemm_i8|attn" 4 > $out/r3_mfma_counters_by_kernel.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $out/sq -o p -- $P > $out/sq.log 2>&1
python tools/pmc_summary.py $(find $out/sq -name "*counter_collection.csv" | head -1) "fq_|gq_|letq|rowq|normq|ropeq|colreduce|gemm_bf16|gemm_i8|attn" 4 > $out/r3_sq_counters_by_kernel.txt
rm -rf $out/mfma $out/fetch $out/write $out/sq
OQ_MB_CODES=1 python tools/microbench_rowq.py > $out/r3_quant_microbench.txt 2>&1
python tools/microbench_gemm_i8.py > $out/r3_gemm_i8_microbench.txt 2>&1
python tools/microbench_wgrad.py > $out/r3_wgrad_microbench.txt 2>&1
for c in llama-7b-w4a4-aug llama-7b-w3a16g128 llama-2-13b-w4a4 llama-2-70b-w2a16g64 opt-125m-w4a16; do python bench.py --config $c --steps 256 --warmup 32 --no-cpu-baseline 2>/dev/null; done > $out/r3_bench_configs.jsonl
{
  for kv in "OQ_INT_FPROP 0 1" "OQ_GEMM_W256 0 1" "OQ_INT_PRE_F32_QKV 0 1" "OQ_INT_PRE_F32_MLP 0 1" "OQ_GRID_ATTN 0 1" "OQ_WIDE 0 1"; do
    set -- $kv
    for rep in 1 2; do for v in $2 $3; do
      echo -n "llama-7b-w4a4 $1=$v : "; env $1=$v python bench.py --steps 512 --warmup 128 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), 'sample-steps/s', round(d['ms_per_step'],4), 'ms')"
    done; done
  done
  for rep in 1 2; do for v in 0 1; do
    echo -n "llama-7b-w4a4 OQ_WIDE=$v OQ_GRID_ATTN=$v : "; OQ_WIDE=$v OQ_GRID_ATTN=$v python bench.py --steps 512 --warmup 128 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), 'sample-steps/s', round(d['ms_per_step'],4), 'ms')"
  done; done
  for c in llama-7b-w3a16g128 llama-2-70b-w2a16g64; do for v in 0 1; do
    echo -n "$c OQ_GROUPQ=$v : "; OQ_GROUPQ=$v python bench.py --config $c --steps 128 --warmup 32 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), 'sample-steps/s')"
  done; done
} > $out/r3_ab_same_box.txt 2>&1
python tools/blaslt_ceiling.py > $out/r3_gemm_vs_hipblaslt.txt 2>&1
python tools/soak.py 3 3 > $out/r3_soak.txt 2>&1
python tests/diag/forward_flips.py 2048 > $out/r3_forward_flips_t2048.txt 2>&1
ls -la $out | head -60
cut -c1-400 $out/r3_bench_default.json

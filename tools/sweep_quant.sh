echo "DEFAULT"; python tools/microbench_quant.py 2>&1 | grep -E "fwd L|bwd L|bwd act"
for ch in 2 4 8; do for b in 1024; do echo "CH=$ch BLOCKS=$b"; OQ_DBG_FQ_CH=$ch OQ_DBG_FQ_BLOCKS=$b python tools/microbench_quant.py 2>&1 | grep -E "fwd L|bwd L|bwd act"; done; done
echo "DEFAULT blocks 2048"; OQ_DBG_FQ_BLOCKS=2048 python tools/microbench_quant.py 2>&1 | grep -E "fwd L|bwd L|bwd act"

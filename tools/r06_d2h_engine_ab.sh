#!/bin/bash
# The stream's D2H copies run as ROCclr blit KERNELS (__amd_rocclr_copyBuffer, profiles/r06z_trace_e2e_gaps.log), the H2D copies on the SDMA
# engines.  Which runtime setting changes that, and what does an end-to-end pass take then?  bash tools/r06_d2h_engine_ab.sh  (GPU box)
OUT=gpurun_out/d2h_ab; mkdir -p $OUT
run() { echo "== $1"; env $1 python3 tools/trace_e2e_pass.py $2 2>&1 | grep "^pass" | awk '{print $3}' | tr '\n' ' '; echo; }
for rep in 1 2; do
run "MS_NOP=1"
run "GPU_BLIT_ENGINE_TYPE=2"
run "GPU_FORCE_BLIT_COPY_SIZE=0"
run "ROC_ENABLE_LARGE_BAR=0"
run "DEBUG_CLR_LIMIT_BLIT_WG=4"
run "DEBUG_CLR_LIMIT_BLIT_WG=64"
run "HSA_ENABLE_SDMA=0"
done
run "MS_NOP=1" cli
run "DEBUG_CLR_LIMIT_BLIT_WG=4" cli

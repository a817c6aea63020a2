#!/bin/bash
# TA / TCP / TCC counters of the fp64 stage (at most two counters of a block per pass: more "exceeds the capabilities of the hardware" and hangs the tool)
export MS_SYNTH_WORKERS=1; cd /tmp; export TMPDIR=/tmp; cd /root/repo
run() { timeout 90 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d gpurun_out/pmc_rs/$1 -- python3 tools/stats_probe.py > /dev/null 2> gpurun_out/pmc_rs_$1.err; }
rm -rf gpurun_out/pmc_rs; mkdir -p gpurun_out/pmc_rs
run a "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
run b "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
run c "TCC_HIT_sum TCC_MISS_sum"
run d "TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
python3 tools/pmc_summary.py gpurun_out/pmc_rs gpurun_out/pmc_rs.csv
grep "rescore\|prefilter" gpurun_out/pmc_rs.csv | sed 's/"ms::\([a-z_0-9]*\)[^"]*"/\1/'

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_e1/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_e1/bench_under_rocprof.json 2> gpurun_out/prof_e1/stats.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d gpurun_out/prof_e1/sq1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/prof_e1/sq1.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d gpurun_out/prof_e1/sq2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/prof_e1/sq2.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_e1/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/prof_e1/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_e1/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/prof_e1/write.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/prof_e1/lds -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/prof_e1/lds.err
find gpurun_out/prof_e1 -name "*.csv" | head -30
tail -3 gpurun_out/prof_e1/sq2.err

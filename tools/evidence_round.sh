# Round evidence on the GPU box: tests, bench lines, profiles.  Usage: bash tools/evidence_round.sh <tag>
TAG=${1:-r01}
OUT=gpurun_out/evidence_$TAG
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
if [ "$2" = "quick" ]; then QUICK=1; fi
python bench.py > $OUT/bench_c4shard.json 2> $OUT/bench_c4shard.err
if [ -z "$QUICK" ]; then
python bench.py --workload c3 > $OUT/bench_c3.json 2> /dev/null
python bench.py --workload c2 --steps 200 --warmup 20 > $OUT/bench_c2.json 2> /dev/null
python bench.py --workload c5shard --steps 4 --warmup 1 > $OUT/bench_c5shard.json 2> /dev/null
python tools/pf_clock.py 2>&1 | grep -v amdgpu.ids > $OUT/prefilter_clock.log
python tools/pf_uniform.py 2>&1 | grep -v amdgpu.ids > $OUT/prefilter_uniform_width.log
python tools/pf_variants.py c4shard 16:1 18:1 17:2 4:1 2>&1 | grep -v amdgpu.ids > $OUT/prefilter_variants.log
python tools/e2e_time.py 2>&1 | grep -v amdgpu.ids > $OUT/end_to_end_pcie.log
./tools/ubench/mfma_i8_rate > $OUT/mfma_i8_rate.log 2>&1
fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=$OUT/prof
mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $P/bench_under_rocprof.json 2> $P/stats.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $P/sq1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $P/sq1.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $P/sq2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $P/sq2.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $P/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $P/write.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $P/lds -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $P/lds.err
ls $OUT $P

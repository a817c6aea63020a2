# Round evidence on the GPU box: tests, bench lines, profiles.  Usage: bash tools/evidence_round.sh <tag> [quick]
TAG=${1:-r02}
OUT=gpurun_out/evidence_$TAG
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -n 2 $OUT/pytest_gpu.log
if [ "$2" = "quick" ]; then QUICK=1; fi
python bench.py > $OUT/bench_c4.json 2> $OUT/bench_c4.err
MS_BENCH_BACKEND=gloo MS_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 5 --no-cpu-baseline > $OUT/bench_c4_2ranks_one_gpu_gloo.json 2> $OUT/bench_c4_2ranks_one_gpu_gloo.log
if [ -z "$QUICK" ]; then
python bench.py --workload c3 --no-end-to-end > $OUT/bench_c3.json 2> /dev/null
python bench.py --workload c2 --steps 200 --warmup 20 --no-end-to-end > $OUT/bench_c2.json 2> /dev/null
python bench.py --workload c5shard --steps 4 --warmup 1 > $OUT/bench_c5shard.json 2> /dev/null
python bench.py --workload c5 --genome-mbp 3000 --steps 2 --warmup 1 --min-warm-seconds 0 > $OUT/bench_c5_3000mbp.json 2> $OUT/bench_c5.err
python tools/pf_clock.py 2>&1 | grep -v amdgpu.ids > $OUT/prefilter_clock.log
python tools/pf_uniform.py 2>&1 | grep -v amdgpu.ids > $OUT/prefilter_uniform_width.log
python tools/pf_variants.py c4shard 46:2 44:2 47:1 31:1 28:1 29:1 30:2 33:2 16:1 4:1 2>&1 | grep -v amdgpu.ids > $OUT/prefilter_variants.log
python tools/once_overlap.py 2>&1 | grep -v amdgpu.ids > $OUT/scan_once_overlap.log
python tools/n_fraction.py 2>&1 | grep -v amdgpu.ids > $OUT/n_fraction.log
./tools/ubench/mfma_i8_rate > $OUT/mfma_i8_rate.log 2>&1
./tools/ubench/mfma_f6_probe > $OUT/mfma_f6_probe.log 2>&1
timeout 60 ./tools/ubench/valu_rate.bin > $OUT/valu_rate.log 2>&1
timeout 60 ./tools/ubench/issue_model.bin > $OUT/issue_model.log 2>&1
timeout 100 ./tools/ubench/cumask_probe.bin > $OUT/cumask_probe.log 2>&1
timeout 60 ./tools/ubench/cu_share_probe.bin > $OUT/cu_share_probe.log 2>&1
python tools/h2d_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/h2d_probe.log
python tools/c2_scaling.py 2>&1 | grep -v amdgpu.ids > $OUT/c2_scaling.log
python bench.py --workload c5 --genome-mbp 3000 --steps 2 --warmup 1 --min-warm-seconds 0 > $OUT/bench_c5_3000mbp_again.json 2> /dev/null
python tests/fuzz_parity.py --cases 1500 --seed 20000 > $OUT/fuzz.log 2>&1
python tests/fuzz_parity.py --cases 300 --seed 30000 --sweep >> $OUT/fuzz.log 2>&1
fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MS_SYNTH_WORKERS=1      # no forked workers under rocprofv3 (a forked child once hung in the tool's signal handler: 46 minutes)
P=$OUT/prof
mkdir -p $P
B="python3 bench.py --no-cpu-baseline --no-end-to-end"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- $B --steps 5 --warmup 2 > $P/bench_under_rocprof.json 2> $P/stats.err
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $P/sq1 -- $B --steps 2 --warmup 1 --min-warm-seconds 0 > /dev/null 2> $P/sq1.err
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F8 SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $P/sq2 -- $B --steps 2 --warmup 1 --min-warm-seconds 0 > /dev/null 2> $P/sq2.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/fetch -- $B --steps 2 --warmup 1 --min-warm-seconds 0 > /dev/null 2> $P/fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/write -- $B --steps 2 --warmup 1 --min-warm-seconds 0 > /dev/null 2> $P/write.err
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $P/lds -- $B --steps 2 --warmup 1 --min-warm-seconds 0 > /dev/null 2> $P/lds.err
for q in sq1 sq2 fetch write lds; do python3 tools/pmc_summary.py $P/$q $OUT/pmc_$q.csv; done
cp $(ls $P/stats/*/*kernel_stats.csv | head -n 1) $OUT/kernel_stats_c4.csv
rm -rf $P/sq1 $P/sq2 $P/fetch $P/write $P/lds $P/stats
ls $OUT

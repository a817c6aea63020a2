# Round evidence on the GPU box: tests, bench lines, profiles.  Usage: bash tools/evidence_round.sh <tag> [quick]
# Everything lands in gpurun_out/evidence_<tag>/ (scratch); what is to be judged is copied into profiles/ afterwards (profiles/INDEX.md).
TAG=${1:-r06}
OUT=gpurun_out/evidence_$TAG
mkdir -p $OUT
T="timeout 900"
$T python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -n 2 $OUT/pytest_gpu.log
if [ "$2" = "quick" ]; then QUICK=1; fi
$T python bench.py > $OUT/bench_c4.json 2> $OUT/bench_c4.err
MS_BENCH_BACKEND=gloo MS_BENCH_SHARE_GPU=1 $T python bench.py --gpus 2 --steps 5 --no-cpu-baseline > $OUT/bench_c4_2ranks_one_gpu_gloo.json 2> $OUT/bench_c4_2ranks_one_gpu_gloo.log
MS_BENCH_FORCE_PG=1 $T python bench.py --steps 5 --no-cpu-baseline --no-api --no-scale-projection > $OUT/bench_c4_rccl_one_rank.json 2> $OUT/bench_c4_rccl_one_rank.log      # the N > 1 branches over a ONE-rank RCCL communicator
# the shape of the driver's 8-GPU SCALE run on this box's one GPU (gloo: RCCL refuses several ranks per device)
MS_BENCH_BACKEND=gloo MS_BENCH_SHARE_GPU=1 $T python bench.py --gpus 8 --regions-per-set 160000 --steps 3 > $OUT/bench_c4_8ranks_one_gpu_gloo.json 2> $OUT/bench_c4_8ranks_one_gpu_gloo.log
MS_BENCH_BACKEND=gloo MS_BENCH_SHARE_GPU=1 $T python bench.py --gpus 8 --workload c5 --genome-mbp 800 --steps 2 --warmup 1 --min-warm-seconds 0 > $OUT/bench_c5_8ranks_one_gpu_gloo.json 2> $OUT/bench_c5_8ranks_one_gpu_gloo.log
python bench.py --gpus 9 > $OUT/bench_refuses_9_ranks.log 2>&1; echo "exit code $?" >> $OUT/bench_refuses_9_ranks.log
$T python tools/api_time.py --resident > $OUT/api_time.json 2> /dev/null
if [ -z "$QUICK" ]; then
$T python bench.py --workload c3 --no-end-to-end --no-api > $OUT/bench_c3.json 2> /dev/null
$T python bench.py --workload c2 --steps 200 --warmup 20 --no-end-to-end --no-api > $OUT/bench_c2.json 2> /dev/null
$T python bench.py --workload c5shard --steps 4 --warmup 1 > $OUT/bench_c5shard.json 2> /dev/null
$T python bench.py --workload c5 --genome-mbp 3000 --steps 2 --warmup 1 --min-warm-seconds 0 > $OUT/bench_c5_3000mbp.json 2> $OUT/bench_c5.err
# side workloads: the reference's other CLI settings on the full configs[3] regions (p = 1e-2: on one GPU's shard of an 8-GPU run -- 3.6e8 hits per scan there)
$T python bench.py --no-cpu-baseline --no-end-to-end --p-value 1e-3 > $OUT/bench_c4_p1e-3.json 2> /dev/null
$T python bench.py --no-cpu-baseline --no-end-to-end --p-value 1e-5 > $OUT/bench_c4_p1e-5.json 2> /dev/null
$T python bench.py --no-cpu-baseline --no-end-to-end --p-value 1e-2 --regions-per-set 125000 --steps 3 > $OUT/bench_c4shard_p1e-2.json 2> /dev/null
$T python bench.py --no-cpu-baseline --no-end-to-end --strand + > $OUT/bench_c4_strand_plus.json 2> /dev/null
# the side set with a JASPAR-like information profile: candidates per hit, fp64-stage ms
$T python bench.py --no-cpu-baseline --no-end-to-end --no-api --motif-set lowinfo > $OUT/bench_c4_lowinfo.json 2> /dev/null
# end to end with other batch schedules (8 equal batches of 250k; batches of 62.5k ... 125k)
$T python bench.py --no-cpu-baseline --no-api --no-scale-projection --batch-regions 250000 --max-batch-regions 250000 --no-batch-ramp --steps 4 > $OUT/bench_c4_e2e_8_equal_batches.json 2> /dev/null
$T python bench.py --no-cpu-baseline --no-api --no-scale-projection --batch-regions 62500 --max-batch-regions 125000 --steps 4 > $OUT/bench_c4_e2e_small_batches.json 2> /dev/null
$T python bench.py --no-cpu-baseline --no-end-to-end --extra-widths 33,40 > $OUT/bench_c4_plus_w33_w40.json 2> /dev/null
# what the all-fp64 kernel costs (the fence of VERDICT r5 #8): two motifs the pre-filter cannot take, on one GPU's shard of an 8-GPU run
$T python bench.py --no-cpu-baseline --no-end-to-end --no-api --extra-widths 70,90 --regions-per-set 125000 --steps 3 > $OUT/bench_c4shard_plus_w70_w90.json 2> $OUT/bench_c4shard_plus_w70_w90.err
$T python tools/pf_account.py 1e-4 full 2>&1 | grep -v amdgpu.ids > $OUT/pf_account.log
$T python tools/pf_account.py 1e-3 2>&1 | grep -v amdgpu.ids >> $OUT/pf_account.log
$T python tools/pf_class_clock.py 3 1e-4 2>&1 | grep -v amdgpu.ids > $OUT/class_clock.log
$T python tools/e2e_bounds.py 2>&1 | grep -v amdgpu.ids > $OUT/e2e_bounds.log
for i in 1 2; do
$T python tools/ab_full.py 3 1e-4 2>&1 | grep Mbase | sed "s/^/this build: /" >> $OUT/full_size_stage_times.log
done
$T python tools/ab_full.py 1 1e-4 2>&1 | grep Mbase >> $OUT/full_size_stage_times.log
$T python tools/ab_full.py 3 1e-3 2>&1 | grep Mbase >> $OUT/full_size_stage_times.log
$T python tools/pf_many_motifs.py 2>&1 | grep -v amdgpu.ids > $OUT/many_motifs.log
$T python tools/c2_latency.py 2>&1 | grep -v amdgpu.ids > $OUT/c2_latency.log
timeout 120 ./tools/ubench/atomic_rate.bin > $OUT/atomic_rate.log 2>&1
timeout 120 ./tools/ubench/pair_probe.bin > $OUT/pair_probe.log 2>&1
$T python tests/fuzz_parity.py --cases 1500 --seed 60000 > $OUT/fuzz.log 2>&1
$T python tests/fuzz_parity.py --cases 300 --seed 70000 --sweep >> $OUT/fuzz.log 2>&1
fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MS_SYNTH_WORKERS=1      # no forked workers under rocprofv3 (a forked child once hung in the tool's signal handler: 46 minutes)
P=$OUT/prof
mkdir -p $P
B="python3 bench.py --no-cpu-baseline --no-end-to-end --no-api --no-scale-projection"      # (the projection adds shrunk launches: every average here is over full-size launches only)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- $B --steps 5 --warmup 2 > $P/bench_under_rocprof.json 2> $P/stats.err
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $P/sq1 -- $B --steps 2 --warmup 1 --min-warm-seconds 0 > /dev/null 2> $P/sq1.err
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F8 SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $P/sq2 -- $B --steps 2 --warmup 1 --min-warm-seconds 0 > /dev/null 2> $P/sq2.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/fetch -- $B --steps 2 --warmup 1 --min-warm-seconds 0 > /dev/null 2> $P/fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/write -- $B --steps 2 --warmup 1 --min-warm-seconds 0 > /dev/null 2> $P/write.err
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $P/lds -- $B --steps 2 --warmup 1 --min-warm-seconds 0 > /dev/null 2> $P/lds.err
for q in sq1 sq2 fetch write lds; do python3 tools/pmc_summary.py $P/$q $OUT/pmc_$q.csv; done
f=$(ls $P/stats/*/*kernel_stats.csv 2>/dev/null | head -n 1); if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats_c4.csv; fi
cp $P/bench_under_rocprof.json $OUT/bench_under_rocprof_c4.json
rm -rf $P
# the line once more, with the traffic of the passes above (the record is rewritten on the box for this run only; the committed one is
# rewritten from the copied-back summaries by tools/pmc_traffic_update.py)
python3 tools/pmc_traffic_update.py c4 $OUT/pmc_fetch.csv $OUT/pmc_write.csv "evidence round $TAG" > /dev/null
$T python bench.py > $OUT/bench_c4_with_traffic.json 2> /dev/null
# round 6: the floor table of the pre-filter, the asm variant beside the product, the 8-rank end-to-end leg on this one GPU
$T python tools/pf_floor.py 1e-4 3 > $OUT/pf_floor.log 2>&1
for i in 1 2; do
MS_LIB_VARIANT=asm $T python bench.py --no-cpu-baseline --no-end-to-end --no-api --no-scale-projection > $OUT/bench_c4_asm_variant_$i.json 2> /dev/null
$T python bench.py --no-cpu-baseline --no-end-to-end --no-api --no-scale-projection > $OUT/bench_c4_product_same_box_$i.json 2> /dev/null
done
timeout 300 ./tools/ubench/shape_probe16.bin > $OUT/shape_probe16.log 2>&1
$T python tools/e2e_cli_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/e2e_cli_probe.log
bash tools/r06_ranks_one_gpu.sh evidence_$TAG/ranks > $OUT/ranks_one_gpu_summary.log 2>&1
ls $OUT

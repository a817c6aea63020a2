# Short refresh of the evidence a kernel-source change invalidates (the traffic record is stamped with the source hash): GPU suite, the
# driver-style line, rocprofv3 kernel stats and the two HBM-traffic PMC passes.  Usage: bash tools/evidence_refresh.sh <tag>
TAG=${1:-r05h}
OUT=gpurun_out/evidence_$TAG
mkdir -p $OUT
T="timeout 900"
$T python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -n 2 $OUT/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MS_SYNTH_WORKERS=1
P=$OUT/prof
mkdir -p $P
B="python3 bench.py --no-cpu-baseline --no-end-to-end --no-api --no-scale-projection"      # (the projection adds shrunk launches: every average here is over full-size launches only)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- $B --steps 5 --warmup 2 > $P/bench_under_rocprof.json 2> $P/stats.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/fetch -- $B --steps 2 --warmup 1 --min-warm-seconds 0 > /dev/null 2> $P/fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/write -- $B --steps 2 --warmup 1 --min-warm-seconds 0 > /dev/null 2> $P/write.err
for q in fetch write; do python3 tools/pmc_summary.py $P/$q $OUT/pmc_$q.csv; done
f=$(ls $P/stats/*/*kernel_stats.csv 2>/dev/null | head -n 1); if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats_c4.csv; fi
cp $P/bench_under_rocprof.json $OUT/bench_under_rocprof_c4.json
rm -rf $P
# the line itself, with the traffic of the passes above (the record is rewritten on the box for this run only; the committed one is
# rewritten from the copied-back summaries by tools/pmc_traffic_update.py)
python3 tools/pmc_traffic_update.py c4 $OUT/pmc_fetch.csv $OUT/pmc_write.csv "evidence refresh $TAG" > /dev/null
$T python bench.py > $OUT/bench_c4.json 2> $OUT/bench_c4.err
ls $OUT

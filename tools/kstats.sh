# Per-kernel device time of the default bench step (rocprofv3 --kernel-trace --stats).  Usage on the GPU box: bash tools/kstats.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/kstats
rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $D/bench.json 2> $D/err.log
python3 - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/kstats/*/*_kernel_stats.csv")[0]
for row in list(csv.DictReader(open(f)))[:12]:
    print(f"{row['Name'][:70]:70s} calls {row['Calls']:>4s} avg {float(row['AverageNs'])/1e3:9.1f} us  {row['Percentage']:>6s} %")
d = json.load(open("gpurun_out/kstats/bench.json"))
print("value %.3e  ms/step %.2f" % (d["value"], d["ms_per_step"]), d["stage_ms_per_scan"])
PY

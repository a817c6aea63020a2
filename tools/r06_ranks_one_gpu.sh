# Round 6 (VERDICT r5 #3a): the END-TO-END stream leg as 8 ranks sharing the box's one GPU (gloo): what 24 stream threads + 8 pinned pools do to one host.
# Usage: bash tools/r06_ranks_one_gpu.sh <tag>
TAG=${1:-r06c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
T="timeout 900"
E="--gpus 8 --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-scale-projection"
export MS_BENCH_BACKEND=gloo MS_BENCH_SHARE_GPU=1
lscpu | grep -i "numa\|socket\|model name\|^CPU(s)" > $OUT/lscpu.txt
cat /sys/bus/pci/devices/*/numa_node 2>/dev/null | sort | uniq -c > $OUT/pci_numa_nodes.txt
$T python bench.py $E > $OUT/e2e_8ranks_one_gpu.json 2> $OUT/e2e_8ranks_one_gpu.err
MS_NUMA_BIND=1 $T python bench.py $E > $OUT/e2e_8ranks_one_gpu_numa_bound.json 2> /dev/null
MS_NUMA_BIND=0 $T python bench.py $E --host-pack > $OUT/e2e_8ranks_one_gpu_host_pack.json 2> /dev/null
MS_NUMA_BIND=1 $T python bench.py $E --host-pack > $OUT/e2e_8ranks_one_gpu_host_pack_numa_bound.json 2> /dev/null
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/e2e_8ranks*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith('{"metric"')][-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    e = j["value_end_to_end"]; pr = e["per_rank"]
    print(os.path.basename(f), "whole-job e2e", f"{j['value_8d_end_to_end']:.3e}", "cli", f"{j['value_8d_cli_job']:.3e}", "resident", f"{j['value']:.3e}")
    print("   ms/pass by rank", pr["ms_per_pass_pipelined"], "cpu ms/pass", pr["cpu_ms_per_pass_pipelined"], "box busy", round(pr["host_cpu_busy_fraction_of_box"], 3))
    print("   pinned alloc+fill ms", pr["pinned_input_alloc_and_fill_ms"], "numa", pr["numa_node_bound"], "cpus allowed", pr["cpus_allowed"])
    print("   stage", {k: v for k, v in pr["stage_ms_last_pass_by_rank"].items()})
PY

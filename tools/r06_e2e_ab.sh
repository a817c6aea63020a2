# Round 6, end-to-end A/B on one box: the stream that packs on the scan stage + 12-byte copy-out (this build) against round 5's form
# (MS_MEASURE=1 MS_STREAM_PACK_IN_UPLOAD=1), batch schedules, and the asm / intrinsic-only pre-filter side by side.
# Usage: bash tools/r06_e2e_ab.sh <tag>      -> gpurun_out/<tag>/
TAG=${1:-r06b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
T="timeout 600"
E="--no-cpu-baseline --no-api --no-scale-projection --steps 4 --warmup 2"
$T python bench.py $E > $OUT/e2e_new.json 2> /dev/null
MS_MEASURE=1 MS_STREAM_PACK_IN_UPLOAD=1 $T python bench.py $E > $OUT/e2e_pack_in_upload.json 2> /dev/null
$T python bench.py $E > $OUT/e2e_new_2.json 2> /dev/null
$T python bench.py $E --batch-regions 250000 --max-batch-regions 500000 > $OUT/e2e_new_b250k_m500k.json 2> /dev/null
$T python bench.py $E --batch-regions 125000 --max-batch-regions 500000 > $OUT/e2e_new_b125k_m500k.json 2> /dev/null
$T python bench.py $E --batch-regions 250000 --max-batch-regions 250000 > $OUT/e2e_new_b250k_m250k.json 2> /dev/null
R="--no-cpu-baseline --no-end-to-end --no-api --no-scale-projection"
for i in 1 2; do
$T python bench.py $R > $OUT/resident_asm_$i.json 2> /dev/null
MS_LIB_VARIANT=noasm $T python bench.py $R > $OUT/resident_noasm_$i.json 2> /dev/null
done
$T python bench.py $R --p-value 1e-3 > $OUT/resident_asm_p1e-3.json 2> /dev/null
MS_LIB_VARIANT=noasm $T python bench.py $R --p-value 1e-3 > $OUT/resident_noasm_p1e-3.json 2> /dev/null
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith('{"metric"')][-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    e2e = j.get("value_end_to_end")
    line = f"{os.path.basename(f):36s} value {j['value']:.4e}  step {j['ms_per_step']:.2f} ms  kernel {j['roofline']['kernel_ms']:.2f} ms"
    if e2e:
        m = e2e["ms_per_pass"]
        line += "  | e2e ms/pass " + " ".join(f"{k}={v:.1f}" for k, v in m.items())
        st = e2e.get("stage_ms_last_pass", {}).get("pipelined", {})
        line += "  | stages " + " ".join(f"{k}:{v.get('ms_work')}/{v.get('ms_wait_in')}" for k, v in st.items() if isinstance(v, dict) and 'ms_work' in v)
    print(line)
PY

#!/bin/bash
# The bench's end-to-end legs N times on one box: how steady are the timed passes?  bash tools/r06_bench_e2e_repeat.sh [n]  (GPU box)
for rep in $(seq 1 ${1:-6}); do
  python bench.py --steps 6 --no-cpu-baseline --no-api --no-scale-projection 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); e=d['value_end_to_end']
        print('value %.3e  e2e %.3f cli %.3f sustained %.3f | passes' % (d['value'], d['end_to_end_over_resident']['pipelined'], d['end_to_end_over_resident']['pipelined_cli'], d['end_to_end_over_resident']['pipelined_sustained']), e['ms_each_pass']['pipelined'], e['ms_each_pass']['pipelined_cli'], e['ms_each_pass'].get('pipelined_16B'))
"
done

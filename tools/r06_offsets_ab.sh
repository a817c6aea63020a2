#!/bin/bash
# A queued scan's per-motif offsets: copied in stream order into pinned words (the product) against round 5's blocking hipMemcpy in scan_complete
# (MS_MEASURE=1 MS_OFFSETS_BLOCKING=1) -- the bench's end-to-end legs, alternating on one box.  bash tools/r06_offsets_ab.sh  (GPU box)
for rep in 1 2 3; do
for m in 0 1; do
  if [ $m = 1 ]; then export MS_MEASURE=1 MS_OFFSETS_BLOCKING=1; else unset MS_MEASURE MS_OFFSETS_BLOCKING; fi
  python bench.py --steps 6 --no-cpu-baseline --no-api --no-scale-projection 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); e=d['value_end_to_end']
        print('blocking=$m value %.3e  e2e %.3f cli %.3f sustained %.3f | passes' % (d['value'], d['end_to_end_over_resident']['pipelined'], d['end_to_end_over_resident']['pipelined_cli'], d['end_to_end_over_resident']['pipelined_sustained']), e['ms_each_pass']['pipelined'], e['ms_each_pass']['pipelined_cli'])
"
done
done

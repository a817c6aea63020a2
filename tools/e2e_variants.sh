#!/bin/bash
# end-to-end leg of the bench under different batch schedules / the CU partition: bash tools/e2e_variants.sh
run() { python bench.py --no-cpu-baseline $1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); e=d['value_end_to_end']
print('$2 $1', '| value %.4g'%d['value'], '| e2e %.4g'%e['pipelined'], 'ms %.1f'%e['ms_per_pass']['pipelined'], 'ratio %.3f'%(e['pipelined']/d['value']), len(e['batch_sizes']), 'batches', {k:v['ms_work'] for k,v in e['stage_ms_last_pass']['pipelined'].items() if isinstance(v,dict) and 'ms_work' in v}, e['stage_ms_last_pass']['pipelined']['scan_device_ms'])
"; }
run "" default
MS_MEASURE=1 MS_CU_PARTITION=1 run "" "CU partition:"
run "--max-batch-regions 500000" ""

for args in "" "--max-batch-regions 500000" "--batch-regions 250000 --max-batch-regions 500000" "--batch-regions 62500 --max-batch-regions 250000"; do
python bench.py --no-cpu-baseline $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); e=d['value_end_to_end']
print('$args', '| value %.4g'%d['value'], '| e2e %.4g'%e['pipelined'], 'ms %.1f'%e['ms_per_pass']['pipelined'], 'ratio %.3f'%(e['pipelined']/d['value']), e['batch_sizes'], {k:v['ms_work'] for k,v in e['stage_ms_last_pass']['pipelined'].items() if isinstance(v,dict) and 'ms_work' in v})
"
done

#!/bin/bash
# checkpoint: full GPU suite, smoke(), the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/box_log.sh > /dev/null 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $R/gpurun_out/r06_gpu_suite_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 600 python bench.py 2>/dev/null | grep '^{"metric' > $R/gpurun_out/r06_bench_line_final.json ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_line_final.json').read())
r=d['roofline']
print('value',d['value'],'ms',d['ms_per_step'],'h2d',d['value_with_per_step_h2d'],'epoch',d['epoch_ms'],d['epoch_train_ms'],d['epoch_eval_test_ms'])
print('roofline',r['kernel'][:60],r['frac'],r['avg_us'],r.get('traffic'))
print('riders',d['instep_kernels'].get('adamw_riders'))
print('adamw', [ (x['kernel'][:30], x['frac'], x.get('parameters_swept_per_step'), x.get('parameters_updated_by_riders_per_step')) for x in d['roofline_trace'] if 'adamw' in x['kernel']])
for s_ in d['secondary']: print(s_['metric'], s_['value'], s_['roofline']['frac'])
PY

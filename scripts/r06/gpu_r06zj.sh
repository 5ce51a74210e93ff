#!/bin/bash
# the default bench line (with the rider-free fields) + XLNet / C5 lines
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/box_log.sh > /dev/null 2>&1
( time timeout 600 python bench.py 2>/dev/null | grep '^{"metric' > $R/gpurun_out/r06_bench_line_final.json ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_line_final.json').read())
r=d['roofline']
print('value',d['value'],'ms',d['ms_per_step'],'h2d',d['value_with_per_step_h2d'],'epoch',d['epoch_ms'],d['epoch_train_ms'],d['epoch_eval_test_ms'])
print('roofline',r['kernel'][:60],r['frac'],r['avg_us'],r.get('rider_free_avg_us'),r.get('rider_free_frac'),r.get('traffic'))
for s_ in d['secondary']: print(s_['metric'], s_['value'], s_['roofline']['frac'], s_['roofline'].get('rider_free_frac'))
PY

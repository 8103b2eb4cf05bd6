#!/bin/bash
# Socket power and shader clock (rocm-smi, 5 samples/s) while bench.py runs a long timed region.
#   gpurun -- 'bash tools/power_trace.sh f16x3 600'
prec=${1:-f16x3}; steps=${2:-600}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
( for i in $(seq 1 60); do rocm-smi --showpower --showclocks --json 2>/dev/null | python -c "
import sys, json, time
try:
    d = json.load(sys.stdin)['card0']
    pw = [v for k, v in d.items() if 'ower' in k and 'W' in k]
    sclk = [v for k, v in d.items() if 'sclk' in k]
    mclk = [v for k, v in d.items() if 'mclk' in k]
    print(round(time.time(), 2), 'power', pw, 'sclk', sclk, 'mclk', mclk, flush=True)
except Exception as e:
    print('smi parse error', e, flush=True)
"; sleep 0.1; done ) > /tmp/smi.log 2>&1 &
smi=$!
sleep 1.0
python bench.py --no-cpu-baseline --precision $prec --steps $steps --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['config']['precision'], 'ms/step', round(d['ms_per_step'], 3))"
kill $smi 2>/dev/null
cat /tmp/smi.log

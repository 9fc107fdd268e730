#!/bin/bash
# Samples power / shader clock / temperature (rocm-smi) while the benchmark iteration runs: is the compositing clock
# (1.8-2.0 GHz measured inside the kernels, 2.4 GHz nominal) a power cap?   bash tools/power_clock.sh [bench args]
( for i in $(seq 60); do rocm-smi --showpower --showclocks --showtemp --showperflevel --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done ) > gpurun_out/power_samples.jsonl &
SM=$!
sleep 1.5
python bench.py --steps 3000 --warmup 20 --no-cpu-baseline --no-substep "$@" 2>&1 | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('bench', d['value'],d['ms_per_step'])"
wait $SM
python - <<'PY'
import json
rows=[]
for l in open('gpurun_out/power_samples.jsonl'):
    l=l.strip()
    if not l.startswith('{'): continue
    try: d=json.loads(l)
    except Exception: continue
    c=d.get('card0',{})
    rows.append({k:v for k,v in c.items() if any(s in k.lower() for s in ('power','sclk','mclk','fclk','temperature (sensor junction)','performance'))})
for i,r in enumerate(rows):
    if i%4==0: print(i*0.25, r)
PY
rocm-smi --showmaxpower 2>/dev/null | grep -i "max\|power" | head -5

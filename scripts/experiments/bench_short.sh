#!/bin/bash
# prints ms/step, G edges/s and per-class kernel ms (serial classes) for quick A/B on the GPU box
for mode in 1 0; do
LFR_SERIAL_CLASSES=$mode python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('serial=$mode ms/step %.3f  Gedges/s %.3f  kernels %.3f' % (d['ms_per_step'], d['value']/1e9, d['all_kernels_ms']))
for k,v in d['class_ms'].items():
    e=d['class_edges'][k]
    if e: print('   %-28s %8.3f ms  %9d edges  %.2f Gedges/s' % (k, v, e, e/v/1e6))
"
done

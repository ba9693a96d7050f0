#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for kv in VITS_X=0 VITS_NO_RBBLOCK32=1; do
  env $kv python bench.py --no-cpu-baseline --no-sub-results --no-extra-passes --steps 10 --warmup 3 > /tmp/b.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('/tmp/b.json')); print('$kv', round(d['ms_per_step'],3), 'ms instrumented', round(d['value']/1e6,2), 'M/s', 'plain', round(d.get('ms_per_step_without_kernel_events',0),3))
for k in d['top_kernels']:
    if k['kernel'].startswith('k3|') and ('f32' in k['kernel'] or 'f64' in k['kernel'] or '|b' in k['kernel']): print('    ', k['kernel'], round(k['ms_per_step'],3), round(k['tflops']/157.3,3))"
done; done

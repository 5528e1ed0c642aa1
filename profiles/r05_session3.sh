#!/bin/bash
# Round 5, GPU session 3: the whole -m gpu suite on the tree after the fixes (nz = 16 carve; one feature set; safe compiler flags;
# second opinion on the device; one family of compiled kernels), the default bench line, cycle stamps of the headline kernel.
set -o pipefail
mkdir -p gpurun_out/r05_s3
( timeout -k 10 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r05_s3/pytest_gpu.txt 2>&1
rc=$?; echo "pytest gpu: $rc"; tail -15 gpurun_out/r05_s3/pytest_gpu.txt
[ $rc -eq 0 ] || exit $rc
( timeout -k 10 900 python bench.py ) > gpurun_out/r05_s3/bench.json 2> gpurun_out/r05_s3/bench.err
echo "bench: $?"; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_s3/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', d['roofline']['frac'], 'nodes', d['nodes'])
print('parity_flags', d['parity_flags'])
for k, v in d.get('frontiers', {}).items():
    if isinstance(v, dict): print(' ', k, v.get('qp_per_s'), v.get('not_converged'))
for k, v in d.get('other_configs', {}).items():
    if isinstance(v, dict): print(' ', k, v.get('qp_per_s'), v.get('not_converged'), v.get('statuses_equal'), v.get('speedup'))
for k, v in d.get('mpc_steps_per_sec', {}).items():
    if isinstance(v, dict): print(' ', k, v.get('value'))
PY
( DBG_B=4096 timeout -k 10 300 python tests/gpu_dev_stamps_sized.py ) > gpurun_out/r05_s3/stamps.txt 2>&1
echo "stamps: $?"; grep -v "^hip ph" gpurun_out/r05_s3/stamps.txt | tail -40

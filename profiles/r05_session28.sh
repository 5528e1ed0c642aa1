#!/bin/bash
# Round 5, GPU session 28 (the tree with the LDS-staged shift kernel and the pinned search counts): the whole -m gpu suite, the default
# bench line, kernel stats and HBM counters of the shift kernels (separate --pmc passes, as the pool prescribes).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r05_s28; mkdir -p $O
( timeout -k 10 1700 python -m pytest tests -m gpu -x -q --timeout=600 ) > $O/pytest_gpu.txt 2>&1 &
PID=$!
while kill -0 $PID 2>/dev/null; do sleep 60; echo "pytest running: $(tail -c 120 $O/pytest_gpu.txt | tr '\n' ' ')"; done
wait $PID; rc=$?; echo "pytest gpu: $rc"; tail -6 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit $rc
( timeout -k 10 900 python bench.py ) > $O/bench.json 2> $O/bench.err
echo "bench: $?"; python - <<'PY'
import json, os
d = json.loads(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r05_s28/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', d['roofline']['frac'], 'kinds', d['roofline']['kernel_kinds_1_2_4_waves'], 'ilp', d['roofline']['ilp_schedule_1_2_4_waves'])
print('parity_flags', d['parity_flags'])
print('shift', d.get('warm_start_shift'))
for k, v in d.get('mpc_steps_per_sec', {}).items():
    if isinstance(v, dict): print(' ', k, v.get('value'))
PY
export PYTHONPATH=$GRAFT_REPO_ROOT/warm-start-hybrid-mpc_amd:$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests HMPC_JIT_SELFCHECK=0
S="python3 tests/gpu_shift_time.py"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/shift_stats -- $S > $O/shift_stats.log 2>&1; echo "shift stats rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/shift_fetch -- $S > $O/shift_fetch.log 2>&1; echo "shift fetch rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/shift_write -- $S > $O/shift_write.log 2>&1; echo "shift write rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/shift_sq -- $S > $O/shift_sq.log 2>&1; echo "shift sq rc $?"
python3 - <<'PY'
import csv, glob, os
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r05_s28'
def rows(d, pat):
    for f in glob.glob(O + '/' + d + '/**/*' + pat, recursive=True):
        yield from csv.DictReader(open(f))
out = open(O + '/shift_summary.txt', 'w')
def say(s):
    print(s); out.write(s + '\n')
for r in rows('shift_stats', 'kernel_stats.csv'):
    if 'shift' in r['Name']: say('stats  %-60s calls %s avg %.1f us min %.1f max %.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
for d, c in (('shift_fetch', 'FETCH_SIZE'), ('shift_write', 'WRITE_SIZE')):
    acc = {}
    for r in rows(d, 'counter_collection.csv'):
        if 'shift_row' in r['Kernel_Name'] and r['Counter_Name'] == c:
            acc.setdefault(r['Dispatch_Id'], 0.0)
            acc[r['Dispatch_Id']] += float(r['Counter_Value'])
    v = sorted(acc.values())
    say('%s per hmpc_shift_row_kernel dispatch (raw counter units, KB as rocprofv3 reports them): n %d, the three sizes (4096 / 65 536 / 262 144 leaves): min %.0f median %.0f max %.0f' % (c, len(v), v[0] if v else 0, v[len(v) // 2] if v else 0, v[-1] if v else 0))
for r in rows('shift_sq', 'counter_collection.csv'):
    pass
acc = {}
for r in rows('shift_sq', 'counter_collection.csv'):
    if 'shift_row' in r['Kernel_Name']:
        acc.setdefault(r['Counter_Name'], 0.0)
        acc[r['Counter_Name']] += float(r['Counter_Value'])
say('SQ counters summed over the hmpc_shift_row_kernel dispatches: %s' % {k: '%.3g' % v for k, v in acc.items()})
if acc.get('SQ_WAVE_CYCLES'): say('wait %.1f %% of wave cycles, instructions active %.1f %%' % (100 * acc.get('SQ_WAIT_ANY', 0) / acc['SQ_WAVE_CYCLES'], 100 * acc.get('SQ_ACTIVE_INST_ANY', 0) / acc['SQ_WAVE_CYCLES']))
PY
rm -rf $O/shift_stats $O/shift_fetch $O/shift_write $O/shift_sq

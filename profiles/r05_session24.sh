#!/bin/bash
# Round 5, session 24: kernel trace of the fleet of 1024 loops (what a warm step's launches last, and the gaps between them)
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
export PYTHONPATH=$R/warm-start-hybrid-mpc_amd:$R:$R/tests
O=$R/gpurun_out/r05_s24; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tests/gpu_dev_fleet_steps.py 1024 > $O/run.txt 2>&1; echo "rc $?"
grep -v amdgpu.ids $O/run.txt | tail -4 | cut -c1-200
F=$(find $O/trace -name "*kernel_trace.csv" | head -1); echo $F; python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
out = open(sys.argv[1].rsplit('/', 1)[0] + '/../../kernels_in_order.txt', 'w')
t0 = int(rows[0]['Start_Timestamp'])
prev_end = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.write('%10.3f ms  +%8.3f  dur %8.3f ms  grid %6s wg %4s  %s\n' % ((s - t0) / 1e6, 0 if prev_end is None else (s - prev_end) / 1e6, (e - s) / 1e6, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')), r['Kernel_Name'][:60]))
    prev_end = e
print(len(rows), 'kernels')
PY
rm -rf $O/trace

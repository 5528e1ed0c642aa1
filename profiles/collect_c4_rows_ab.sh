#!/bin/bash
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
B="python3 bench.py --workload random_mld --frontier 4096 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
for V in plain rows; do
  if [ $V = rows ]; then export HMPC_JIT_SIZED_ROWS=16; fi
  rm -rf $O/ab_${V}_*
  timeout -k 10 300 $B > $O/ab_${V}_bench.json 2> $O/ab_${V}_bench.err
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/ab_${V}_fetch -- $B > $O/ab_${V}_fetch.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/ab_${V}_write -- $B > $O/ab_${V}_write.log 2>&1
done
python3 - <<'PY'
import csv, glob, os
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out'
for v in ('plain', 'rows'):
    for c in ('fetch', 'write'):
        best = 0
        for f in glob.glob('%s/ab_%s_%s/**/*counter_collection.csv' % (O, v, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if 'hmpc_qp_kernel' in r['Kernel_Name']:
                    best = max(best, float(r['Counter_Value']))
        print(v, c, 'max per launch', best)
    print(v, open('%s/ab_%s_bench.json' % (O, v)).read()[-600:][:300])
PY

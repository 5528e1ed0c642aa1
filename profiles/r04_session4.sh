#!/bin/bash
# Round 4, GPU session 4: the two-launch form of the lazy terminal set -- parity test, then the distinct-states frontier
# with and without it, and the headline.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "two_launch or batch_position or arbitrary" 2>&1 | tail -12
for NS in 0 1; do
  if [ $NS = 1 ]; then export HMPC_NO_SPLIT=1; fi
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --states distinct > $O/split_distinct_$NS.json 2>> $O/split.err; echo "distinct nosplit=$NS rc $?"
  python -c "import json; d=json.loads(open('$O/split_distinct_$NS.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['nodes'])"
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/split_nominal_$NS.json 2>> $O/split.err; echo "nominal nosplit=$NS rc $?"
  python -c "import json; d=json.loads(open('$O/split_nominal_$NS.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['nodes'])"
done

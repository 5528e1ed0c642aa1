#!/bin/bash
# HBM traffic of the streaming kernel on BASELINE configs[4], 1024 distinct nodes (run through gpurun from the repo root):
#   bash profiles/collect_c4_traffic.sh <tag> <library name>
set -e -o pipefail
TAG=${1:-c4tr}; L=${2:-libhmpc.so}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
export HMPC_LIBRARY_NAME=$L DBG_PARITY=8
B="python3 tests/gpu_dev_cfg4.py"
rm -rf $O/${TAG}_fetch $O/${TAG}_write
timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -- $B > $O/${TAG}_fetch.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -- $B > $O/${TAG}_write.log 2>&1
python3 profiles/summarise.py pmc $O/${TAG}_traffic.json --grid 65536 $O/${TAG}_fetch $O/${TAG}_write > /dev/null
python3 -c "
import json; d=json.load(open('$O/${TAG}_traffic.json')); print('traffic per launch of 1024 nodes: %.1f MB (fetch %.1f, write %.1f)' % (d['hbm_traffic_bytes_per_launch']/1e6, d['FETCH_SIZE']['bytes_corrected']/1e6, d['WRITE_SIZE']['bytes_corrected']/1e6), d['_kernel']['scratch'])"
tail -1 $O/${TAG}_write.log

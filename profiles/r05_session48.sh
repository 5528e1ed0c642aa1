#!/bin/bash
# Round 5, GPU session 48: fourteen more random MLD shapes nobody has validated, on the final sources (default recipe, nets on)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s48; mkdir -p $O
( DBG_SHAPES="3,2,2,11,51;5,3,3,7,52;6,1,4,13,53;7,2,5,9,54;8,2,3,18,55;9,4,2,10,56;10,2,3,12,57;4,4,4,15,58;2,2,2,28,59;11,3,1,8,60;12,2,2,14,61;6,3,6,6,62;15,1,1,9,63;7,7,2,5,64" timeout -k 10 1100 python tests/gpu_sized_shapes.py ) > $O/sized_shapes_more.txt 2>&1 &
PID=$!
while kill -0 $PID 2>/dev/null; do sleep 60; echo "running: $(tail -c 160 $O/sized_shapes_more.txt | tr '\n' ' ')"; done
wait $PID; echo "shapes: $?"; cut -c1-230 $O/sized_shapes_more.txt | tail -18

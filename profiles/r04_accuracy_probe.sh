mkdir -p gpurun_out/r04
run() { echo "=== $LIB $@"; HMPC_LIBRARY_NAME=${LIB:-libhmpc.so} "$@" 2> /tmp/acc.err | grep RESULT; grep "^hip ph" /tmp/acc.err | awk '{printf "k%s:%s ", $5, $15} END {print ""}'; grep "^it " /tmp/acc.err | awk '{printf "o%s:%s ", $2, $12} END {print ""}'; }
run python tests/gpu_dev_accuracy.py c4 532
run python tests/gpu_dev_accuracy.py c4 1000
python tests/gpu_dev_accuracy2.py 532 14 | grep "after"
python tests/gpu_dev_escalation.py 2>&1 | grep -v "^it \|^hip ph\|polish round\|cert \|sigma " | head -30

#!/bin/bash
# Runs the given commands one after the other, each under its own time limit, and stops at the first one that ends
# abnormally (anything but exit code 0 or 1: a fault, an abort, a time limit).  usage: gpu_steps.sh "<limit> <cmd>" ...
for step in "$@"; do
    limit=${step%% *}; cmd=${step#* }
    echo "== [$limit s] $cmd"
    timeout -k 10 $limit bash -c "$cmd"
    rc=$?
    echo "== rc=$rc"
    if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "== abnormal end, stopping"; exit $rc; fi
done
exit 0

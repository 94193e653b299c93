#!/bin/bash
# gpurun with a record of WHAT ran: the commit and the uncommitted diff of the tree that is sent are kept beside the call's log (gpurun_out/<tag>_call.log,
# gpurun_out/<tag>_tree.txt).  Round 5 lost the cause of an abort on a work-in-progress kernel because the tree of that call was edited and never saved.
#   usage: scripts/gpu.sh <tag> [--timeout S] -- '<command>'
T=${1:?tag}; shift
{ echo "HEAD $(git rev-parse HEAD)"; git status --short; echo "---- git diff HEAD"; git diff HEAD; } > gpurun_out/${T}_tree.txt 2>&1
gpurun "$@" > gpurun_out/${T}_call.log 2>&1; rc=$?
tail -15 gpurun_out/${T}_call.log
exit $rc

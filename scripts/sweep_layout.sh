#!/bin/bash
# node-record order below the top levels (pt_set_option bfs_nodes: that many records breadth-first, whole levels; the rest depth-first) on the final kernel
run() { c=$1; shift; bash scripts/ab.sh -r 1 -t -c "$c" -f "$*" final; }
for c in C3 C4 C5; do
  run $c ""
  run $c --bfs-nodes 0
  run $c --bfs-nodes 204
  run $c --bfs-nodes 1024
done
run C3 --streams 1
run C3 --streams 1 --bfs-nodes 409
run C4 --streams 1
run C4 --streams 1 --bfs-nodes 409

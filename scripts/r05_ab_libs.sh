#!/bin/bash
# same-box A/B of library builds, alternating: scripts/r05_ab_libs.sh TAG "shapes" name:lib [name:lib ...]   (ROUNDS=3)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out/r05
tag=$1; shapes=$2; shift; shift
: > gpurun_out/r05/ab_$tag.txt
for round in $(seq 1 ${ROUNDS:-3}); do
  for spec in "$@"; do
    name=${spec%%:*}; lib=${spec#*:}
    GAT_LIBRARY=$REPO/$lib bash scripts/r05_quick.sh ab_$name $shapes | sed "s/^/$name round $round: /" | tee -a gpurun_out/r05/ab_$tag.txt
  done
done

#!/bin/bash
# usage: tools/ab.sh <tag> v1 v2 ...  -- per-kernel times of the cfg3 share and of cfg4 for library variants ("base" = in-tree) -> gpurun_out/<tag>_ab.txt
TAG=$1; shift
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
echo "== cfg3 share"; bash tools/var_run.sh "--config cfg3 --groups 125" "$@"
echo "== cfg4"; bash tools/var_run.sh "--config cfg4" "$@"
} > gpurun_out/${TAG}_ab.txt 2>&1
cat gpurun_out/${TAG}_ab.txt

#!/bin/bash
# usage: tools/mkvar.sh <name> [-DKNOB=value ...]  -- ipk_amd/_variants/v_<name>.so: the library with extra defines (for tools/var_run.sh, tools/ab.sh)
N=$1; shift
cd $(dirname $0)/..
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -fno-fast-math -pthread "$@" \
  -o ipk_amd/_variants/v_$N.so ipk_amd/csrc/db_merge.cpp ipk_amd/csrc/ipkgpu.hip ipk_amd/csrc/phylo_host.cpp ipk_amd/csrc/raxml_reader.cpp 2>&1 | grep -i "error" ; ls -la ipk_amd/_variants/v_$N.so

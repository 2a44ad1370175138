#!/bin/bash
# Dev aid: like build_dbg.sh for kernel files that need a definition of dmp::g_exact_fp32 (dbg_stub2.hip).
C=$(dirname $0)/../dualmessagepassing_amd/csrc; F=$1; M=$2; L=$3; shift 3
mkdir -p $(dirname $0)/_dbg
for k in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -D$M=$k -shared -o $(dirname $0)/_dbg/lib${L}_$k.so $C/$F $(dirname $0)/dbg_stub.hip $(dirname $0)/dbg_stub2.hip &
done
wait

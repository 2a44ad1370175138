#!/bin/bash
# Dev aid: standalone debug builds of one kernel file with a knob macro.  usage: build_dbg.sh <file.hip> <MACRO> <lib prefix> <knob>...
C=$(dirname $0)/../dualmessagepassing_amd/csrc; F=$1; M=$2; L=$3; shift 3
mkdir -p $(dirname $0)/_dbg
for k in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -D$M=$k -shared -o $(dirname $0)/_dbg/lib${L}_$k.so $C/$F $(dirname $0)/dbg_stub.hip &
done
wait

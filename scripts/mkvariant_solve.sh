#!/bin/bash
# usage: scripts/mkvariant_solve.sh NAME "<extra hipcc flags>"  -> lfr_amd/_variants/NAME.so: lfr_solve.hip rebuilt with the flags, every other
# object taken from csrc/_obj (run `python __graft_entry__.py` first) - a kernel variant in one compile instead of seven
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/local-feature-refinement_amd/csrc; O=$R/local-feature-refinement_amd/lfr_amd/_variants; mkdir -p $O /tmp/lfr_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I $R/include -I $C $2 -c $C/lfr_solve.hip -o /tmp/lfr_var/$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/_obj/lfr_wire.cpp.o $C/_obj/lfr_graph.cpp.o $C/_obj/lfr_treeplan.cpp.o $C/_obj/lfr_devctx.cpp.o \
  /tmp/lfr_var/$1.o $C/_obj/lfr_assemble.hip.o $C/_obj/lfr_graphstage.hip.o -o $O/$1.so
echo built $O/$1.so

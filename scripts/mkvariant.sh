#!/bin/bash
# usage: scripts/mkvariant.sh NAME "<extra hipcc flags>"  -> local-feature-refinement_amd/lfr_amd/_variants/NAME.so (built here, travels with gpurun)
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/local-feature-refinement_amd/csrc; O=$R/local-feature-refinement_amd/lfr_amd/_variants; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -I $R/include -I $C $2 \
  $C/lfr_wire.cpp $C/lfr_graph.cpp $C/lfr_treeplan.cpp $C/lfr_devctx.cpp $C/lfr_solve.hip $C/lfr_assemble.hip $C/lfr_graphstage.hip -o $O/$1.so
echo built $O/$1.so

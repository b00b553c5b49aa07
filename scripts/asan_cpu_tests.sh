#!/bin/bash
# AddressSanitizer over the host code (parser, graph stage on the host, size cap, device-context bookkeeping) with the CPU test
# suite: builds lfr_wire.cpp / lfr_graph.cpp / lfr_devctx.cpp with -fsanitize=address, links them with the regular device
# objects (csrc/_obj, run `python __graft_entry__.py` first) into /tmp/asan/liblfr_asan.so and runs `pytest -m "not gpu"` on it.
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/local-feature-refinement_amd/csrc; O=/tmp/asan; mkdir -p $O
H=/opt/rocm/bin/hipcc
for f in lfr_wire.cpp lfr_graph.cpp lfr_treeplan.cpp lfr_devctx.cpp; do
  $H --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-omit-frame-pointer -I $R/include -I $C -c $C/$f -o $O/$f.o
done
$H --offload-arch=gfx950 -shared -fPIC -fsanitize=address -shared-libsan $O/*.o $C/_obj/lfr_solve.hip.o $C/_obj/lfr_assemble.hip.o $C/_obj/lfr_graphstage.hip.o -o $O/liblfr_asan.so
ASAN=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd $R && LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 LFR_LIB_OVERRIDE=$O/liblfr_asan.so python -m pytest tests -x -q -m "not gpu" "$@"

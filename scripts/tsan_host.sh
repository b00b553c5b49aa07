#!/bin/bash
# ThreadSanitizer over the host code: the parallel whole-input scanner (speculative split, concurrent first-appearance map), the
# host graph stage and the size cap with its parallel halves.  Builds the three .cpp files with -fsanitize=thread, links them with
# the regular device objects (csrc/_obj) and scripts/probes/host_harness.cpp, and runs the harness on two generated inputs (short
# tracks with many small cuts; long tracks on few images = one giant component, deep recursion).  Runs here, no GPU.
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/local-feature-refinement_amd/csrc; O=/tmp/tsan; mkdir -p $O
H=/opt/rocm/bin/hipcc
for f in lfr_wire.cpp lfr_graph.cpp lfr_treeplan.cpp lfr_devctx.cpp; do
  $H --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=thread -I $R/include -I $C -c $C/$f -o $O/$f.o
done
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -fsanitize=thread -I $R/include -c $R/scripts/probes/host_harness.cpp -o $O/harness.o
$H --offload-arch=gfx950 -fsanitize=thread $O/harness.o $O/lfr_wire.cpp.o $O/lfr_graph.cpp.o $O/lfr_treeplan.cpp.o $O/lfr_devctx.cpp.o $C/_obj/lfr_solve.hip.o $C/_obj/lfr_assemble.hip.o $C/_obj/lfr_graphstage.hip.o -o $O/harness
python - <<PY
import sys
sys.path.insert(0, "$R/local-feature-refinement_amd")
from lfr_amd import capi, synthetic
capi.write_matching_file("$O/in.pb", synthetic.generate(seed=5, n_images=48, n_tracks=20000, len_dist="uniform", len_lo=2, len_hi=12, eps_out=0.02))
capi.write_matching_file("$O/in2.pb", synthetic.generate(seed=6, n_images=40, n_tracks=3000, len_dist="uniform", len_lo=20, len_hi=40, eps_out=0.02))
PY
for f in in.pb in2.pb; do
  LFR_HOST_THREADS=16 TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" $O/harness $O/$f 2>&1 | grep -c "WARNING: ThreadSanitizer" | sed "s/^/$f: ThreadSanitizer warnings: /"
done

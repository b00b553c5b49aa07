import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
from lfr_amd import capi
L = capi.lib()
t = time.perf_counter(); rc = L.lfr_hip_warmup(0); print("warmup #1 rc %d: %.1f ms" % (rc, (time.perf_counter() - t) * 1e3))
t = time.perf_counter(); rc = L.lfr_hip_warmup(0); print("warmup #2 rc %d: %.1f ms" % (rc, (time.perf_counter() - t) * 1e3))

"""End-to-end drop-in timing on the GPU box: config 4 -> .pb (native writer) -> `solve` CLI -> SolutionFile."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
from lfr_amd import capi, synthetic
t = time.time(); ma = synthetic.config4(); print("generate %.2f s" % (time.time() - t))
pb = "/tmp/config4.pb"
t = time.time(); capi.write_matching_file(pb, ma); print("write .pb %.2f s (%.0f MB)" % (time.time() - t, os.path.getsize(pb) / 1e6))
for rep in range(2):
    t = time.time()
    r = subprocess.run([os.path.join(ROOT, "multi-view-refinement/build/solve"), "--matches_file", pb, "--output_file", "/tmp/sol.pb"],
                       capture_output=True, text=True, env=dict(os.environ, LFR_VERBOSE=os.environ.get("LFR_VERBOSE", "1")))
    print("solve CLI wall %.2f s rc=%d" % (time.time() - t, r.returncode)); print(r.stdout.strip()); print(r.stderr.strip())

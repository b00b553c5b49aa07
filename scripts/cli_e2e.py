"""End-to-end drop-in timing on the GPU box: config 4 -> .pb (native writer) -> `solve` CLI -> SolutionFile.
Default launcher (one process) and LFR_DETACH_TEARDOWN=1 (the caller gets the exit code before the driver tears the process down)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
from lfr_amd import capi, synthetic
t = time.time(); ma = synthetic.config4(); print("generate %.2f s" % (time.time() - t))
pb = "/tmp/config4.pb"
t = time.time(); capi.write_matching_file(pb, ma); print("write .pb %.2f s (%.0f MB)" % (time.time() - t, os.path.getsize(pb) / 1e6))
for mode in ({}, {"LFR_DETACH_TEARDOWN": "1"}):
    for rep in range(3):
        t = time.time()
        r = subprocess.run([os.path.join(ROOT, "multi-view-refinement/build/solve"), "--matches_file", pb, "--output_file", "/tmp/sol.pb"],
                           capture_output=True, text=True, env=dict(os.environ, **mode))
        print("%s solve CLI wall %.3f s rc=%d" % (mode or "default (one process)", time.time() - t, r.returncode))
        if rep == 2:
            print(r.stdout.strip()); print(r.stderr.strip())
# back to back (a driver looping over scenes): the second call starts while nothing of the first is left (default) / while the first's
# detached child may still be tearing down (LFR_DETACH_TEARDOWN=1)
for mode in ({}, {"LFR_DETACH_TEARDOWN": "1"}):
    t = time.time()
    for rep in range(3):
        r = subprocess.run([os.path.join(ROOT, "multi-view-refinement/build/solve"), "--matches_file", pb, "--output_file", "/tmp/sol%d.pb" % rep],
                           capture_output=True, text=True, env=dict(os.environ, **mode))
        assert r.returncode == 0, r.stderr
    print("%s three calls back to back: %.3f s, outputs identical: %s" % (mode or "default", time.time() - t,
          open("/tmp/sol0.pb", "rb").read() == open("/tmp/sol2.pb", "rb").read()))

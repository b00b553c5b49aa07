import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for level in ("0", "1", "2"):
    for rep in range(2):
        t = time.time()
        r = subprocess.run([os.path.join(ROOT, "multi-view-refinement/build/solve"), "--matches_file", "/tmp/config4.pb", "--output_file", "/tmp/s.pb"],
                           capture_output=True, text=True, env=dict(os.environ, LFR_VERBOSE="1", LFR_WARMUP_LEVEL=level))
        w = time.time() - t
        keep = [l for l in (r.stdout + r.stderr).splitlines() if l.startswith(("Total", "Solver")) or "wall inside" in l]
        print("warm-up level %s: CLI wall %.2f s | %s" % (level, w, " | ".join(keep)))

"""Where does a kernel of lfr_solve.hip touch scratch (register spills)?  Lists the basic blocks of the chosen kernel that hold
scratch loads/stores with their loop depth and a few instruction counts that identify the region (runs here, no GPU).
usage: python scripts/spill_report.py [kernel-substring] [extra hipcc flags...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(ROOT, "local-feature-refinement_amd", "csrc")
want = sys.argv[1] if len(sys.argv) > 1 else "solve_block_kernelILb0ELi512"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-I", os.path.join(ROOT, "include"), "-I", C,
                       "-S", "--cuda-device-only", os.path.join(C, "lfr_solve.hip"), "-o", "/tmp/lfr_solve_spill.s"] + sys.argv[2:], stderr=subprocess.DEVNULL)
lines = open("/tmp/lfr_solve_spill.s").read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(want), l)][0]
end = [i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm")][0]
body = lines[start:end]
blk, info, order = "entry", {}, ["entry"]
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB[0-9_]+):(.*)", l)
    if m:
        blk = m.group(1); order.append(blk)
        d = re.search(r"Depth=(\d+)", l)
        info[blk] = dict(depth=int(d.group(1)) if d else 0, first=i)
    d = info.setdefault(blk, dict(depth=0, first=i))
    m2 = re.search(r"Depth=(\d+)", l)
    if m2 and l.strip().startswith(";"):
        d["depth"] = max(d["depth"], int(m2.group(1)))
    for key, pat in (("sld", "scratch_load"), ("sst", "scratch_store"), ("mfma", "v_mfma"), ("rl", "v_readlane"), ("ds", r"\bds_"), ("glob", "global_"), ("fma", r"v_fmac?_f64")):
        if re.search(pat, l): d[key] = d.get(key, 0) + 1
    d["n"] = d.get("n", 0) + 1
tot_l = tot_s = 0
print("%-14s %6s %5s %4s | %4s %4s | %4s %4s %4s %4s %4s" % ("block", "line", "depth", "n", "ld", "st", "mfma", "rl", "ds", "glob", "fma"))
for b in order:
    d = info[b]
    if d.get("sld") or d.get("sst"):
        tot_l += d.get("sld", 0); tot_s += d.get("sst", 0)
        print("%-14s %6d %5d %4d | %4d %4d | %4d %4d %4d %4d %4d" % (b, d["first"], d["depth"], d["n"], d.get("sld", 0), d.get("sst", 0), d.get("mfma", 0), d.get("rl", 0), d.get("ds", 0), d.get("glob", 0), d.get("fma", 0)))
print("total scratch loads %d stores %d in %d lines" % (tot_l, tot_s, len(body)))

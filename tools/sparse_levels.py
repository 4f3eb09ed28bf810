"""Joins a rocprofv3 kernel trace of a SPARSE_SCHUR bench run with the per-level plan statistics printed under CX_SPARSE_PLAN_STATS:
per level the duration of k_sp_diag / k_sp_panel / k_sp_update_slices of the LAST solve, beside targets, products and the longest chain."""
import csv, glob, re, sys
out = sys.argv[1]
stats = {}
for line in open(out + "/err.txt"):
    m = re.match(r"cxsp level (\d+) rows (\d+) panels (\d+) targets (\d+) products (\d+) longest (\d+)", line)
    if m:
        stats[int(m.group(1))] = tuple(int(x) for x in m.groups()[1:])
L = len(stats)
trace = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
diag = [r for r in rows if "k_sp_diag" in r["Kernel_Name"]][-L:]
t0, t1 = int(diag[0]["Start_Timestamp"]), None
seq = [r for r in rows if int(r["Start_Timestamp"]) >= t0 and re.search(r"k_sp_(diag|panel|update)", r["Kernel_Name"])]
level, per = -1, {}
for r in seq:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "k_sp_diag" in r["Kernel_Name"]:
        level += 1
        per[level] = {"diag": d, "panel": 0.0, "update": 0.0, "start": int(r["Start_Timestamp"]), "end": int(r["End_Timestamp"])}
    elif "k_sp_panel" in r["Kernel_Name"]:
        per[level]["panel"] = d
    else:
        per[level]["update"] = d
    per[level]["end"] = int(r["End_Timestamp"])
print("level rows panels targets products longest | diag_us panel_us update_us wall_us | us/product")
tot = {"diag": 0, "panel": 0, "update": 0, "wall": 0}
for l in range(L):
    s, p = stats[l], per.get(l)
    if p is None:
        continue
    nxt = per[l + 1]["start"] if l + 1 in per else p["end"]
    wall = (nxt - p["start"]) / 1e3
    for k in ("diag", "panel", "update"):
        tot[k] += p[k]
    tot["wall"] += wall
    print("%4d %6d %7d %7d %8d %5d | %7.1f %7.1f %8.1f %8.1f | %.4f" % (l, *s, p["diag"], p["panel"], p["update"], wall, p["update"] / max(1, s[3])))
print("total_ms diag %.2f panel %.2f update %.2f wall %.2f" % tuple(tot[k] / 1e3 for k in ("diag", "panel", "update", "wall")))

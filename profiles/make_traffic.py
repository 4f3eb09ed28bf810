#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md
prescribes) into per-kernel HBM bytes per launch:

    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024       # counters are in KB; FETCH_SIZE reports
                                                            # half of a wide coalesced stream on gfx950

usage: make_traffic.py <workload> <fetch_counter_collection.csv> <write_counter_collection.csv> <out.csv> <traffic.json>
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    depth, out = 0, []
    for ch in name:  # cut the argument list, keep template arguments
        if ch == "(" and depth == 0:
            break
        depth += ch == "<"
        depth -= ch == ">"
        out.append(ch)
    # the fp64 instantiations keep the names the library reports (k_chunk_pass<0>, not k_chunk_pass<0, double>)
    return "".join(out).strip().replace(", double>", ">")


def read(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def main():
    workload, fetch_csv, write_csv, out_csv, out_json = sys.argv[1:6]
    f, w = read(fetch_csv, "FETCH_SIZE"), read(write_csv, "WRITE_SIZE")
    rows = {}
    for k in sorted(set(f) | set(w)):
        fk, n = f.get(k, (0.0, 0))
        wk, _ = w.get(k, (0.0, 0))
        rows[k] = (n, fk, wk, int((2.0 * fk + wk) * 1024))
    with open(out_csv, "w") as o:
        o.write("kernel,launches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,hbm_bytes_per_launch_corrected\n")
        for k, (n, fk, wk, b) in rows.items():
            o.write("%s,%d,%.1f,%.1f,%d\n" % (k, n, fk, wk, b))
    try:
        traffic = json.load(open(out_json))
    except Exception:
        traffic = {}
    traffic[workload] = {k: v[3] for k, v in rows.items()}
    json.dump(traffic, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Analyses a rocprofv3 --kernel-trace (+ --memory-copy-trace) of tools/stream_probe.py: per tick, when the front chain started
relative to what it waits for, and the kernels' durations - plain ticks against streamed ticks.
    python tools/tl_analyze.py gpurun_out/tl_xxx/runc/<pid>"""
import csv
import sys

pre = sys.argv[1]
K = list(csv.DictReader(open(pre + "_kernel_trace.csv")))
ev = []
for r in K:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dmpp::", "")
    ev.append(dict(c=int(r["Correlation_Id"]), s=int(r["Start_Timestamp"]) / 1e3, e=int(r["End_Timestamp"]) / 1e3, n=name, q=r["Queue_Id"]))
ev.sort(key=lambda x: x["c"])
ticks = []
cur = None
for x in ev:
    if x["n"].startswith("k_effective"):
        cur = {"eff": x}
        ticks.append(cur)
    elif cur is not None:
        for key in ("k_decision", "k_planning", "k_search", "k_score"):
            if x["n"].startswith(key) and key not in cur:
                cur[key] = x
san = [x for x in ev if x["n"].startswith("k_sanitise")]
first_san = san[0]["c"] if san else 1 << 60
ticks = [t for t in ticks if all(k in t for k in ("k_decision", "k_planning", "k_search", "k_score"))]


def stats(sel, label):
    rows = []
    for i, t in enumerate(sel):
        if i < 6:
            continue
        deps = {"planning(t-1)": sel[i - 1]["k_planning"]["e"], "search(t-3)": sel[i - 3]["k_search"]["e"], "score(t-6)": sel[i - 6]["k_score"]["e"]}
        last = max(deps, key=deps.get)
        rows.append(dict(slack=t["eff"]["s"] - deps[last], last=last, eff=t["eff"]["e"] - t["eff"]["s"],
                         dec_gap=t["k_decision"]["s"] - t["eff"]["e"], dec=t["k_decision"]["e"] - t["k_decision"]["s"],
                         pl_gap=t["k_planning"]["s"] - t["k_decision"]["e"], pl=t["k_planning"]["e"] - t["k_planning"]["s"],
                         s_gap=t["k_search"]["s"] - max(t["eff"]["e"], sel[i - 3]["k_score"]["e"]), search=t["k_search"]["e"] - t["k_search"]["s"],
                         sc_gap=t["k_score"]["s"] - t["k_search"]["e"], score=t["k_score"]["e"] - t["k_score"]["s"],
                         period=t["eff"]["s"] - sel[i - 1]["eff"]["s"]))
    if not rows:
        return
    med = lambda k: sorted(r[k] for r in rows)[len(rows) // 2]
    mean = lambda k: sum(r[k] for r in rows) / len(rows)
    print("%s: %d ticks" % (label, len(rows)))
    for k in ("period", "slack", "eff", "dec_gap", "dec", "pl_gap", "pl", "s_gap", "search", "sc_gap", "score"):
        print("   %-8s median %7.1f  mean %7.1f us" % (k, med(k), mean(k)))
    from collections import Counter
    print("   last dependency of the front chain:", dict(Counter(r["last"] for r in rows)))


stats([t for t in ticks if t["eff"]["c"] < first_san], "plain")
stats([t for t in ticks if t["eff"]["c"] >= first_san], "streamed")

#!/usr/bin/env python3
"""Summarise a rocprofv3 results .db: per-kernel average duration and PMC counter sums per dispatch."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
try:
    for r in c.execute("select name,total_calls,average from top_kernels"):
        print("kernel %-60s calls %4d avg %.3f ms" % (r[0][:60], r[1], r[2] / 1e6 if r[2] > 1e5 else r[2] / 1e3))
except Exception as e:
    print("no top_kernels:", e)
if "counters_collection" in tabs:
    cols = [d[1] for d in c.execute("pragma table_info(counters_collection)")]
    rows = list(c.execute("select * from counters_collection"))
    ik, ic, iv = cols.index("kernel_name") if "kernel_name" in cols else None, cols.index("counter_name"), cols.index("value")
    agg = collections.defaultdict(lambda: [0.0, 0])
    idisp = cols.index("dispatch_id") if "dispatch_id" in cols else None
    seen = collections.defaultdict(set)
    for r in rows:
        k = (r[ik][:40] if ik is not None else "?", r[ic])
        agg[k][0] += float(r[iv])
        if idisp is not None:
            seen[k].add(r[idisp])
    for k, (v, _) in sorted(agg.items()):
        n = max(1, len(seen[k]))
        print("pmc %-40s %-24s sum %.4g  per-dispatch %.4g (%d dispatches)" % (k[0], k[1], v, v / n, n))

#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel table.

    python scripts/rocpd_stats.py gpurun_out/prof1/bench_results.db [--skip-first N] > profiles/xxx.md
"""
import re
import sqlite3
import sys


def main():
    path = sys.argv[1]
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % disp)]
    scol = [r[1] for r in c.execute("pragma table_info(%s)" % sym)]
    namecol = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else "name")
    q = ("select s.%s, d.end - d.start from %s d join %s s on d.kernel_id = s.id order by d.start"
         % (namecol, disp, sym))
    rows = list(c.execute(q))
    stats = {}
    for name, dur in rows:
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        st = stats.setdefault(name, [0, 0, 1 << 62, 0])
        st[0] += 1
        st[1] += dur
        st[2] = min(st[2], dur)
        st[3] = max(st[3], dur)
    total = sum(s[1] for s in stats.values())
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for name, (n, tot, mn, mx) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        if len(name) > 90:   # rocPRIM template instantiations
            name = name[:87] + "..."
        print("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (name, n, tot / 1e6, tot / n / 1e3, mn / 1e3, mx / 1e3,
                                                                100.0 * tot / total))
    print("\ntotal kernel time %.3f ms over %d dispatches" % (total / 1e6, len(rows)))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Print the kernel timeline of ONE training step from a rocprofv3 rocpd database
(rocprofv3 --kernel-trace -d DIR -o bench -- python bench.py ...; eager is the default).

    python scripts/step_timeline.py gpurun_out/prof/bench_results.db [step_index_from_end=2]
"""
import re
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(c.execute(
        "select s.kernel_name, d.start, d.end - d.start, d.grid_size_x, d.grid_size_y, d.workgroup_size_x, d.queue_id "
        "from %s d join %s s on d.kernel_id = s.id order by d.start" % (disp, sym)))
    # a step starts with the forward's pack_batch launch (two pack_batch launches per step)
    idx = [i for i, r in enumerate(rows) if "pack_batch_kernel" in r[0]]
    starts = idx[0::2]
    lo, hi = starts[-back - 1], starts[-back]
    t0 = rows[lo][1]
    busy = 0
    for name, st, du, gx, gy, wx, q in rows[lo:hi]:
        nm = re.sub(r"\(.*", "", name)
        nm = re.sub(r"^_Z\d+", "", nm)[:46]
        busy += du
        print("%8.1f %7.1f us  %-46s grid %5d x%2d q%d" % ((st - t0) / 1e3, du / 1e3, nm, gx // wx, gy, q))
    print("step wall %.1f us, summed kernel time %.1f us" % ((rows[hi][1] - t0) / 1e3, busy / 1e3))
    # device idle time: the union of the kernel intervals of all queues against the step wall
    ivs = sorted((st, st + du) for _, st, du, *_ in rows[lo:hi])
    covered, cur_s, cur_e, gaps = 0, ivs[0][0], ivs[0][1], []
    for s, e in ivs[1:]:
        if s > cur_e:
            covered += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    covered += cur_e - cur_s
    wall = rows[hi][1] - t0
    print("device busy (any queue) %.1f us, idle %.1f us in %d gaps (median %.2f us, max %.1f us)" % (
        covered / 1e3, (wall - covered) / 1e3, len(gaps) + 1, sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0,
        max(gaps) / 1e3 if gaps else 0))


if __name__ == "__main__":
    main()

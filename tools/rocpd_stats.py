"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) as a --stats style table.
    python tools/rocpd_stats.py gpurun_out/prof/xxx_results.db [--csv]
"""
import re
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    q = "select s.%s, d.end - d.start from %s d join %s s on d.kernel_id = s.id" % (name_col, kd, ks)
    agg = {}
    for name, dur in c.execute(q):
        name = name.replace("(anonymous namespace)::", "")
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*$", "", name)
        a = agg.setdefault(name, [0, 0, 10 ** 18, 0])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    print("%-88s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-88s %7d %12.1f %10.1f %10.1f %10.1f %6.2f" % (name[:88], a[0], a[1] / 1e3, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / total))
    print("TOTAL kernel time: %.3f ms over %d dispatches" % (total / 1e6, sum(a[0] for a in agg.values())))
    # how much of the wall clock had at least one kernel resident (union of the dispatch intervals), and how often two
    # overlapped (streams): gaps = launch latency / host stalls, overlap = concurrent streams
    iv = sorted(c.execute("select d.start, d.end from %s d" % kd))
    if iv:
        busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
        for a, b in iv[1:]:
            if a > cur_e:
                busy += cur_e - cur_s
                cur_s, cur_e = a, b
            else:
                cur_e = max(cur_e, b)
        busy += cur_e - cur_s
        span = max(b for _, b in iv) - iv[0][0]
        print("first-to-last dispatch span %.3f ms; >=1 kernel resident %.3f ms (%.1f %%); sum of durations / resident time = %.3f" % (
            span / 1e6, busy / 1e6, 100.0 * busy / span, total / busy))

    # per optimizer step (between consecutive launches of the step's last kernel, the fused AdamW): wall span, time with at least
    # one kernel resident, sum of kernel durations -- idle = span - resident is launch latency / dependency stalls, sum / resident
    # > 1 is what the two tower streams overlap
    q = "select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, kd, ks)
    rows = list(c.execute(q))
    marks = [i for i, r in enumerate(rows) if "adamw_seg_kernel" in r[0]]
    if len(marks) >= 3:
        print("per step (AdamW to AdamW): span_ms resident_ms idle_ms sum_ms dispatches")
        for a, b in zip(marks[:-1], marks[1:]):
            seg = rows[a + 1:b + 1]
            t0, t1 = rows[a][2], rows[b][2]
            busy, cs, ce, tot = 0, None, None, 0
            for _, st, en in seg:
                tot += en - st
                if cs is None:
                    cs, ce = st, en
                elif st > ce:
                    busy += ce - cs
                    cs, ce = st, en
                else:
                    ce = max(ce, en)
            busy += ce - cs
            print("   %.3f %.3f %.3f %.3f %d" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, tot / 1e6, len(seg)))


if __name__ == "__main__":
    main(sys.argv[1])

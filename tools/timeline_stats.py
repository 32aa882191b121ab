"""Two-stream timeline of the captured CLIP step from a rocprofv3 --kernel-trace database (rocpd SQLite): per step (AdamW to AdamW) how long
only one queue / both queues had a kernel resident, what ran while the other queue was empty, and the in-step duration of every kernel
class next to its dispatch count.   python tools/timeline_stats.py <results.db>"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:70]


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    print("dispatch columns:", cols)
    q = "select s.%s, d.start, d.end, d.%s from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, qcol, kd, ks)
    rows = [(short(n), a, b, qq) for n, a, b, qq in c.execute(q)]
    marks = [i for i, r in enumerate(rows) if "adamw_seg_kernel" in r[0]]
    if len(marks) < 3:
        print("fewer than 3 steps in the trace")
        return
    spans = sorted((rows[y][2] - rows[x][2], x, y) for x, y in zip(marks[:-1], marks[1:]))
    _, a, b = spans[len(spans) // 2]                 # the step with the median span (a replayed, undisturbed one)
    seg = rows[a + 1:b + 1]
    t0, t1 = rows[a][2], rows[b][2]
    queues = sorted(set(r[3] for r in seg), key=lambda x: -sum(r[2] - r[1] for r in seg if r[3] == x))
    print("step span %.3f ms, %d dispatches, queues by busy time: %s" % ((t1 - t0) / 1e6, len(seg), [(qq, round(sum(r[2] - r[1] for r in seg if r[3] == qq) / 1e6, 3)) for qq in queues]))
    # sweep: at every instant which queues have a kernel resident
    ev = []
    for n, s, e, qq in seg:
        ev.append((s, 1, qq, n))
        ev.append((e, -1, qq, n))
    ev.sort()
    active = defaultdict(int)
    last = t0
    both = only = defaultdict(float)
    state_time = defaultdict(float)
    alone_by_kernel = defaultdict(float)
    cur = {}
    for t, d, qq, n in ev:
        key = tuple(sorted(k for k, v in active.items() if v > 0))
        state_time[key] += t - last
        if len(key) == 1:
            for kn in cur.get(key[0], []):
                alone_by_kernel[(key[0], kn)] += t - last
        last = t
        active[qq] += d
        if d > 0:
            cur.setdefault(qq, []).append(n)
        else:
            cur[qq].remove(n)
    for key, v in sorted(state_time.items(), key=lambda kv: -kv[1]):
        print("  queues resident %-24s %8.3f ms" % (key, v / 1e6))
    print("what ran while it was the ONLY queue with work (ms per step, top 14):")
    for (qq, kn), v in sorted(alone_by_kernel.items(), key=lambda kv: -kv[1])[:14]:
        print("   queue %-6s %-72s %7.3f" % (qq, kn, v / 1e6))
    agg = defaultdict(lambda: [0, 0])
    for n, s, e, qq in seg:
        agg[n][0] += 1
        agg[n][1] += e - s
    print("kernel classes of the step (in-step durations, overlapped kernels stretch each other):")
    for n, (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
        print("   %-72s %4d %9.3f ms %8.1f us avg" % (n, cnt, tot / 1e6, tot / cnt / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])

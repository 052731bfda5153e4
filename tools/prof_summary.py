"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel table (text)."""
import re
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(top_kernels)")]
    rows = [dict(zip(cols, r)) for r in cur.execute("select * from top_kernels")]
    tot = sum(r["total_duration"] for r in rows)
    lines = ["# rocprofv3 --kernel-trace --stats summary of %s" % db,
             "# total kernel time %.3f s over %d kernels (durations from the rocpd 'top_kernels' view, microseconds)" % (tot / 1e6, len(rows)),
             "%7s %11s %8s %11s  %s" % ("pct", "total_ms", "calls", "avg_us", "kernel")]
    for r in rows[:60]:
        n = re.sub(r"\(anonymous namespace\)::", "", r["name"])
        n = re.sub(r"^void ", "", n)
        lines.append("%6.2f%% %11.3f %8d %11.1f  %s" % (100.0 * r["total_duration"] / tot, r["total_duration"] / 1e3,
                                                          r["total_calls"], r["average"], n[:150]))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(*sys.argv[1:])

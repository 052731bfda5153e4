"""Per-kernel totals of one PMC counter from a rocprofv3 rocpd database (``pmc_events`` view).
usage: pmc_summary.py <db> <counter> [out.txt]   -- counter e.g. FETCH_SIZE or WRITE_SIZE (KB units on gfx950)"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return name.split("(")[0][:90]


def main():
    db, counter = sys.argv[1], sys.argv[2]
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, dispatch_id, sum(counter_value) from pmc_events where counter_name = ? "
                       "group by name, dispatch_id", (counter,)).fetchall()
    agg = {}
    for name, _, v in rows:
        a = agg.setdefault(short(name), [0.0, 0])
        a[0] += v
        a[1] += 1
    lines = ["# %s per kernel from %s (sum over XCD instances per dispatch; KB as rocprofv3 reports it)" % (counter, db),
             "%14s %8s %14s  kernel" % ("total_KB", "launches", "KB/launch")]
    for k, (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
        lines.append("%14.0f %8d %14.1f  %s" % (v, n, v / n, k))
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text + "\n")


if __name__ == "__main__":
    main()

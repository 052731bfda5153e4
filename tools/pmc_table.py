"""Per-kernel table of several PMC counters from a rocprofv3 rocpd database (``pmc_events`` view), plus the derived
matrix-pipe utilisation  MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256 CUs * 4 SIMDs)  when both are present
(the gfx94x formula of rocprofiler's derived_counters.xml; ROCm 7.2 ships no gfx950 section, MI355X_MICROARCH.md).
usage: pmc_table.py <db> [out.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return name.split("(")[0][:70]


def main():
    db = sys.argv[1]
    cur = sqlite3.connect(db).cursor()
    # one row per (dispatch, counter, hardware instance): SQ_* are summed over instances; GRBM_GUI_ACTIVE (the chip's busy
    # cycle count, one value per XCD) takes the maximum over instances, as rocprofiler's own MfmaUtil expression does
    rows = cur.execute("select name, counter_name, dispatch_id, sum(counter_value), max(counter_value) from pmc_events "
                       "group by name, counter_name, dispatch_id").fetchall()
    agg, counters = {}, []
    for name, cn, _, vsum, vmax in rows:
        if cn not in counters:
            counters.append(cn)
        a = agg.setdefault(short(name), {})
        t = a.setdefault(cn, [0.0, 0])
        t[0] += vmax if cn.startswith("GRBM") else vsum
        t[1] += 1
    counters.sort()
    key = "GRBM_GUI_ACTIVE" if "GRBM_GUI_ACTIVE" in counters else counters[0]
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_sha256
    lines = ["# per-kernel PMC sums from %s (rocprofv3 --pmc %s)" % (db, " ".join(counters)),
             "# csrc_sha256 %s   (bench.py refuses this summary once the kernel sources change)" % csrc_sha256(),
             "%-72s %8s " % ("kernel", "launches") + " ".join("%22s" % c for c in counters) +
             ("   MfmaUtil" if "SQ_VALU_MFMA_BUSY_CYCLES" in counters and "GRBM_GUI_ACTIVE" in counters else "")]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1].get(key, [0, 0])[0])[:32]:
        n = max(t[1] for t in a.values())
        line = "%-72s %8d " % (k, n) + " ".join("%22.4g" % a.get(c, [0, 0])[0] for c in counters)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "GRBM_GUI_ACTIVE" in a and a["GRBM_GUI_ACTIVE"][0] > 0:
            line += "   %7.3f" % (a["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (a["GRBM_GUI_ACTIVE"][0] * 256 * 4))
        lines.append(line)
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()

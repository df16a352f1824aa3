"""Turn a rocprofv3 results .db (rocpd sqlite) into the text summary committed under profiles/."""
import json
import sqlite3
import sys


def main(db_path, out_path, bench_json=None, title=""):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    lines = []
    lines.append("# rocprofv3 --kernel-trace --stats summary%s" % ((" -- " + title) if title else ""))
    lines.append("# source: %s" % db_path)
    lines.append("")
    lines.append("%-28s %8s %14s %12s %8s" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
    for name, calls, total, avg, pct in cur.execute("select * from top_kernels"):
        lines.append("%-28s %8d %14.3f %12.4f %7.2f%%" % (name.split("(")[0], calls, total / 1e3, avg / 1e3, pct))
    lines.append("")
    lines.append("per-kernel launch geometry / resources (max over dispatches):")
    q = ("select name, count(*), min(duration)/1e6, max(duration)/1e6, max(grid_x), max(workgroup_x), max(lds_size), "
         "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(scratch_size) from kernels group by name")
    lines.append("%-28s %6s %9s %9s %9s %6s %8s %5s %5s %5s %8s" % ("kernel", "calls", "min_ms", "max_ms", "grid_x", "wg_x", "lds_B", "vgpr", "agpr", "sgpr", "scratch"))
    for r in cur.execute(q):
        lines.append("%-28s %6d %9.3f %9.3f %9d %6d %8d %5d %5d %5d %8d" % ((r[0].split("(")[0],) + tuple(r[1:])))
    # counters, if this db came from a --pmc pass
    try:
        rows = list(cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"))
    except Exception:
        rows = []
    if rows:
        lines.append("")
        lines.append("PMC counters (sum over dispatches):")
        for k, c, v, n in rows:
            lines.append("%-28s %-28s %20.6g  (%d dispatches)" % (k.split("(")[0], c, v, n))
    if bench_json:
        try:
            j = json.loads(open(bench_json).read().strip().splitlines()[-1])
            lines.append("")
            lines.append("bench.py line of the same command (HIP-event timing inside bench.py):")
            lines.append("  value %.1f %s, ms_per_step %.2f" % (j["value"], j["unit"], j["ms_per_step"]))
            r = j["roofline"]
            lines.append("  k_pso: launches %d avg_launch_ms %.4f achieved %.1f GB/s (frac %.4f of %g GB/s)" %
                         (r["launches"], r["avg_launch_ms"], r["achieved"], r["frac"], r["peak"]))
        except Exception as e:
            lines.append("(bench json unreadable: %s)" % e)
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(*sys.argv[1:])

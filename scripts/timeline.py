"""Kernel timeline of a rocprofv3 --kernel-trace .db: busy/idle accounting and one round in detail.

usage: python scripts/timeline.py results.db [round_index]
"""
import sqlite3
import sys


def main(db_path, which=20):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    print("columns:", cols)
    sc = "start" if "start" in cols else [c for c in cols if "start" in c][0]
    ec = "end" if "end" in cols else [c for c in cols if "end" in c][0]
    qc = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = list(cur.execute("select name, %s, %s%s from kernels order by %s" % (sc, ec, (", " + qc) if qc else "", sc)))
    rows = [(r[0].split("(")[0], r[1], r[2], r[3] if qc else 0) for r in rows]
    t0 = rows[0][1]
    span = rows[-1][2] - t0
    # union busy
    busy = 0
    cs, ce = rows[0][1], rows[0][2]
    for _, s, e, _q in rows[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    print("kernels %d span %.1f ms, union busy %.1f ms (%.1f%%)" % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span))
    begins = [i for i, r in enumerate(rows) if r[0].startswith("k_begin")]
    print("k_begin launches:", len(begins))
    # per round: time from k_begin start to next k_begin start; gap between last kernel end of round and next k_begin
    gaps = []
    for a, b in zip(begins[:-1], begins[1:]):
        last_end = max(r[2] for r in rows[a:b])
        gaps.append((rows[b][1] - last_end) / 1e3)
    if gaps:
        gs = sorted(gaps)
        print("host gap between rounds (us): median %.0f mean %.0f max %.0f sum %.1f ms" % (gs[len(gs) // 2], sum(gs) / len(gs), gs[-1], sum(gs) / 1e3))
    which = int(which)
    if which < len(begins) - 1:
        a, b = begins[which], begins[which + 1]
        base = rows[a][1]
        print("round %d: %d kernels, %.0f us" % (which, b - a, (rows[b][1] - base) / 1e3))
        for r in rows[a:b]:
            print("  %-22s q%-3s start %8.1f us  dur %7.1f us" % (r[0], r[3], (r[1] - base) / 1e3, (r[2] - r[1]) / 1e3))


if __name__ == "__main__":
    main(*sys.argv[1:])

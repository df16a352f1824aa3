"""Per-round table (candidates, duration, kernel counts) of the second reconstruction in a rocprofv3 kernel-trace db."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = [(r[0].split('(')[0].replace('void ', ''), r[1], r[2], r[3], r[4]) for r in cur.execute("select name,start,end,grid_x,queue_id from kernels order by start")]
begins = [i for i, r in enumerate(rows) if r[0].startswith('k_begin')]
# the second half of the k_begin launches belongs to the timed step (warmup 1, steps 1)
half = (len(begins) // 3) if len(begins) % 3 == 0 else len(begins) // 2   # warm-up, timed, roofline step
end = 2 * half if len(begins) % 3 == 0 else len(begins)
tot = 0
print("round      n   dur_us  iter_kernels  iter_us  after_us  streams  kinds")
for k, (a, b) in enumerate(zip(begins[half:end], (begins[half + 1:] + [len(rows)])[:end - half])):
    seg = rows[a:b]
    n = seg[0][3] // 64
    dur = ((rows[b][1] if b < len(rows) else seg[-1][2]) - seg[0][1]) / 1e3
    it = [r for r in seg if 'k_pso_iter' in r[0] or 'k_pso_eval' in r[0] or 'k_pso_ring' in r[0] or 'k_pso_tile' in r[0]]
    af = sum((r[2] - r[1]) for r in seg if r[0].startswith('k_after') or r[0].startswith('k_region_ratio')) / 1e3
    itus = sum((r[2] - r[1]) for r in it) / 1e3
    kinds = sorted(set(r[0].replace('void ', '') for r in it))
    tot += dur
    print("%3d %8d %8.0f %6d %10.0f %8.0f %6d   %s" % (k, n, dur, len(it), itus, af, len(set(r[4] for r in it)), ",".join(kinds)))
print("total ms %.1f" % (tot / 1e3))
# gaps between dependent launches of single-stream rounds (the thin tail): what a hipGraph of the chain could shave
gaps = []
for a, b in zip(begins[half:end], (begins[half + 1:] + [len(rows)])[:end - half]):
    it = [r for r in rows[a:b] if 'k_pso_iter' in r[0]]
    if len(it) < 20 or len(set(r[4] for r in it)) != 1:
        continue
    gaps += [(it[k + 1][1] - it[k][2]) / 1e3 for k in range(len(it) - 1)]
if gaps:
    gaps.sort()
    print("gap between consecutive k_pso_iter launches of single-stream rounds: median %.2f us, mean %.2f us, p90 %.2f us (%d gaps)"
          % (gaps[len(gaps) // 2], sum(gaps) / len(gaps), gaps[int(len(gaps) * 0.9)], len(gaps)))

"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace run (rocpd database):
python tools/gap_analysis.py <results.db> [first-kernel-prefix last-kernel-prefix]   (default: k_mf_matvec .. pcg_tail, the PCG solve)
Prints the busy / idle split of the solve and the idle time by (previous kernel -> next kernel) pair."""
import collections
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name,start,end from kernels order by start"))
short = lambda n: re.sub(r"\(.*$", "", n).replace("void ", "").replace("miba::", "")
names = [short(r[0]) for r in rows]
p_first = sys.argv[2] if len(sys.argv) > 3 else "k_mf_matvec"
p_last = sys.argv[3] if len(sys.argv) > 3 else "pcg_tail"
first = min(i for i, n in enumerate(names) if n.startswith(p_first))
last = max(i for i, n in enumerate(names) if n.startswith(p_last))
gaps = collections.defaultdict(lambda: [0, 0.0, 0.0])
busy = 0.0
for i in range(first, last):
    g = (rows[i + 1][1] - rows[i][2]) / 1e3
    key = (names[i][:24], names[i + 1][:24])
    gaps[key][0] += 1; gaps[key][1] += g; gaps[key][2] = max(gaps[key][2], g)
    busy += (rows[i][2] - rows[i][1]) / 1e3
span = (rows[last][2] - rows[first][1]) / 1e3
print("span %.1f us, busy %.1f us, idle %.1f us (%.1f %%)" % (span, busy, span - busy, 100 * (span - busy) / span))
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:16]:
    print("%-26s -> %-26s n=%4d total=%9.1f us avg=%7.1f max=%8.1f" % (k[0], k[1], v[0], v[1], v[1] / v[0], v[2]))

#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 --kernel-trace run (rocpd database) as CSV, the layout of profiles/*kernel_stats*.csv.
usage: kernel_stats_csv.py results.db out.csv "first comment line" """
import re, sqlite3, sys
db, out, note = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), max(lds_size), max(scratch_size) "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
short = lambda n: re.sub(r"\(.*$", "", n).replace("void ", "")
mv = c.execute("select end-start from kernels where name like '%k_mf_matvec%'").fetchall()
work = [d[0] for d in mv if d[0] > 20000]
with open(out, "w") as fh:
    if note:
        fh.write("# %s\n" % note)
    if mv:
        fh.write("# k_mf_matvec: %d launches, %d of them did work (the others were enqueued ahead of the host's look at the control "
                 "block and returned at once): average %.1f us over all, %.1f us over the ones that worked (what bench.py's "
                 "roofline times with HIP events)\n" % (len(mv), len(work), sum(d[0] for d in mv) / len(mv) / 1e3,
                                                        (sum(work) / len(work) / 1e3) if work else 0.0))
    fh.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","lds_size","scratch_size"\n')
    for r in rows:
        fh.write('"%s",%d,%d,%.1f,%.2f,%d,%d,%d,%d\n' % (short(r[0]), r[1], r[2], r[3], 100.0 * r[2] / tot, r[4], r[5], r[6], r[7]))

#!/usr/bin/env python3
"""Per-kernel durations out of a rocprofv3 rocpd database (the default output format of this image's rocprofv3 when -o is given).
usage: prof_db.py <dir-or-db> [name-filter]"""
import glob, os, sqlite3, sys
src = sys.argv[1]
dbs = [src] if src.endswith(".db") else sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for db in dbs:
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), avg(end-start)/1000.0, min(end-start)/1000.0, max(end-start)/1000.0, sum(end-start)/1e6 "
                     "from kernels group by name order by 6 desc").fetchall()
    for r in rows:
        if flt in r[0]:
            print("%-110s n %5d  avg %9.1f us  min %9.1f  max %9.1f  total %9.1f ms" % ((r[0][:110],) + r[1:]))

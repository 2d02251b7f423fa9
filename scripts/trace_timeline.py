"""Kernel timeline out of a rocprofv3 results database (rocpd sqlite): the last N dispatches with start, duration and gap."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = list(db.execute("select name, start, end from kernels order by start"))
cp = []
try:
    cp = list(db.execute("select name, start, end from memory_copies order by start"))
except Exception:
    pass
ev = sorted([(s, e, nm) for nm, s, e in rows] + [(s, e, 'COPY ' + str(nm)) for nm, s, e in cp])


def short(nm):
    nm = re.sub(r'^void ', '', nm)
    nm = re.sub(r'orcvio_amd::', '', nm)
    return re.sub(r'\(.*', '', nm)[:44]


ev = ev[-n:]
t0, prev = ev[0][0], None
for s, e, nm in ev:
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} gap {gap:6.1f}  {short(nm)}")
    prev = e

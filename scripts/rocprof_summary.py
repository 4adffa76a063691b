"""Dump the per-kernel summary of a rocprofv3 --kernel-trace --stats run (rocpd sqlite db or stats csv) as a text table."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = ["%-110s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for name, calls, total, avg, pct in rows:
        lines.append("%-110s %8d %14.1f %12.2f %6.2f%%" % (name[:110], calls, total, avg, pct))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "a").write(txt)
    sys.stdout.write(txt)


if __name__ == "__main__":
    main(*sys.argv[1:])

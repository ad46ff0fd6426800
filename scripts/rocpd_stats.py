#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace -d DIR -o NAME -- python3 prog ...`
writes DIR/NAME_results.db).  Usage: rocpd_stats.py DB [--after KERNEL_SUBSTRING] [--top N]
--after: only dispatches that start at or after the first dispatch whose name contains the substring (e.g. the match
phase of bench_db.py starts at the first m_compose_kernel)."""
import argparse
import re
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--after", default=None)
    ap.add_argument("--top", type=int, default=30)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    t0 = min((r[1] for r in rows if a.after and a.after in r[0]), default=0)
    agg = {}
    for name, s, e in rows:
        if s < t0:
            continue
        name = re.sub(r"\(.*", "", name).replace(".kd", "")
        x = agg.setdefault(name, [0, 0])
        x[0] += 1
        x[1] += e - s
    tot = sum(x[1] for x in agg.values())
    print(f"# kernels after {a.after!r}: busy {tot / 1e6:.2f} ms")
    print("kernel,calls,total_ms,avg_us,percent")
    for name, x in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print(f"{name},{x[0]},{x[1] / 1e6:.3f},{x[1] / x[0] / 1e3:.1f},{100 * x[1] / tot:.1f}")


if __name__ == "__main__":
    main()

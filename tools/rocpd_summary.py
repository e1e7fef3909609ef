"""Summarise a rocprofv3 rocpd sqlite (.db) file: per-kernel calls / avg / min / total duration.
Usage: python tools/rocpd_summary.py <results.db> [name-filter]   (prints a markdown table)"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = db.execute(
        "select name, count(*), avg(duration), min(duration), max(duration), sum(duration), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[5] for r in rows) or 1
    print("| kernel | calls | avg us | min us | max us | total ms | % | vgpr | sgpr | lds | grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        if flt and flt not in r[0]:
            continue
        name = r[0] if len(r[0]) < 90 else r[0][:87] + "..."
        print(f"| `{name}` | {r[1]} | {r[2]/1e3:.2f} | {r[3]/1e3:.2f} | {r[4]/1e3:.2f} | "
              f"{r[5]/1e6:.3f} | {100*r[5]/tot:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} |")


if __name__ == "__main__":
    main()

"""ATen (non-hand-written) kernels of a rocprofv3 rocpd .db grouped by (kernel, grid size): which library elementwise / copy /
reduce launches cost the most.  Usage: python tools/rocpd_aten.py <results.db> [steps] [top]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    rows = db.execute("select name, grid_x, count(*), avg(duration), sum(duration) from kernels "
                      "where name like '%at::native%' or name like '%rocprim%' group by name, grid_x "
                      "order by sum(duration) desc").fetchall()
    print("| ms/step | launches/step | avg us | grid | kernel |")
    print("|---|---|---|---|---|")
    for name, grid, n, avg, tot in rows[:top]:
        short = re.sub(r"void |at::native::|\(anonymous namespace\)::", "", name)
        m = re.search(r"(\w+Functor\w*|direct_copy\w*|\w+_kernel\w*<[^,>]*)", short)
        print(f"| {tot / 1e6 / steps:.3f} | {n / steps:.1f} | {avg / 1e3:.1f} | {grid} | `{short[:110]}` |")


if __name__ == "__main__":
    main()

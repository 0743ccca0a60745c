"""Per (kernel, launch grid) summary of a rocprofv3 --kernel-trace run (rocpd sqlite): which level
of the hierarchy the time of a kernel goes to.

    python tools/trace_by_grid.py run_results.db [substring]
"""
import sqlite3
import sys


def main(path, flt=''):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gx = 'grid_x' if 'grid_x' in cols else ('grid_size_x' if 'grid_size_x' in cols else None)
    if gx is None:
        print('columns:', cols)
        return
    gy, gz = gx.replace('x', 'y'), gx.replace('x', 'z')
    wx = 'workgroup_x' if 'workgroup_x' in cols else 'workgroup_size_x'
    rows = cur.execute(
        f"select name, {gx}, {gy}, {gz}, {wx}, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
        f"max(end-start)/1e3, max(lds_size) from kernels where name like ? group by name, {gx}, {gy}, {gz} "
        "order by 7 desc", (f"%{flt}%",)).fetchall()
    tot = sum(r[6] for r in rows)
    print(f"# {path}: {tot:.1f} ms in kernels matching {flt!r}")
    print(f"{'kernel':44s} {'grid (threads)':>20s} {'wg':>5s} {'calls':>7s} {'total_ms':>9s} {'pct':>6s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s} {'lds':>7s}")
    for r in rows[:60]:
        name = r[0].replace('void (anonymous namespace)::', '').split('(')[0][:44]
        print(f"{name:44s} {str((r[1], r[2], r[3])):>20s} {r[4]:5d} {r[5]:7d} {r[6]:9.2f} {100 * r[6] / tot:6.2f} {r[7]:8.2f} {r[8]:8.2f} {r[9]:8.2f} {r[10]:7d}")


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')

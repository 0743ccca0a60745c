"""Summarise a rocprofv3 `--kernel-trace --stats` run (rocpd sqlite output, ROCm 7.2) as a
per-kernel table: calls, total ms, share, average / min / max duration in microseconds.

    python tools/rocpd_summary.py gpurun_out/prof_x/run_results.db > profiles/r01_x.txt
"""
import sqlite3
import sys


def main(path, top=30):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute(
        "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
        "max(end-start)/1e3, max(vgpr_count), max(accum_vgpr_count), max(lds_size), "
        "max(scratch_size) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# source: {path}")
    print(f"# total kernel time: {tot:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'kernel':72s} {'calls':>6s} {'total_ms':>10s} {'pct':>6s} {'avg_us':>10s} "
          f"{'min_us':>9s} {'max_us':>9s} {'vgpr':>5s} {'agpr':>5s} {'lds':>6s} {'scr':>5s}")
    for r in rows[:top]:
        print(f"{r[0][:72]:72s} {r[1]:6d} {r[2]:10.3f} {100*r[2]/tot:6.2f} {r[3]:10.2f} "
              f"{r[4]:9.2f} {r[5]:9.2f} {r[6]:5d} {r[7]:5d} {r[8]:6d} {r[9]:5d}")


if __name__ == '__main__':
    main(sys.argv[1])

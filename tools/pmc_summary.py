"""Summarise `rocprofv3 --pmc ... --output-format csv` counter_collection files per kernel.

    python tools/pmc_summary.py run_counter_collection.csv [ncells]

Prints, per (kernel, counter), the number of dispatches and the mean / max counter value;
with ncells also value*1024/ncells (FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE counts half
of the bytes of wide streaming reads on gfx950 -- MI355X_MICROARCH.md, HBM section).
"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    ncells = float(sys.argv[2]) if len(sys.argv) > 2 else None
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[(row['Kernel_Name'][:70], row['Counter_Name'])].append(float(row['Counter_Value']))
    print(f"{'kernel':70s} {'counter':18s} {'n':>5s} {'mean':>14s} {'max':>14s}" + ("  B/cell(max)" if ncells else ""))
    for (k, c), v in sorted(acc.items()):
        line = f"{k:70s} {c:18s} {len(v):5d} {sum(v) / len(v):14.1f} {max(v):14.1f}"
        if ncells:
            line += f"  {max(v) * 1024 / ncells:10.1f}"
        print(line)


if __name__ == '__main__':
    main()

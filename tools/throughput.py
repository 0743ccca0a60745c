"""Throughput of SEVERAL independent solves sharing one GPU (survey mode).

    python tools/throughput.py [--workload marine128] [--k 1,2,3,4] [--steps 6]

One multigrid cycle leaves most of an MI355X idle on its coarse levels (few, sequential line
recurrences), so independent (source, frequency) pairs -- what emg3d's Simulation farms
out to a process pool -- can share a GPU. Each of the K solves runs in its own host thread on
its own HIP stream with its own device hierarchy; reported is the aggregate
Mcell-sweeps/s over `steps` cycles of every solve.
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch                        # noqa: E402
from bench import Bench, workload   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='marine128')
    ap.add_argument('--k', default='1,2,3,4')
    ap.add_argument('--steps', type=int, default=6)
    args = ap.parse_args()
    device = torch.device('cuda', 0)
    for k in [int(x) for x in args.k.split(',')]:
        benches, streams = [], []
        for i in range(k):
            s = torch.cuda.Stream(device)
            with torch.cuda.stream(s):
                b = Bench(workload(args.workload, source_index=i), device)
                b.cycles((b.solver._GRAPH_AFTER + 1) * b.var.maxcycle + 2)
            benches.append(b)
            streams.append(s)
        torch.cuda.synchronize()
        w0 = [b.var.smoother_cell_sweeps for b in benches]
        start = threading.Barrier(k + 1)

        def run(b, s):
            with torch.cuda.stream(s):
                start.wait()
                b.cycles(args.steps)
                s.synchronize()
        threads = [threading.Thread(target=run, args=(b, s)) for b, s in zip(benches, streams)]
        for t in threads:
            t.start()
        start.wait()
        t0 = time.perf_counter()
        for t in threads:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        work = sum(b.var.smoother_cell_sweeps - w for b, w in zip(benches, w0))
        print(f"{args.workload}: {k} concurrent solves: {work / dt / 1e6:8.1f} Mcell-sweeps/s aggregate, "
              f"{dt / args.steps * 1e3:7.2f} ms per cycle round", flush=True)
        del benches, streams
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()

"""A/B of library options on one full-size level (run through gpurun): same start field, smoother
`lr` with option set A and with option set B -- results compared bit by bit, both timed.

    python tools/ab.py --lr 0 --n 256 --a point_prefetch=0 --b point_prefetch=3
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from emg3d_amd import _lib                      # noqa: E402
from microbench import make_level, timeit, report  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lr', type=int, default=0)
    ap.add_argument('--n', type=int, default=256)
    ap.add_argument('--shape', default='')
    ap.add_argument('--case', default='triaxial')
    ap.add_argument('--nu', type=int, default=2)
    ap.add_argument('--a', default='')
    ap.add_argument('--b', action='append', default=[])
    ap.add_argument('--reps', type=int, default=9)
    args = ap.parse_args()
    lib = _lib.lib()
    shape = tuple(int(x) for x in args.shape.split(',')) if args.shape else None
    lv, grid = make_level(args.n, args.case, shape=shape)
    e0 = lv.e.clone()

    def setopts(spec):
        for o in [x for x in spec.split(',') if x]:
            k, v = o.split('=')
            assert lib.emg3d_set_option(k.encode(), int(v)) == 0, o

    def run(spec):
        setopts(spec)
        lv.e.copy_(e0)
        lv.smooth(args.lr, args.nu)
        torch.cuda.synchronize()
        out = lv.e.clone()
        med, mn = timeit(lambda: lv.smooth(args.lr, args.nu), reps=args.reps)
        report(f"lr={args.lr} [{spec}]", med, grid.n_cells, args.nu, args.case)
        return out

    ra = run(args.a)
    for spec in args.b:
        rb = run(spec)
        same = bool(torch.equal(torch.view_as_real(ra), torch.view_as_real(rb)))
        d = (ra - rb).abs().max().item() / ra.abs().max().item()
        print(f"    [{spec}] vs [{args.a}]: bit-identical {same}, max rel diff {d:.2e}", flush=True)
        # back to A's settings for options B changed
        setopts(args.a)


if __name__ == '__main__':
    main()

"""Per-call times of the level-0 smoothing calls of the bench's timed cycles, in launch order (through gpurun):
is a call slower after the coarse-grid correction (tens of ms of tiny kernels) than before it?
    python tools/level0_calls.py [workload] [cycles]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                              # noqa: E402
import torch                                    # noqa: E402
import emg3d_amd as emg3d                       # noqa: E402
import bench                                    # noqa: E402


def main():
    wlname = sys.argv[1] if len(sys.argv) > 1 else 'triaxial256'
    ncyc = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    wl = bench.workload(wlname)
    model = emg3d.Model(emg3d.TensorMesh(wl['h'], wl['origin']), **wl['res'])
    b = bench.Bench(wl, model, torch.device('cuda', 0))
    b.cycles(9)
    b.recording = True
    b.cycles(ncyc)
    torch.cuda.synchronize()
    per = {}
    seq = []
    for i, (lr, nu, a, e) in enumerate(b.events):
        ms = a.elapsed_time(e) / (4 * nu - (nu - 1))
        # calls come as pre-smoothing (first two of a cycle: two directions) and post-smoothing (last two)
        pos = 'pre' if (i % 4) < 2 else 'post'
        per.setdefault((lr, pos), []).append(ms)
        seq.append(f"{'xyz'[lr - 1]}{pos[:2]} {ms:.3f}")
    print(' | '.join(seq))
    for (lr, pos), v in sorted(per.items()):
        print(f"{'xyz'[lr - 1]}-lines {pos:4s}: {np.mean(v):.4f} ms per launch over {len(v)} calls")


if __name__ == '__main__':
    main()

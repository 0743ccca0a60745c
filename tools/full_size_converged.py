"""Converged parity at FULL size (through gpurun): BASELINE.json configs 2 / 3 / 5 exactly as bench.py builds them,
solved to tol 1e-10 on the GPU (the solver's defaults: compact line records where 'auto' allows them, finest level in
residual form) and by the oracle's multigrid driver in the same smoother ordering (order 1, its independent classes
walked by threads: bit-identical with the serial walk, fp64 everywhere): cycle counts, final errors, rel-L2 of the
converged fields. The reference order (sequential) is not affordable at 256^3 (~30 min per cycle); its converged
field is the same fixed point -- compared on the reduced copies in tests/test_gpu_parity.py.

    python tools/full_size_converged.py [marine128 triaxial256 salt384[:PAIR] ...]    (-> profiles/r06_full_size_converged.txt)
    TOL=1e-10 (default); LINE_COMPACT=auto|0|1 (default auto)

TEST INFRASTRUCTURE: the oracle is the checker here, the product path is emg3d_amd.solve."""
import os
import sys
import time

root = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np                                           # noqa: E402
import torch                                                 # noqa: E402
import emg3d_amd as emg3d                                    # noqa: E402
from oracle import core as ocore, mg_ref                     # noqa: E402
from helpers import relerr, usable_cores                     # noqa: E402
from bench import workload                                   # noqa: E402


def main():
    names = sys.argv[1:] or ['marine128', 'triaxial256', 'salt384:0']
    tol = float(os.environ.get('TOL', 1e-10))
    compact = {'auto': 'auto', '0': False, '1': True}[os.environ.get('LINE_COMPACT', 'auto')]
    nt = usable_cores()
    ocore.lib().oracle_set_threads(nt)
    print(f'oracle threads {nt}, tol {tol}, line_compact={compact}', flush=True)
    for spec in names:
        name, _, pair = spec.partition(':')
        wl = workload(name, source_index=int(pair or 0))
        grid = emg3d.TensorMesh(wl['h'], wl['origin'])
        model = emg3d.Model(grid, **wl['res'])
        sfield = emg3d.get_source_field(grid, wl['source'], wl['frequency'])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e, info = emg3d.solve(model, sfield, sslsolver=False, tol=tol, return_info=True, line_compact=compact, **wl['opts'])
        torch.cuda.synchronize()
        tg = time.perf_counter() - t0
        ogrid = mg_ref.Grid(grid.h, grid.origin)
        cond = {k: 1.0 / np.asarray(v, dtype=float) for k, v in wl['res'].items()}
        vm = mg_ref.volume_model(ogrid, wl['frequency'], cond['property_x'], cond.get('property_y'), cond.get('property_z'))
        t0 = time.perf_counter()
        eo, io = mg_ref.solve(vm, mg_ref.Field(ogrid, sfield.field.copy()), tol=tol, order=1, **wl['opts'])
        to = time.perf_counter() - t0
        print(f"{spec} {grid.shape_cells} {wl['frequency']} Hz {wl['opts']} | GPU exit {info['exit']} cycles {info['it_mg']} "
              f"rel. error {info['rel_error']:.2e} residual_form {info['residual_form']} {tg:.2f} s | oracle (same order) exit "
              f"{io['exit']} cycles {io['it_mg']} rel. error {io['rel_error']:.2e} {to:.0f} s | rel-L2 of the fields "
              f"{relerr(e.field, eo.field):.2e}", flush=True)
        del model, sfield, e, eo, vm
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()

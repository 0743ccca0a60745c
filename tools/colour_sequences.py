"""Which sequence of the four colour classes should the passes of a line-smoothing call follow? Measured with
the oracle on the CPU (round 3): cycles to tol 1e-6 of a bench workload's reduced copy for ALL pairs
(sequence of the first sweep of a call, sequence of the second) -- the oracle's experiment hooks
oracle_set_colour_order / oracle_set_colour_order_backward (mirrored-rule mode of the line smoothers).
    python tools/colour_sequences.py triaxial64 0123,0132,...        (first-sweep sequences to scan)
Result (profiles/r03_colour_sequences_triaxial64.txt, 576 pairs): mirrored pairs 12-13 cycles, cyclic
continuations 10, pairs without a shared class between the sweeps (8 passes per call) 9-12; adopted:
the cyclic sequence 1,2,3,0,1,... (launch.h: line_sweep_colour)."""
import sys, time, ctypes, itertools
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from bench import workload
from oracle import core as ocore, mg_ref
import emg3d_amd as emg3d
from concurrent.futures import ProcessPoolExecutor
name = sys.argv[1]
ws = workload(name)
grid = emg3d.TensorMesh(ws['h'], ws['origin'])
sf = emg3d.get_source_field(grid, ws['source'], ws['frequency'])
og = mg_ref.Grid(grid.h, grid.origin)
cond = {k: 1.0 / np.asarray(v, dtype=float) for k, v in ws['res'].items()}
vm = mg_ref.volume_model(og, ws['frequency'], cond['property_x'], cond.get('property_y'), cond.get('property_z'))
def run(args):
    bwd, fwd = args
    lib = ocore.lib()
    lib.oracle_set_line_order(0, 1, 2, 3, 0)          # mirrored-rule mode: sequences below
    lib.oracle_set_colour_order(*fwd)
    lib.oracle_set_colour_order_backward(1, *bwd)
    _, io = mg_ref.solve(vm, mg_ref.Field(og, sf.field.copy()), tol=1e-6, order=1, maxit=40, **ws['opts'])
    return bwd, fwd, io['it_mg'], io['exit_message'], float(io['rel_error'])
if __name__ == '__main__':
    bwds = [tuple(int(c) for c in b) for b in sys.argv[2].split(',')]
    combos = [(b, f) for b in bwds for f in itertools.permutations(range(4))]
    with ProcessPoolExecutor(16) as ex:
        for r in ex.map(run, combos):
            print(name, 'bwd', r[0], 'fwd', r[1], 'cycles', r[2], r[3], '%.2e' % r[4], flush=True)

"""Over-relaxation INSIDE the colour passes (true SOR: every solved value written as old + omega (new - old)),
measured with the oracle on the CPU before any kernel was touched:
cycles to tol 1e-6 of a bench workload's reduced copy, lexicographic order (0) and the GPU's four-colour
order (1).   python tools/omega_inpass_oracle.py triaxial64 1.0 1.1 1.2 1.3 1.4
Results of round 3: profiles/r03_omega_inpass_oracle.txt (no gain in the four-colour order)."""
import sys, time, ctypes
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from bench import workload
from oracle import core as ocore, mg_ref
import emg3d_amd as emg3d
lib = ocore.lib()
lib.oracle_set_omega.argtypes = [ctypes.c_double]
name = sys.argv[1]
omegas = [float(x) for x in sys.argv[2:]]
ws = workload(name)
grid = emg3d.TensorMesh(ws['h'], ws['origin'])
sf = emg3d.get_source_field(grid, ws['source'], ws['frequency'])
og = mg_ref.Grid(grid.h, grid.origin)
cond = {k: 1.0 / np.asarray(v, dtype=float) for k, v in ws['res'].items()}
vm = mg_ref.volume_model(og, ws['frequency'], cond['property_x'], cond.get('property_y'), cond.get('property_z'))
for order in (0, 1):
    for w in omegas:
        lib.oracle_set_omega(w)
        t0 = time.time()
        _, io = mg_ref.solve(vm, mg_ref.Field(og, sf.field.copy()), tol=1e-6, order=order, maxit=60, **ws['opts'])
        print(name, 'order', order, 'omega', w, 'cycles', io['it_mg'], io['exit_message'], f'{time.time()-t0:.1f}s', flush=True)
lib.oracle_set_omega(1.0)

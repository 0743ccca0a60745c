import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import emg3d_amd as emg3d
import bench
for name in ('triaxial256', 'salt384'):
    wl = bench.workload(name)
    grid = emg3d.TensorMesh(wl['h'], wl['origin'])
    model = emg3d.Model(grid, **wl['res'])
    out = {}
    for lc in (False, 'auto'):
        sf = emg3d.get_source_field(grid, wl['source'], wl['frequency'])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e, info = emg3d.solve(model, sf, sslsolver=True, tol=1e-8, return_info=True, line_compact=lc, **wl['opts'])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out[lc] = e.field.copy()
        print(name, 'bicgstab line_compact', lc, 'exit', info['exit'], 'it_ssl', info['it_ssl'], 'it_mg', info['it_mg'], 'rel_error %.2e' % info['rel_error'], '%.2f s' % dt, flush=True)
    print('   rel-L2 between the two fields %.2e' % (np.linalg.norm(out[False] - out['auto']) / np.linalg.norm(out[False])))

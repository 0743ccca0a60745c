import numpy as np, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from oracle import core as ocore, mg_ref
from emg3d_amd import core
from helpers import relerr
g=np.load('tests/golden/kernels.npz')
def _case(g, name):
    p = name + '_'
    grid = mg_ref.Grid([g[p + 'hx'], g[p + 'hy'], g[p + 'hz']], g[p + 'origin'])
    ex = np.asfortranarray(g[p + 'eta_x']); case = str(g[p + 'case'])
    ey = np.asfortranarray(g[p + 'eta_y']) if case in ('HTI', 'triaxial') else ex
    ez = np.asfortranarray(g[p + 'eta_z']) if case in ('VTI', 'triaxial') else ex
    return grid, mg_ref.VModel(grid, ex, ey, ez, np.asfortranarray(g[p + 'zeta']), case)
for name in g['meta_cases']:
    name=str(name); p=name+'_'
    grid,vm=_case(g,name)
    s = mg_ref.Field(grid, g[p + 'gs_s'].copy())
    for fn in ('gauss_seidel','gauss_seidel_x','gauss_seidel_y','gauss_seidel_z'):
        for nu in (1,2):
            a = mg_ref.Field(grid, g[p + 'gs_e_in'].copy()); b = mg_ref.Field(grid, g[p + 'gs_e_in'].copy())
            args = (s.fx, s.fy, s.fz, vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta, *grid.h, nu)
            getattr(ocore, fn)(a.fx, a.fy, a.fz, *args, order=1)
            getattr(core, fn)(b.fx, b.fy, b.fz, *args)
            print(name, grid.shape_cells, fn, nu, '%.2e'%relerr(b.field,a.field))

"""emg3d_amd -- the multigrid inner loop of emsig/emg3d on AMD MI355X (gfx950).

Drop-in for ONE path of the reference: the kernels of ``emg3d/core.py`` (block
Gauss-Seidel smoothers with and without line relaxation, the matrix-free curl-curl
residual, restriction, the banded LDL^T line solve) and the parts of ``emg3d/solver.py``
that drive them (``solve``, ``multigrid``, prolongation, model restriction, norms),
re-built as hand-written HIP kernels behind a C ABI (include/emg3d_amd.h) with the
reference's Python signatures on top:

>>> import emg3d_amd as emg3d
>>> grid = emg3d.TensorMesh([hx, hy, hz], origin)
>>> model = emg3d.Model(grid, property_x=1.0)
>>> sfield = emg3d.get_source_field(grid, (0, 0, 0, 0, 0), frequency=1.0)
>>> efield = emg3d.solve(model, sfield, sslsolver=False)

Surveys, simulations, gridding, I/O, CLI and inversion of the reference are out of scope
(SURVEY.md section 2). There is no CPU fallback: without the HIP library or a GPU the
device entry points raise.
"""
from emg3d_amd import core, fields, meshes, models, solver
from emg3d_amd.fields import Field, get_source_field, get_magnetic_field, get_receiver
from emg3d_amd.meshes import TensorMesh
from emg3d_amd.models import Model
from emg3d_amd.solver import solve, solve_batch, solve_source

__all__ = ['core', 'fields', 'meshes', 'models', 'solver', 'Field', 'Model', 'TensorMesh',
           'get_source_field', 'get_magnetic_field', 'get_receiver', 'solve', 'solve_batch', 'solve_source']

__version__ = '0.1.0'

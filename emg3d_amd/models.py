"""Model containers (host side).

``Model`` holds the electrical properties on the cells; ``VolumeModel`` turns them into
the two volume-integrated coefficient sets the kernels consume, with the semantics of
the reference's ``emg3d.models.VolumeModel`` (reference emg3d/models.py:627-717):

    eta_{x,y,z} = -s mu_0 V (sigma_{x,y,z} [+ s eps_0 eps_r]),      zeta = V / mu_r,

including its aliasing rule: ``eta_y`` / ``eta_z`` ARE ``eta_x`` unless the anisotropy
case has an independent y / z property (emg3d/models.py:693-712).
"""
import numpy as np

from emg3d_amd import meshes
from emg3d_amd.fields import EPSILON_0

__all__ = ['Model', 'VolumeModel']

_MAPS = {
    'Resistivity': lambda p: 1.0 / p,
    'Conductivity': lambda p: p,
    'LgResistivity': lambda p: 10.0 ** (-p),
    'LgConductivity': lambda p: 10.0 ** p,
    'LnResistivity': lambda p: np.exp(-p),
    'LnConductivity': lambda p: np.exp(p),
}


class Model:
    """Resistivity/conductivity model on a tensor mesh (subset of emg3d.models.Model,
    reference emg3d/models.py:33-620, needed to feed the solver)."""

    def __init__(self, grid, property_x=1., property_y=None, property_z=None, mu_r=None,
                 epsilon_r=None, mapping='Resistivity'):
        if mapping not in _MAPS:
            raise ValueError(f"Unknown mapping {mapping!r}; known: {sorted(_MAPS)}.")
        self.grid = grid
        self.mapping = mapping
        self.shape = grid.shape_cells
        self.property_x = self._init(property_x, 'property_x')
        self.property_y = self._init(property_y, 'property_y')
        self.property_z = self._init(property_z, 'property_z')
        self.mu_r = self._init(mu_r, 'mu_r')
        self.epsilon_r = self._init(epsilon_r, 'epsilon_r')
        self.case = {(False, False): 'isotropic', (True, False): 'HTI', (False, True): 'VTI',
                     (True, True): 'triaxial'}[(self.property_y is not None,
                                                self.property_z is not None)]

    def _init(self, value, name):
        if value is None:
            return None
        value = np.asarray(value, dtype=np.float64)
        if value.ndim == 0:
            out = np.full(self.shape, float(value), order='F')
        elif value.shape == self.shape:
            out = np.asfortranarray(value)
        elif value.size == self.grid.n_cells:
            out = np.asfortranarray(value.reshape(self.shape, order='F'))
        else:
            raise ValueError(f"Shape of {name} must be () or {self.shape}; provided: "
                             f"{value.shape}.")
        if not np.all(np.isfinite(out)):
            raise ValueError(f"`{name}` must be all finite.")
        log_mapped = self.mapping.startswith('L') and name.startswith('property')
        if not log_mapped and np.any(out <= 0):
            raise ValueError(f"`{name}` must be all positive.")
        return out

    def __repr__(self):
        return (f"Model: {self.mapping}; {self.case}; {self.shape[0]} x {self.shape[1]} x "
                f"{self.shape[2]} ({self.grid.n_cells:,})")

    def conductivity(self, name):
        prop = getattr(self, name)
        return None if prop is None else _MAPS[self.mapping](prop)

    def interpolate_to_grid(self, grid):
        """Identity on the same grid (reference emg3d/models.py:346-347); re-gridding is
        out of scope of this package."""
        if grid == self.grid:
            return self
        raise NotImplementedError("emg3d_amd: model re-gridding is out of scope.")


class VolumeModel:
    """Volume-integrated eta_{x,y,z} and zeta for one frequency (taken from `sfield`)."""

    def __init__(self, model, sfield):
        self.case = model.case
        self.grid = meshes.BaseMesh(model.grid.h, model.grid.origin)
        vol = self.grid.cell_volumes.reshape(model.shape, order='F')
        if sfield.sval is None:
            raise ValueError("Source field is missing frequency information.")

        etas = {}
        for name in ('property_x', 'property_y', 'property_z'):
            cond = model.conductivity(name)
            if cond is None:
                etas[name] = None
            elif model.epsilon_r is None:                       # diffusive approximation
                etas[name] = np.asfortranarray(-sfield.smu0 * vol * cond)
            else:
                smu = sfield.sval * EPSILON_0 * model.epsilon_r
                etas[name] = np.asfortranarray(-sfield.smu0 * vol * (cond + smu))
        self._eta_x, self._eta_y, self._eta_z = (etas['property_x'], etas['property_y'],
                                                 etas['property_z'])
        self._zeta = np.asfortranarray(vol.copy() if model.mu_r is None else vol / model.mu_r)

    @property
    def eta_x(self):
        return self._eta_x

    @property
    def eta_y(self):
        return self._eta_y if self.case in ('HTI', 'triaxial') else self._eta_x

    @property
    def eta_z(self):
        return self._eta_z if self.case in ('VTI', 'triaxial') else self._eta_x

    @property
    def zeta(self):
        return self._zeta

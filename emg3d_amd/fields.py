"""Electric field container and source-field construction (host side).

``Field`` keeps the reference's layout contract (reference emg3d/fields.py:88-119,
191-293): ONE 1-D buffer ``[fx | fy | fz]`` with Fortran-ordered 3-D views, dtype
complex128 in the frequency domain (f > 0) and float64 in the Laplace domain (f < 0).
That buffer is what is uploaded to / downloaded from HBM as a whole.

``get_source_field`` builds the right-hand side ``-s mu_0 J_s`` for electric dipole
sources (reference emg3d/fields.py:386-519). It runs once per solve on the host and is
not part of the accelerated path; it is own code (segment clipping against the grid
planes + adjoint trilinear weights), checked against source fields of the reference in
tests/test_host_api.py.
"""
import numpy as np

from emg3d_amd import meshes

__all__ = ['Field', 'get_source_field', 'MU_0', 'EPSILON_0']

# scipy 1.15.3 CODATA-2022 values (the reference takes them from scipy.constants;
# SURVEY.md section 0 item 8). Hard-coded so that results do not move with scipy.
MU_0 = 1.25663706127e-06
EPSILON_0 = 8.8541878188e-12


class Field:
    """A 3-D electric field on the edges of a tensor mesh."""

    def __init__(self, grid, data=None, frequency=None, dtype=None, electric=True):
        if not electric:
            raise NotImplementedError("emg3d_amd: only electric (edge) fields are supported.")
        if frequency is not None:
            if frequency > 0:
                dtype = np.complex128
            elif frequency < 0:
                dtype = np.float64
            else:
                raise ValueError(
                    "`frequency` must be f>0 (frequency domain) or f<0 "
                    f"(Laplace domain). Provided: {frequency} Hz.")
        elif data is not None:
            dtype = np.asarray(data).dtype
        elif dtype is None:
            dtype = np.complex128

        self.grid = grid
        self._frequency = frequency
        self.electric = True
        if data is None:
            self._field = np.zeros(grid.n_edges, dtype=dtype)
        else:
            self._field = np.asarray(data, dtype=dtype)
            if self._field.shape != (grid.n_edges,):
                raise ValueError(f"Field data must have shape ({grid.n_edges},); "
                                 f"provided: {self._field.shape}.")
        self._sval = None
        self._smu0 = None

    def __repr__(self):
        return (f"{self.__class__.__name__}: electric; {self.grid.shape_cells[0]} x "
                f"{self.grid.shape_cells[1]} x {self.grid.shape_cells[2]}; "
                f"{self.field.size:,}")

    def __eq__(self, field):
        """Same comparison as the reference (rtol 1e-10, emg3d/fields.py:128-136)."""
        equal = isinstance(field, Field) and self.grid == field.grid
        equal = equal and self._frequency == field._frequency
        return bool(equal and np.allclose(self._field, field._field, atol=0, rtol=1e-10))

    def copy(self):
        return Field(self.grid, self._field.copy(), frequency=self._frequency)

    def to_dict(self, copy=False):
        return {'__class__': 'Field', 'grid': self.grid.to_dict(copy),
                'data': self._field.copy() if copy else self._field,
                'frequency': self._frequency, 'electric': True}

    @classmethod
    def from_dict(cls, inp):
        return cls(meshes.TensorMesh.from_dict(inp['grid']), inp['data'],
                   frequency=inp.get('frequency'))

    # -- buffer and views --------------------------------------------------------------
    @property
    def field(self):
        """Entire field as 1-D array [fx, fy, fz]."""
        return self._field

    @field.setter
    def field(self, value):
        self._field[:] = value

    @property
    def fx(self):
        """x-directed field, shape (nx, ny+1, nz+1), Fortran-ordered view."""
        return self._field[:self.grid.n_edges_x].reshape(self.grid.shape_edges_x, order='F')

    @fx.setter
    def fx(self, value):
        self._field[:self.grid.n_edges_x] = np.asarray(value).ravel('F')

    @property
    def fy(self):
        """y-directed field, shape (nx+1, ny, nz+1), Fortran-ordered view."""
        i0 = self.grid.n_edges_x
        return self._field[i0:i0 + self.grid.n_edges_y].reshape(self.grid.shape_edges_y, order='F')

    @fy.setter
    def fy(self, value):
        i0 = self.grid.n_edges_x
        self._field[i0:i0 + self.grid.n_edges_y] = np.asarray(value).ravel('F')

    @property
    def fz(self):
        """z-directed field, shape (nx+1, ny+1, nz), Fortran-ordered view."""
        i0 = self.grid.n_edges_x + self.grid.n_edges_y
        return self._field[i0:].reshape(self.grid.shape_edges_z, order='F')

    @fz.setter
    def fz(self, value):
        self._field[self.grid.n_edges_x + self.grid.n_edges_y:] = np.asarray(value).ravel('F')

    # -- frequency ---------------------------------------------------------------------
    @property
    def frequency(self):
        return None if self._frequency is None else abs(self._frequency)

    @property
    def sval(self):
        """Laplace parameter: s = 2 pi i f (f > 0) or s = -f (f < 0)."""
        if self._sval is None and self._frequency is not None:
            if self._frequency < 0:
                self._sval = np.array(-self._frequency)
            else:
                self._sval = np.array(2j * np.pi * self._frequency)
        return self._sval

    @property
    def smu0(self):
        """s * mu_0."""
        if self._smu0 is None and self.sval is not None:
            self._smu0 = self.sval * MU_0
        return self._smu0


# ---------------------------------------------------------------------------------------
def _direction(azimuth, elevation):
    """Unit vector for azimuth (x towards y) and elevation (xy-plane towards +z), degrees."""
    az, el = np.deg2rad(azimuth), np.deg2rad(elevation)
    v = np.array([np.cos(az) * np.cos(el), np.sin(az) * np.cos(el), np.sin(el)])
    v[np.abs(v) < 1e-16] = 0.0
    return v


def _segment_to_edges(grid, p0, p1, out):
    """Distribute the straight current segment p0 -> p1 (vector moment p1 - p0) onto the
    edges: the segment is cut at every grid plane it crosses; each piece deposits its
    x/y/z extent on the four edges of that direction of its cell, weighted bilinearly in
    the two transverse coordinates of the piece's midpoint."""
    nodes = (grid.nodes_x, grid.nodes_y, grid.nodes_z)
    d = p1 - p0
    cuts = [0.0, 1.0]
    for a in range(3):
        if d[a] != 0.0:
            t = (nodes[a] - p0[a]) / d[a]
            cuts.extend(t[(t > 0.0) & (t < 1.0)].tolist())
    cuts = np.unique(np.round(np.array(cuts), 14))
    nx, ny, nz = grid.shape_cells
    oy = nx * (ny + 1) * (nz + 1)
    oz = oy + (nx + 1) * ny * (nz + 1)

    def add(index, value):                  # `out`: flat edge index -> moment, in order of arrival
        out[index] = out.get(index, 0.0) + value

    for t0, t1 in zip(cuts[:-1], cuts[1:]):
        if t1 - t0 <= 0:
            continue
        mid = p0 + 0.5 * (t0 + t1) * d
        piece = (t1 - t0) * d
        idx, r = [], []
        for a in range(3):
            i = int(np.searchsorted(nodes[a], mid[a], side='right')) - 1
            i = min(max(i, 0), nodes[a].size - 2)
            idx.append(i)
            r.append((mid[a] - nodes[a][i]) / grid.h[a][i])
        (ix, iy, iz), (rx, ry, rz) = idx, r
        if piece[0] != 0.0:                 # fx[i, j, k] -> i + nx (j + (ny + 1) k)
            ex = lambda i, j, k: i + nx * (j + (ny + 1) * k)               # noqa: E731
            add(ex(ix, iy, iz), piece[0] * (1 - ry) * (1 - rz))
            add(ex(ix, iy + 1, iz), piece[0] * ry * (1 - rz))
            add(ex(ix, iy, iz + 1), piece[0] * (1 - ry) * rz)
            add(ex(ix, iy + 1, iz + 1), piece[0] * ry * rz)
        if piece[1] != 0.0:                 # fy[i, j, k] -> oy + i + (nx + 1) (j + ny k)
            ey = lambda i, j, k: oy + i + (nx + 1) * (j + ny * k)          # noqa: E731
            add(ey(ix, iy, iz), piece[1] * (1 - rx) * (1 - rz))
            add(ey(ix + 1, iy, iz), piece[1] * rx * (1 - rz))
            add(ey(ix, iy, iz + 1), piece[1] * (1 - rx) * rz)
            add(ey(ix + 1, iy, iz + 1), piece[1] * rx * rz)
        if piece[2] != 0.0:                 # fz[i, j, k] -> oz + i + (nx + 1) (j + (ny + 1) k)
            ez = lambda i, j, k: oz + i + (nx + 1) * (j + (ny + 1) * k)    # noqa: E731
            add(ez(ix, iy, iz), piece[2] * (1 - rx) * (1 - ry))
            add(ez(ix + 1, iy, iz), piece[2] * rx * (1 - ry))
            add(ez(ix, iy + 1, iz), piece[2] * (1 - rx) * ry)
            add(ez(ix + 1, iy + 1, iz), piece[2] * rx * ry)


def get_source_field(grid, source, frequency, strength=1.0, length=1.0, **kwargs):
    """Source field ``-s mu_0 J_s`` of an electric dipole or wire.

    Same call as the reference's ``emg3d.get_source_field`` (emg3d/fields.py:386-519)
    for the electric sources given as coordinates:

    - ``(x, y, z, azimuth, elevation)``: dipole of ``length`` (default 1 m) centred at
      (x, y, z);
    - ``(x0, x1, y0, y1, z0, z1)``: finite dipole between the two electrodes;
    - array of shape (n, 3), n > 2: wire through the given points.

    ``frequency`` > 0: frequency domain (complex), < 0: Laplace domain (real), ``None``:
    the bare source vector.
    """
    if kwargs.get('electric', True) is not True:
        raise NotImplementedError("emg3d_amd: magnetic sources are out of scope.")
    src = np.asarray(source, dtype=float)
    if src.size == 5:
        c = src[:3]
        half = 0.5 * length * _direction(src[3], src[4])
        pts = np.array([c - half, c + half])
    elif src.size == 6 and src.ndim == 1:
        pts = np.array([[src[0], src[2], src[4]], [src[1], src[3], src[5]]])
    elif src.ndim == 2 and src.shape[1] == 3:
        pts = src
    else:
        raise ValueError(f"Source format not understood: {source!r}.")
    pts = np.round(pts, 9)
    lo = np.array([grid.nodes_x[0], grid.nodes_y[0], grid.nodes_z[0]])
    hi = np.array([grid.nodes_x[-1], grid.nodes_y[-1], grid.nodes_z[-1]])
    if np.any(pts < lo - 1e-9) or np.any(pts > hi + 1e-9):
        raise ValueError(f"Provided source outside grid: {pts}.")

    moments = {}
    for p0, p1 in zip(pts[:-1], pts[1:]):
        if np.linalg.norm(p1 - p0) < 1e-15:
            raise ValueError(f"Provided finite dipole has no length: {pts}.")
        _segment_to_edges(grid, p0, p1, moments)

    # A source touches a handful of edges: the scaling is done on those, and the field
    # (100 MB for 128^3) is written once -- zero pages of a fresh allocation stay untouched.
    sfield = Field(grid, frequency=frequency, dtype=None if frequency is not None else np.float64)
    index = np.fromiter(moments.keys(), dtype=np.int64, count=len(moments))
    values = np.fromiter(moments.values(), dtype=np.float64, count=len(moments)).astype(sfield._field.dtype)
    values = values * strength
    if frequency is not None:
        values = values * -sfield.smu0
    sfield._field[index] = values
    sfield._sparse = (index, values)     # valid only while the field is not modified (parallel.solve)
    return sfield

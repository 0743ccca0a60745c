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

__all__ = ['Field', 'get_source_field', 'get_magnetic_field', 'get_receiver', 'get_responses', 'MU_0', 'EPSILON_0']

# scipy 1.15.3 CODATA-2022 values (the reference takes them from scipy.constants;
# SURVEY.md section 0 item 8). Hard-coded so that results do not move with scipy.
MU_0 = 1.25663706127e-06
EPSILON_0 = 8.8541878188e-12


class Field:
    """A 3-D electric field on the edges of a tensor mesh."""

    def __init__(self, grid, data=None, frequency=None, dtype=None, electric=True):
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
        self.electric = bool(electric)
        # electric fields live on the edges, magnetic fields on the faces (emg3d/fields.py:88-119)
        if self.electric:
            self._shapes = (grid.shape_edges_x, grid.shape_edges_y, grid.shape_edges_z)
        else:
            self._shapes = (grid.shape_faces_x, grid.shape_faces_y, grid.shape_faces_z)
        self._sizes = tuple(int(np.prod(sh)) for sh in self._shapes)
        n = sum(self._sizes)
        self._n, self._dtype = n, np.dtype(dtype)
        # The dense buffer exists from the first time somebody looks at it (``field``, ``fx`` ...). Until then a
        # field without data is zero, or zero except for the few entries a source routine deposited
        # (``_deposit``): a dipole source on 384 x 256 x 256 cells is a dozen numbers, its dense form 1.2 GB --
        # whose norm and whose way to the GPU cost a solve ~100 ms each (solver.solve reads ``_untouched``).
        self._dense = None
        self._lazy = None
        if data is not None:
            self._dense = np.asarray(data, dtype=dtype)
            if self._dense.shape != (n,):
                raise ValueError(f"Field data must have shape ({n},); "
                                 f"provided: {self._dense.shape}.")
        self._sval = None
        self._smu0 = None

    @property
    def _field(self):
        if self._dense is None:
            self._dense = np.zeros(self._n, dtype=self._dtype)
            if self._lazy is not None:
                self._dense[self._lazy[0]] = self._lazy[1]
                self._lazy = None
        return self._dense

    @property
    def _untouched(self):
        """True while no dense buffer exists: the field IS its deposited entries (or zero) -- nobody can have
        modified what was never handed out."""
        return self._dense is None

    def _deposit(self, index, values):
        """Entries of a field that is zero elsewhere (unique indices), without creating the dense buffer."""
        if self._dense is None and self._lazy is None:
            self._lazy = (index, values)
        else:
            self._field[index] = values
        self._sparse = (index, values)      # (kept for callers that vouch for an unmodified field: parallel.solve)

    @property
    def dtype(self):
        return self._dtype

    def __repr__(self):
        return (f"{self.__class__.__name__}: {['magnetic', 'electric'][self.electric]}; {self.grid.shape_cells[0]} x "
                f"{self.grid.shape_cells[1]} x {self.grid.shape_cells[2]}; "
                f"{self.field.size:,}")

    def __eq__(self, field):
        """Same comparison as the reference (rtol 1e-10, emg3d/fields.py:128-136)."""
        equal = isinstance(field, Field) and self.grid == field.grid
        equal = equal and self._frequency == field._frequency and self.electric == field.electric
        return bool(equal and np.allclose(self._field, field._field, atol=0, rtol=1e-10))

    def copy(self):
        return Field(self.grid, self._field.copy(), frequency=self._frequency, electric=self.electric)

    def to_dict(self, copy=False):
        return {'__class__': 'Field', 'grid': self.grid.to_dict(copy),
                'data': self._field.copy() if copy else self._field,
                'frequency': self._frequency, 'electric': self.electric}

    @classmethod
    def from_dict(cls, inp):
        return cls(meshes.TensorMesh.from_dict(inp['grid']), inp['data'],
                   frequency=inp.get('frequency'), electric=inp.get('electric', True))

    # -- buffer and views --------------------------------------------------------------
    @property
    def field(self):
        """Entire field as 1-D array [fx, fy, fz]."""
        return self._field

    @field.setter
    def field(self, value):
        self._field[:] = value

    def _component(self, c):
        i0 = sum(self._sizes[:c])
        return self._field[i0:i0 + self._sizes[c]].reshape(self._shapes[c], order='F')

    def _set_component(self, c, value):
        i0 = sum(self._sizes[:c])
        self._field[i0:i0 + self._sizes[c]] = np.asarray(value).ravel('F')

    @property
    def fx(self):
        """x-directed field, Fortran-ordered view: (nx, ny+1, nz+1) on edges, (nx+1, ny, nz) on faces."""
        return self._component(0)

    @fx.setter
    def fx(self, value):
        self._set_component(0, value)

    @property
    def fy(self):
        """y-directed field, Fortran-ordered view: (nx+1, ny, nz+1) on edges, (nx, ny+1, nz) on faces."""
        return self._component(1)

    @fy.setter
    def fy(self, value):
        self._set_component(1, value)

    @property
    def fz(self):
        """z-directed field, Fortran-ordered view: (nx+1, ny+1, nz) on edges, (nx, ny, nz+1) on faces."""
        return self._component(2)

    @fz.setter
    def fz(self, value):
        self._set_component(2, value)

    def get_receiver(self, receiver, method='cubic'):
        """Field at receiver positions (``get_receiver``)."""
        return get_receiver(self, receiver, method)

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


def _point_weights(nodes, value):
    """Lower index, upper index and weight of the upper one for `value` between the ascending
    `nodes` (emg3d/fields.py:695-713: `point_source` / `get_index_and_strength`)."""
    i = max(0, int(np.nonzero(value < np.r_[nodes, np.inf])[0][0]) - 1)
    if i == nodes.size - 1:
        return i, i, 1.0            # (the reference sets both weights to 1 on the last node)
    return i, i + 1, (value - nodes[i]) / (nodes[i + 1] - nodes[i])


def get_point_source_field(grid, coordinates, frequency, strength=1.0):
    """Source field of an electric POINT dipole ``(x, y, z, azimuth, elevation)``: the adjoint of
    the tri-linear interpolation of each field component to that point (the reference's
    ``_point_vector``, emg3d/fields.py:662-746, used for ``TxElectricPoint`` -- the adjoint source
    of a point receiver, emg3d/electrodes.py:683). Component c is spread over the eight values of
    that component around the point, in the coordinates that component lives on (cell centres
    along its own direction, nodes across)."""
    c = np.asarray(coordinates, dtype=float)
    lo = np.array([grid.nodes_x[0], grid.nodes_y[0], grid.nodes_z[0]])
    hi = np.array([grid.nodes_x[-1], grid.nodes_y[-1], grid.nodes_z[-1]])
    if np.any(c[:3] < lo) or np.any(c[:3] > hi):
        raise ValueError(f"Provided source outside grid: {c}.")
    sfield = Field(grid, frequency=frequency, dtype=None if frequency is not None else np.float64)
    direction = _rotation(c[3], c[4])
    nodes = (grid.nodes_x, grid.nodes_y, grid.nodes_z)
    centres = (grid.cell_centers_x, grid.cell_centers_y, grid.cell_centers_z)
    scale = strength * (-sfield.smu0 if frequency is not None else 1.0)
    index, values = [], []
    offset = 0
    for comp in range(3):
        shape = sfield._shapes[comp]
        axes = [centres[a] if a == comp else nodes[a] for a in range(3)]
        (i0, i1, rx), (j0, j1, ry), (k0, k1, rz) = (_point_weights(axes[a], c[a]) for a in range(3))
        wx = ((i0, 1.0 - rx), (i1, rx)) if i0 != i1 else ((i0, 1.0),)
        wy = ((j0, 1.0 - ry), (j1, ry)) if j0 != j1 else ((j0, 1.0),)
        wz = ((k0, 1.0 - rz), (k1, rz)) if k0 != k1 else ((k0, 1.0),)
        for k, wk in wz:
            for j, wj in wy:
                for i, wi in wx:
                    index.append(offset + i + shape[0] * (j + shape[1] * k))
                    values.append(wi * wj * wk * direction[comp])
        offset += sfield._sizes[comp]
    index = np.array(index, dtype=np.int64)
    values = (np.array(values) * scale).astype(sfield.dtype)
    keep = values != 0
    index, values = index[keep], values[keep]
    sfield._deposit(index, values)
    return sfield


def get_magnetic_point_source_field(grid, coordinates, frequency, strength=1.0):
    """Source field of a magnetic POINT dipole ``(x, y, z, azimuth, elevation)`` -- the adjoint source
    of a magnetic point receiver (reference ``_point_vector_magnetic``, emg3d/fields.py:749-789, behind
    ``TxMagneticPoint``, emg3d/electrodes.py:715): the transpose of "magnetic field from the electric
    field, interpolated linearly to the point", i.e. of ``get_receiver(get_magnetic_field(model, e),
    ..., 'linear')`` as a linear map of e. The reference builds it from discretize's face
    interpolation and edge curl; here it is the transpose of this package's own operators
    (``emg3d_dev_magnetic_field`` = the reference's ``_edge_curl_factor``, fields.py:941-1009, and the
    linear receiver interpolation), entry by entry -- mu_r = 1, as the reference requires for it."""
    c = np.asarray(coordinates, dtype=float)
    if frequency is None:
        raise ValueError("The magnetic point source needs a frequency.")
    sfield = Field(grid, frequency=frequency)
    hshapes = Field(grid, frequency=frequency, electric=False)._shapes
    direction = _rotation(c[3], c[4])
    nx, ny, nz = grid.shape_cells
    hx, hy, hz = grid.h
    vol = grid.cell_volumes.reshape(grid.shape_cells, order='F')
    n_ex, n_ey = sfield._sizes[0], sfield._sizes[1]
    ex = lambda i, j, k: i + nx * (j + (ny + 1) * k)                           # noqa: E731
    ey = lambda i, j, k: n_ex + i + (nx + 1) * (j + ny * k)                    # noqa: E731
    ez = lambda i, j, k: n_ex + n_ey + i + (nx + 1) * (j + (ny + 1) * k)       # noqa: E731
    acc = {}

    def add(index, value):
        acc[index] = acc.get(index, 0.0) + value

    for comp in range(3):
        if abs(direction[comp]) <= 1e-10:
            continue
        pts = _component_points(grid, hshapes[comp])
        corner = []
        for d in range(3):                  # the linear receiver interpolation (get_receiver)
            g = pts[d]
            if not g[0] <= c[d] <= g[-1]:
                corner = None
                break
            i = min(max(int(np.searchsorted(g, c[d])) - 1, 0), g.size - 2)
            w = (c[d] - g[i]) / (g[i + 1] - g[i])
            corner.append(((i, 1.0 - w), (i + 1, w)))
        if corner is None:
            continue
        for k, wk in corner[2]:
            for j, wj in corner[1]:
                for i, wi in corner[0]:
                    wf = wi * wj * wk * direction[comp]
                    if wf == 0.0:
                        continue
                    # face (comp; i, j, k) as _edge_curl_factor forms it; faces it leaves at zero: skipped
                    if comp == 0:
                        if i == 0 or i >= nx:
                            continue
                        cf = wf * (vol[i - 1, j, k] + vol[i, j, k]) / ((hx[i - 1] + hx[i]) * hy[j] * hz[k])
                        add(ez(i, j + 1, k), cf / hy[j]); add(ez(i, j, k), -cf / hy[j])
                        add(ey(i, j, k + 1), -cf / hz[k]); add(ey(i, j, k), cf / hz[k])
                    elif comp == 1:
                        if j == 0 or j >= ny:
                            continue
                        cf = wf * (vol[i, j - 1, k] + vol[i, j, k]) / (hx[i] * (hy[j - 1] + hy[j]) * hz[k])
                        add(ex(i, j, k + 1), cf / hz[k]); add(ex(i, j, k), -cf / hz[k])
                        add(ez(i + 1, j, k), -cf / hx[i]); add(ez(i, j, k), cf / hx[i])
                    else:
                        if k == 0 or k >= nz:
                            continue
                        cf = wf * (vol[i, j, k - 1] + vol[i, j, k]) / (hx[i] * hy[j] * (hz[k - 1] + hz[k]))
                        add(ey(i + 1, j, k), cf / hx[i]); add(ey(i, j, k), -cf / hx[i])
                        add(ex(i, j + 1, k), -cf / hy[j]); add(ex(i, j, k), cf / hy[j])
    index = np.array(sorted(acc), dtype=np.int64)
    # H = curl E * zeta / (s mu0): the 1 / (s mu0) of the operator, then strength * (-s mu0) as for every source
    values = np.array([acc[i] for i in index], dtype=complex) / sfield.smu0 * strength * -sfield.smu0
    values = values.astype(sfield.dtype)
    sfield._deposit(index, values)
    return sfield


def get_source_field(grid, source, frequency, strength=1.0, length=1.0, **kwargs):
    """Source field ``-s mu_0 J_s`` of an electric dipole or wire, or (``electric=False``) of a
    magnetic dipole = an electric square loop around it.

    Same call as the reference's ``emg3d.get_source_field`` (emg3d/fields.py:386-519)
    for sources given as coordinates:

    - ``(x, y, z, azimuth, elevation)``: dipole of ``length`` (default 1 m) centred at
      (x, y, z);
    - ``(x0, x1, y0, y1, z0, z1)``: finite dipole between the two electrodes;
    - array of shape (n, 3), n > 2: wire through the given points.

    ``frequency`` > 0: frequency domain (complex), < 0: Laplace domain (real), ``None``:
    the bare source vector.
    """
    src = np.asarray(source, dtype=float)
    magnetic = kwargs.get('electric', True) is not True
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
    if magnetic:
        # ``electric=False``: a magnetic dipole, represented by an electric square loop perpendicular to
        # it and centred on it, whose AREA equals the dipole's length (TxMagneticDipole,
        # emg3d/electrodes.py:536-568, point_to_square_loop :795-822): five points, a closed wire
        if pts.shape[0] != 2:
            raise ValueError(f"A magnetic dipole is given by a point or two electrodes: {source!r}.")
        if src.size == 5:
            centre, azm, elv, area = src[:3], src[3], src[4], float(length)
        else:
            d = pts[1] - pts[0]
            area = float(np.linalg.norm(d))
            azm = np.angle(d[0] + 1j * d[1], deg=True)
            elv = np.angle(np.sqrt(d[0] ** 2 + d[1] ** 2) + 1j * d[2], deg=True)
            centre = pts.sum(0) / 2
        half_diag = np.sqrt(area / 2)
        hor, ver = _rotation(azm + 90.0, 0.0) * half_diag, _rotation(azm, elv + 90.0) * half_diag
        pts = centre + np.stack([hor, ver, -hor, -ver, hor])
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
    values = np.fromiter(moments.values(), dtype=np.float64, count=len(moments)).astype(sfield.dtype)
    values = values * strength
    if frequency is not None:
        values = values * -sfield.smu0
    sfield._deposit(index, values)       # (no dense buffer until somebody asks for it: Field._untouched)
    sfield._segments = (pts, complex(strength) if np.iscomplexobj(strength) else float(strength))
    return sfield


def source_field_device(grid, points, frequency, strength=1.0, out=None):
    """The source field of a wire through ``points`` (n x 3, n >= 2; a dipole is its two end
    points) assembled ON THE DEVICE (``emg3d_dev_source_field``: one thread walks one segment
    from grid plane to grid plane, csrc/adjoint.h) -- what ``get_source_field`` computes on the
    host (reference emg3d/fields.py:386-519, 792-938), without a field-sized host array or
    upload. Returns the device tensor ``[sx | sy | sz]``; ``out``: tensor to fill instead."""
    torch, _lib, _ptr, _stream, dev = _device_tools()
    meta = Field(grid, frequency=frequency, dtype=None if frequency is not None else np.float64)
    dtype = torch.complex128 if np.iscomplexobj(meta.field) else torch.float64
    if out is None:
        out = torch.empty(grid.n_edges, dtype=dtype, device=dev)
    lib = _lib.lib()
    _lib.check(lib.emg3d_dev_zero(_ptr(out), out.numel() * out.element_size(), _stream()), 'emg3d_dev_zero')
    pts = np.round(np.asarray(points, dtype=float), 9)
    geo = np.concatenate([grid.nodes_x, grid.nodes_y, grid.nodes_z, grid.h[0], grid.h[1], grid.h[2], pts.ravel()])
    g = torch.from_numpy(geo).to(dev)
    nx, ny, nz = grid.shape_cells
    o = np.cumsum([0, nx + 1, ny + 1, nz + 1, nx, ny, nz])
    scale = complex(strength) * (-complex(meta.smu0) if frequency is not None else 1.0)
    o1, o2 = grid.n_edges_x, grid.n_edges_x + grid.n_edges_y
    _lib.check(lib.emg3d_dev_source_field(
        nx, ny, nz, int(dtype == torch.complex128), *[_ptr(g, int(o[k])) for k in range(7)], pts.shape[0],
        scale.real, scale.imag, _ptr(out), _ptr(out, o1), _ptr(out, o2), _stream()), 'emg3d_dev_source_field')
    return out


# ---------------------------------------------------------------------------------------
# After a solve: magnetic field and receiver responses, computed
# on the device. The reference does both on the host with NumPy / SciPy.
def _device_tools():
    import torch
    from emg3d_amd import _lib
    from emg3d_amd._device import _ptr, _stream
    _lib.require_gpu()
    return torch, _lib, _ptr, _stream, torch.device('cuda', torch.cuda.current_device())


def get_magnetic_field(model, efield):
    r"""Magnetic field H on the faces from the electric field on the edges with Faraday's law,
    :math:`\nabla \times \mathbf{E} = \rm{i}\omega\mu\mathbf{H}`; same call and result as the
    reference's ``emg3d.get_magnetic_field`` (emg3d/fields.py:617-659, kernel
    ``_edge_curl_factor`` :941-1009), computed by ``emg3d_dev_magnetic_field``."""
    torch, _lib, _ptr, _stream, dev = _device_tools()
    grid = efield.grid
    nx, ny, nz = grid.shape_cells
    hfield = Field(grid, frequency=efield._frequency, dtype=efield.field.dtype, electric=False)
    mu_r = getattr(model, 'mu_r', None)
    vol = grid.cell_volumes
    zeta = vol if mu_r is None else vol / np.asarray(mu_r, dtype=float).ravel('F')     # models.py:688-691
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)                   # noqa: E731
    e = up(efield.field)
    m = torch.empty(hfield.field.size, dtype=e.dtype, device=dev)
    z, hx, hy, hz = up(np.asarray(zeta, dtype=np.float64)), up(grid.h[0]), up(grid.h[1]), up(grid.h[2])
    eo = np.cumsum([0] + list(efield._sizes))          # element offsets of the components
    mo = np.cumsum([0] + list(hfield._sizes))
    smu0 = complex(efield.smu0)
    _lib.check(_lib.lib().emg3d_dev_magnetic_field(
        nx, ny, nz, int(e.is_complex()), _ptr(e, int(eo[0])), _ptr(e, int(eo[1])), _ptr(e, int(eo[2])),
        _ptr(z), _ptr(hx), _ptr(hy), _ptr(hz), smu0.real, smu0.imag,
        _ptr(m, int(mo[0])), _ptr(m, int(mo[1])), _ptr(m, int(mo[2])), _stream()), 'emg3d_dev_magnetic_field')
    torch.from_numpy(hfield.field).copy_(m)
    return hfield


def magnetic_field_device(grid, frequency, e_dev, mu_r=None):
    """``get_magnetic_field`` for an electric field that is in HBM (device tensor ``[ex | ey | ez]``):
    returns (a host ``Field`` describing the magnetic field -- grid, frequency, kind; its values are
    NOT filled --, the device tensor ``[hx | hy | hz]``). For responses of magnetic receivers taken
    from a solution that never leaves the device: ``get_receiver(meta, rec, method, device_field=h)``."""
    torch, _lib, _ptr, _stream, dev = _device_tools()
    nx, ny, nz = grid.shape_cells
    emeta = Field(grid, frequency=frequency)
    hmeta = Field(grid, frequency=frequency, electric=False)
    vol = grid.cell_volumes
    zeta = vol if mu_r is None else vol / np.asarray(mu_r, dtype=float).ravel('F')
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)   # noqa: E731
    z, hx, hy, hz = up(zeta), up(grid.h[0]), up(grid.h[1]), up(grid.h[2])
    hdev = torch.empty(hmeta.field.size, dtype=e_dev.dtype, device=dev)
    eo, mo = np.cumsum([0] + list(emeta._sizes)), np.cumsum([0] + list(hmeta._sizes))
    smu0 = complex(emeta.smu0)
    _lib.check(_lib.lib().emg3d_dev_magnetic_field(
        nx, ny, nz, int(e_dev.is_complex()), _ptr(e_dev, int(eo[0])), _ptr(e_dev, int(eo[1])), _ptr(e_dev, int(eo[2])),
        _ptr(z), _ptr(hx), _ptr(hy), _ptr(hz), smu0.real, smu0.imag,
        _ptr(hdev, int(mo[0])), _ptr(hdev, int(mo[1])), _ptr(hdev, int(mo[2])), _stream()), 'emg3d_dev_magnetic_field')
    return hmeta, hdev


def get_responses(meta, e_dev, receivers, method='cubic', magnetic=None, mu_r=None, efield=None):
    """Responses at point receivers ``(x, y, z, azimuth, elevation)`` -- electric ones [V/m], magnetic
    ones [A/m] where the boolean sequence ``magnetic`` says so (what ``Simulation._get_responses`` collects,
    emg3d/simulations.py:759-792) -- from the solution in HBM (``e_dev``) or, without it, from the host
    field ``efield``. ``meta``: a Field that describes grid and frequency."""
    rec = tuple(np.atleast_1d(np.asarray(c, dtype=float)) for c in receivers)
    src = meta if efield is None else efield
    if magnetic is None or not np.any(magnetic):
        return get_receiver(src, receivers, method, device_field=e_dev)
    mag = np.broadcast_to(np.asarray(magnetic, dtype=bool), np.broadcast(*rec[:3]).shape).ravel()
    rec = tuple(np.broadcast_to(c, mag.shape) for c in rec)
    out = np.array(get_receiver(src, rec, method, device_field=e_dev))
    if e_dev is None:
        torch = _device_tools()[0]
        e_dev = torch.from_numpy(np.ascontiguousarray(efield.field)).to(_device_tools()[4])
    hmeta, hdev = magnetic_field_device(meta.grid, meta._frequency, e_dev, mu_r)
    out[mag] = get_receiver(hmeta, tuple(c[mag] for c in rec), method, device_field=hdev)
    return out


def _rotation(azimuth, elevation):
    """Direction cosines of (azimuth, elevation) in degrees (emg3d/electrodes.py:825-872)."""
    from scipy.special import cosdg, sindg
    return np.array([cosdg(azimuth) * cosdg(elevation), sindg(azimuth) * cosdg(elevation),
                     sindg(elevation)])


def _component_points(grid, shape):
    """Coordinates of the values of one field component per dimension (emg3d/maps.py:439-459):
    as many as nodes -> the nodes, as many as cells -> the cell centres."""
    pts = []
    for d, c in enumerate('xyz'):
        on_nodes = shape[d] == grid.shape_nodes[d]
        pts.append(getattr(grid, ('nodes_' if on_nodes else 'cell_centers_') + c))
    return pts


def get_receiver(field, receiver, method='cubic', device_field=None):
    """Field (response) at receiver coordinates; same call and result as the reference's
    ``emg3d.fields.get_receiver`` (emg3d/fields.py:522-614): ``receiver`` is an object with
    ``.coordinates``, a list of such, or a tuple ``(x, y, z, azimuth, elevation)``; ``method``
    'cubic' (cubic B-spline in index space, the reference's ``maps.interp_spline_3d``) or
    'linear'. Receivers outside the grid or in its outermost cells give NaN.

    The interpolation runs on the device: the spline prefilter of a component is three passes
    of a recursive filter over the whole array -- ~0.2 s per 128^3 component with
    ``scipy.ndimage`` on the host, well under a millisecond here.

    Not in the reference: ``device_field`` -- a device tensor ``[fx | fy | fz]`` that holds the
    values (``field`` then only describes grid, dtype and kind): the field of a solve that is
    still in HBM (``parallel.compute(receivers=...)``), no 100 MB download."""
    if hasattr(receiver, 'coordinates'):
        coordinates = receiver.coordinates
    elif hasattr(tuple(receiver)[0], 'coordinates'):
        coordinates = tuple(np.array([r.coordinates for r in receiver], dtype=float).T)
    else:
        coordinates = receiver
        if len(coordinates) != 5:
            raise ValueError("`receiver` needs to be in the form (x, y, z, azimuth, elevation). "
                             f"Length of provided `receiver`: {len(coordinates)}.")
    if method not in ('cubic', 'linear'):
        raise ValueError(f"Method {method!r} is not defined; 'cubic' or 'linear'.")
    torch, _lib, _ptr, _stream, dev = _device_tools()
    data = field if device_field is None else device_field       # device tensor [fx | fy | fz]
    grid = field.grid
    x, y, z = np.broadcast_arrays(*[np.asarray(c, dtype=float) for c in coordinates[:3]])
    shape = x.shape
    xi = np.stack([x.ravel('F'), y.ravel('F'), z.ravel('F')], axis=1)
    npts = xi.shape[0]
    factors = _rotation(np.asarray(coordinates[3], dtype=float), np.asarray(coordinates[4], dtype=float))
    factors = [np.broadcast_to(f, shape).ravel('F') for f in factors]
    resp = np.zeros(npts, dtype=field.field.dtype)
    is_complex = int(np.iscomplexobj(resp))
    lib = _lib.lib()
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)                   # noqa: E731
    offsets = np.cumsum([0] + list(field._sizes))
    for c in range(3):
        if not np.any(abs(factors[c]) > 1e-10):
            continue
        n0, n1, n2 = field._shapes[c]
        pts = _component_points(grid, field._shapes[c])
        if device_field is None:
            dvals = up(field._component(c).ravel('F'))
        else:                          # the spline filter works in place: on a copy
            dvals = data[int(offsets[c]):int(offsets[c + 1])].clone()
        out = torch.empty(npts, dtype=dvals.dtype, device=dev)
        if method == 'cubic':
            from scipy.interpolate import interp1d
            coords = np.empty((3, npts))
            for d in range(3):          # metres -> index space, as maps.interp_spline_3d (maps.py:545-550)
                coords[d] = interp1d(pts[d], np.arange(len(pts[d])), kind='cubic', bounds_error=False,
                                     fill_value='extrapolate')(xi[:, d])
            dcoords = up(coords)
            _lib.check(lib.emg3d_dev_spline_filter(_ptr(dvals), n0, n1, n2, is_complex, _stream()),
                       'emg3d_dev_spline_filter')
            _lib.check(lib.emg3d_dev_spline_eval(_ptr(dvals), n0, n1, n2, is_complex, _ptr(dcoords), npts,
                                                 _ptr(out), _stream()), 'emg3d_dev_spline_eval')
        else:
            idx = np.empty((3, npts), dtype=np.int32)
            w = np.empty((3, npts))
            for d in range(3):          # scipy RegularGridInterpolator._find_indices
                g = pts[d]
                i = np.searchsorted(g, xi[:, d]) - 1
                i[i < 0] = 0
                i[i > g.size - 2] = g.size - 2
                w[d] = (xi[:, d] - g[i]) / (g[i + 1] - g[i])
                inside = (xi[:, d] >= g[0]) & (xi[:, d] <= g[-1])
                idx[d] = np.where(inside, i, -1)
            didx, dw = up(idx), up(w)
            _lib.check(lib.emg3d_dev_linear_eval(_ptr(dvals), n0, n1, n2, is_complex, _ptr(didx), _ptr(dw),
                                                 npts, _ptr(out), _stream()), 'emg3d_dev_linear_eval')
        resp += factors[c] * out.cpu().numpy()
    # PEC: receivers in the outermost cells are set to NaN (fields.py:604-610)
    ind = ((xi[:, 0] < grid.nodes_x[1]) | (xi[:, 0] > grid.nodes_x[-2]) |
           (xi[:, 1] < grid.nodes_y[1]) | (xi[:, 1] > grid.nodes_y[-2]) |
           (xi[:, 2] < grid.nodes_z[1]) | (xi[:, 2] > grid.nodes_z[-2]))
    resp[ind] = np.nan
    return resp.reshape(shape, order='F')

"""CPU restatement (numpy) of the receiver extraction of the reference -- TEST INFRASTRUCTURE.

The reference's ``fields.get_receiver`` (emg3d/fields.py:522-614) interpolates each field
component with ``maps.interpolate`` (emg3d/maps.py:232-368):

* ``method='cubic'`` -> ``interp_spline_3d`` (maps.py:500-552): the coordinates are mapped to
  index space with ``scipy.interpolate.interp1d(kind='cubic', fill_value='extrapolate')`` per
  dimension, then ``scipy.ndimage.map_coordinates(values, coords, order=3, mode='constant',
  cval=nan)``. The algorithm lives in SciPy (1.15.3 here; ndimage/src/ni_splines.c,
  ni_interpolation.c), not in the reference; restated below from its published behaviour and
  pinned against SciPy itself (tests/test_oracle.py) and against the reference's outputs
  (tests/golden/receivers.npz):
    - prefilter: separable cubic B-spline recursive filter, pole z = sqrt(3) - 2, gain 6,
      exact mirror initialisation over the whole line (mode 'constant' filters like 'mirror');
    - evaluation: points with an index coordinate outside [0, n-1] give cval; inside, the
      four-point support start = floor(x) - 1 is folded back by mirroring at 0 and n-1.
* ``method='linear'`` -> ``scipy.interpolate.RegularGridInterpolator(bounds_error=False,
  fill_value=nan)``: trilinear in the cell that contains the point; NaN outside.
"""
import numpy as np

POLE = np.sqrt(3.0) - 2.0


def spline_filter_axis(c, axis):
    """In-place cubic B-spline prefilter of a real array along one axis (mirror)."""
    c = np.moveaxis(c, axis, 0)
    n = c.shape[0]
    if n < 2:
        return
    z = POLE
    c *= (1.0 - z) * (1.0 - 1.0 / z)
    # causal initialisation, mirror: sum over the mirrored, periodic extension
    z_n_1 = z ** (n - 1)
    c0 = c[0] + z_n_1 * c[n - 1]
    z_i = z
    for i in range(1, n - 1):
        c0 = c0 + z_i * (c[i] + z_n_1 * c[n - 1 - i])
        z_i *= z
    c[0] = c0 / (1.0 - z_n_1 * z_n_1)
    for i in range(1, n):
        c[i] += z * c[i - 1]
    c[n - 1] = (z * c[n - 2] + c[n - 1]) * z / (z * z - 1.0)
    for i in range(n - 2, -1, -1):
        c[i] = z * (c[i + 1] - c[i])


def spline_filter(values):
    """Cubic B-spline coefficients of a real or complex 3-D array (all three axes)."""
    values = np.asarray(values)
    if np.iscomplexobj(values):
        return spline_filter(values.real) + 1j * spline_filter(values.imag)
    c = np.array(values, dtype=np.float64, order='C')
    for axis in range(c.ndim):
        spline_filter_axis(c, axis)
    return c


def _weights(x):
    """Start index and the four cubic B-spline weights for coordinate x."""
    start = int(np.floor(x)) - 1
    t = x - np.floor(x)
    w = np.empty(4)
    w[0] = (1.0 - t) ** 3 / 6.0
    w[1] = (3.0 * t ** 3 - 6.0 * t ** 2 + 4.0) / 6.0
    w[2] = (-3.0 * t ** 3 + 3.0 * t ** 2 + 3.0 * t + 1.0) / 6.0
    w[3] = t ** 3 / 6.0
    return start, w


def _mirror(i, n):
    if n == 1:
        return 0
    p = 2 * (n - 1)
    i = abs(i) % p
    return p - i if i >= n else i


def map_coordinates_cubic(values, coords, cval=np.nan):
    """scipy.ndimage.map_coordinates(values, coords, order=3, mode='constant', cval=cval)."""
    coef = spline_filter(values)
    shape = coef.shape
    coords = np.asarray(coords, dtype=float)
    out = np.empty(coords.shape[1], dtype=coef.dtype)
    for p in range(coords.shape[1]):
        x = coords[:, p]
        if any((not np.isfinite(x[d])) or x[d] < 0 or x[d] > shape[d] - 1 for d in range(3)):
            out[p] = cval
            continue
        sw = [_weights(x[d]) for d in range(3)]
        acc = 0.0
        for a in range(4):
            ia = _mirror(sw[0][0] + a, shape[0])
            for b in range(4):
                ib = _mirror(sw[1][0] + b, shape[1])
                for c in range(4):
                    ic = _mirror(sw[2][0] + c, shape[2])
                    acc = acc + sw[0][1][a] * sw[1][1][b] * sw[2][1][c] * coef[ia, ib, ic]
        out[p] = acc
    return out


def interp_linear(points, values, xi, fill=np.nan):
    """RegularGridInterpolator(points, values, method='linear', bounds_error=False,
    fill_value=fill)(xi) for 3-D data."""
    xi = np.asarray(xi, dtype=float)
    out = np.empty(xi.shape[0], dtype=np.result_type(values.dtype, float))
    for p in range(xi.shape[0]):
        idx, w, inside = [], [], True
        for d in range(3):
            g = points[d]
            x = xi[p, d]
            if not (g[0] <= x <= g[-1]):
                inside = False
                break
            i = int(np.searchsorted(g, x)) - 1
            i = min(max(i, 0), g.size - 2)
            idx.append(i)
            w.append((x - g[i]) / (g[i + 1] - g[i]))
        if not inside:
            out[p] = fill
            continue
        acc = 0.0
        for a in (0, 1):
            for b in (0, 1):
                for c in (0, 1):
                    wt = (w[0] if a else 1 - w[0]) * (w[1] if b else 1 - w[1]) * (w[2] if c else 1 - w[2])
                    acc = acc + wt * values[idx[0] + a, idx[1] + b, idx[2] + c]
        out[p] = acc
    return out


# ---- model re-gridding: volume averaging (emg3d/maps.py:555-664) -------------------------
def volume_average_weights(x_i, x_o):
    """maps._volume_average_weights (maps.py:619-664): the union of input and output nodes cuts
    the axis into segments; kept are the segments whose centre lies inside the OUTPUT grid, each
    with its length, the input cell (clamped: nearest extrapolation) and the output cell."""
    xs = np.unique(np.concatenate((x_i, x_o)))
    n1, n2 = len(x_i), len(x_o)
    w, ii, io = [], [], []
    i1 = i2 = 0
    for i in range(len(xs) - 1):
        center = 0.5 * (xs[i] + xs[i + 1])
        if x_o[0] <= center <= x_o[n2 - 1]:
            while i1 < n1 - 1 and center >= x_i[i1]:
                i1 += 1
            while i2 < n2 - 1 and center >= x_o[i2]:
                i2 += 1
            w.append(xs[i + 1] - xs[i])
            ii.append(min(max(i1 - 1, 0), n1 - 1))
            io.append(min(max(i2 - 1, 0), n2 - 1))
    return np.array(w), np.array(ii, dtype=np.int32), np.array(io, dtype=np.int32)


def volume_average(nodes, values, new_nodes):
    """maps.interp_volume_average (maps.py:555-616): sum of (wz wy) wx v over the segments of
    every output cell, in the reference's z, y, x order, divided by the output cell volume."""
    (wx, ixi, ixo), (wy, iyi, iyo), (wz, izi, izo) = [volume_average_weights(a, b) for a, b in zip(nodes, new_nodes)]
    shape = tuple(len(n) - 1 for n in new_nodes)
    out = np.zeros(shape)
    for a, w_z in enumerate(wz):
        for b, w_y in enumerate(wy):
            w_zy = w_z * w_y
            for c, w_x in enumerate(wx):
                out[ixo[c], iyo[b], izo[a]] += w_zy * w_x * values[ixi[c], iyi[b], izi[a]]
    vol = (np.diff(new_nodes[0])[:, None, None] * np.diff(new_nodes[1])[None, :, None] *
           np.diff(new_nodes[2])[None, None, :])
    return out / vol

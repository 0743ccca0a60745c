"""Tensor-product grid: the attribute set the multigrid path uses.

Same names and meaning as the reference's ``emg3d.meshes.BaseMesh`` / ``TensorMesh``
(reference emg3d/meshes.py:42-134), restated; gridding helpers are out of scope.
"""
import numpy as np

__all__ = ['BaseMesh', 'TensorMesh']


class BaseMesh:
    """Minimal tensor mesh defined by cell widths ``h = [hx, hy, hz]`` and ``origin``."""

    def __init__(self, h, origin, **kwargs):
        self.origin = np.array(origin, dtype=float)
        self.h = [np.array(h[0], dtype=float), np.array(h[1], dtype=float),
                  np.array(h[2], dtype=float)]
        nx, ny, nz = (w.size for w in self.h)

        self.shape_cells = (nx, ny, nz)
        self.shape_nodes = (nx + 1, ny + 1, nz + 1)
        self.n_cells = nx * ny * nz
        self.nodes_x = np.r_[0., self.h[0].cumsum()] + self.origin[0]
        self.nodes_y = np.r_[0., self.h[1].cumsum()] + self.origin[1]
        self.nodes_z = np.r_[0., self.h[2].cumsum()] + self.origin[2]
        self.cell_centers_x = (self.nodes_x[1:] + self.nodes_x[:-1]) / 2
        self.cell_centers_y = (self.nodes_y[1:] + self.nodes_y[:-1]) / 2
        self.cell_centers_z = (self.nodes_z[1:] + self.nodes_z[:-1]) / 2

        self.shape_edges_x = (nx, ny + 1, nz + 1)
        self.shape_edges_y = (nx + 1, ny, nz + 1)
        self.shape_edges_z = (nx + 1, ny + 1, nz)
        self.n_edges_x = nx * (ny + 1) * (nz + 1)
        self.n_edges_y = (nx + 1) * ny * (nz + 1)
        self.n_edges_z = (nx + 1) * (ny + 1) * nz
        self.n_edges = self.n_edges_x + self.n_edges_y + self.n_edges_z

        self.shape_faces_x = (nx + 1, ny, nz)
        self.shape_faces_y = (nx, ny + 1, nz)
        self.shape_faces_z = (nx, ny, nz + 1)
        self.n_faces_x = (nx + 1) * ny * nz
        self.n_faces_y = nx * (ny + 1) * nz
        self.n_faces_z = nx * ny * (nz + 1)
        self.n_faces = self.n_faces_x + self.n_faces_y + self.n_faces_z
        self._cell_volumes = None

    def __repr__(self):
        return (f"TensorMesh: {self.shape_cells[0]} x {self.shape_cells[1]} x "
                f"{self.shape_cells[2]} ({self.n_cells:,})")

    def __eq__(self, mesh):
        if not isinstance(mesh, BaseMesh) or self.shape_cells != mesh.shape_cells:
            return False
        return bool(np.allclose(self.origin, mesh.origin, atol=0) and
                    all(np.allclose(a, b, atol=0) for a, b in zip(self.h, mesh.h)))

    @property
    def cell_volumes(self):
        """Cell volumes as 1-D array, x fastest."""
        if self._cell_volumes is None:
            self._cell_volumes = (self.h[0][None, None, :] * self.h[1][None, :, None] *
                                  self.h[2][:, None, None]).ravel()
        return self._cell_volumes

    def copy(self):
        return self.__class__([w.copy() for w in self.h], self.origin.copy())

    def to_dict(self, copy=False):
        out = {'hx': self.h[0], 'hy': self.h[1], 'hz': self.h[2], 'origin': self.origin,
               '__class__': self.__class__.__name__}
        return {k: (v.copy() if copy and hasattr(v, 'copy') else v) for k, v in out.items()}

    @classmethod
    def from_dict(cls, inp):
        return cls([inp['hx'], inp['hy'], inp['hz']], inp['origin'])


class TensorMesh(BaseMesh):
    """Name used by the reference's public API (``emg3d.TensorMesh``)."""

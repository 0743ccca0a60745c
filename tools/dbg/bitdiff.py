"""Where do the batched / staged line kernels differ from the single-source ones? (debug, through gpurun)"""
import sys, os
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np, torch
from emg3d_amd import _lib
from emg3d_amd._device import DeviceLevel
from oracle import mg_ref
from test_gpu_parity import _random_level_fields
lib = _lib.lib()
shape, lr = (130, 72, 72), 1
if len(sys.argv) > 1:
    shape = tuple(int(x) for x in sys.argv[1].split(',')); lr = int(sys.argv[2])
nu = int(os.environ.get('NU', 1))
grid, vm, s0, e0 = _random_level_fields(shape, complex, 5)
dev = torch.device('cuda')
single = DeviceLevel.from_host(vm, dev)
n = e0.field.size
def run1(mode):
    lib.emg3d_set_option(b'line_stream', mode)
    single.s.copy_(torch.from_numpy(s0.field)); single.e.copy_(torch.from_numpy(e0.field))
    single.smooth(lr, nu)
    return single.e.cpu().numpy(), lib.emg3d_line_kernel_name(lr, *shape, 1, 1).decode()
res = {m: run1(m) for m in (0, 2, 4)}
lib.emg3d_set_option(b'line_stream', 1)
for m in (2, 4):
    d = res[m][0] != res[0][0]
    print('B=1 line_stream', m, res[m][1], 'vs', res[0][1], 'differing entries', d.sum(), 'of', n,
          'max rel', np.abs(res[m][0] - res[0][0]).max() / np.abs(res[0][0]).max())
for B in (2, 4):
    many = DeviceLevel.from_host(vm, dev, batch=B)
    many._factors = single._factors
    many.s.copy_(torch.from_numpy(np.tile(s0.field, B))); many.e.copy_(torch.from_numpy(np.tile(e0.field, B)))
    many.smooth(lr, nu)
    got = many.e.cpu().numpy().reshape(B, n)
    for b in range(B):
        d = got[b] != res[0][0]
        o1, o2 = grid.n_edges_x, grid.n_edges_x + grid.n_edges_y
        print('B=%d source %d' % (B, b), lib.emg3d_line_kernel_name(lr, *shape, 1, B).decode(), 'differing', d.sum(), 'x/y/z', d[:o1].sum(), d[o1:o2].sum(), d[o2:].sum(),
              'max rel', np.abs(got[b] - res[0][0]).max() / np.abs(res[0][0]).max(), 'same as source 0:', np.array_equal(got[b], got[0]))
    if B == 2:
        # where along the line do the x-components differ?
        dx = (got[0][:o1] != res[0][0][:o1]).reshape(grid.shape_edges_x, order='F')
        print('  differing ex entries per ix (first 140):', dx.sum(axis=(1, 2))[:140].tolist())

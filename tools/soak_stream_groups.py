"""Soak of k_line_stream with groups of right-hand sides (through gpurun): random level shapes with one long direction
(64 ... 300 blocks, enough lines per colour class for 16-line workgroups), random batch sizes 2 ... 8, complex and real
fields, with and without epsilon_r / mu_r, nu = 1 ... 3: every right-hand side of the batched level must equal, bit for
bit, the same right-hand side swept alone on a single-source level (which runs k_line_colour or k_line_stream<B = 1>).
    python tools/soak_stream_groups.py [cases] [seed]           (profiles/r04_soak_stream_groups.txt)"""
import os, sys
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np
import torch
from emg3d_amd import _lib
from emg3d_amd._device import DeviceLevel
from oracle import mg_ref

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = _lib.lib()
dev = torch.device('cuda')
bad = 0
kinds = {}
for case in range(ncases):
    rng = np.random.default_rng(seed0 + case)
    lr = int(rng.integers(1, 4))
    long_n = int(rng.choice([64, 66, 96, 100, 128, 130, 160, 200, 256, 258, 300]))
    batch = int(rng.integers(2, 9))
    # enough lines per class that the launcher takes 16 per workgroup: cdiv(lines, 8) * batch > 256
    need = 2048 // batch + 64
    t = int(np.ceil(np.sqrt(4 * need))) + 2
    t1, t2 = t + int(rng.integers(0, 6)), t + int(rng.integers(0, 6))
    shape = [t1, t2]
    shape.insert(lr - 1, long_n)
    shape = tuple(shape)
    dtype = complex if rng.random() < 0.75 else float
    extras = rng.random() < 0.3
    h = [rng.uniform(5., 15., n) * 1.02 ** np.abs(np.arange(n) - n // 2) for n in shape]
    grid = mg_ref.Grid(h, (0., 0., 0.))
    sig = [10 ** rng.uniform(-1, 1, shape) for _ in range(3)]
    kw = dict(mu_r=rng.uniform(0.8, 2.0, shape), epsilon_r=rng.uniform(1., 80., shape)) if extras else {}
    f = (2e5 if extras else 0.7) * (1 if dtype is complex else -1)
    vm = mg_ref.volume_model(grid, f, *sig, **kw)
    n = grid.n_edges
    nu = int(rng.integers(1, 4))

    def field():
        v = rng.standard_normal(n) + (1j * rng.standard_normal(n) if dtype is complex else 0)
        return v.astype(dtype)
    srcs = [field() for _ in range(batch)]
    single = DeviceLevel.from_host(vm, dev)
    starts = []
    for b in range(batch):
        single.e.copy_(torch.from_numpy(field())); single.pec_zero(); starts.append(single.e.cpu().numpy())
    want = []
    for b in range(batch):
        single.s.copy_(torch.from_numpy(srcs[b])); single.e.copy_(torch.from_numpy(starts[b]))
        single.smooth(lr, nu)
        want.append(single.e.cpu().numpy())
    many = DeviceLevel.from_host(vm, dev, batch=batch)
    many._factors = single._factors
    many.s.copy_(torch.from_numpy(np.concatenate(srcs))); many.e.copy_(torch.from_numpy(np.concatenate(starts)))
    many.smooth(lr, nu)
    got = many.e.cpu().numpy().reshape(batch, n)
    k1 = lib.emg3d_line_kernel_name(lr, *shape, int(dtype is complex), 1).decode()
    kb = lib.emg3d_line_kernel_name(lr, *shape, int(dtype is complex), batch).decode()
    kinds[(k1, kb)] = kinds.get((k1, kb), 0) + 1
    ok = all(np.array_equal(got[b], want[b]) for b in range(batch)) and all(np.any(want[b] != starts[b]) for b in range(batch))
    bad += not ok
    print(f"case {case:3d} shape {shape} lr {lr} batch {batch} {dtype.__name__:7s} extras {int(extras)} nu {nu} "
          f"single {k1} batched {kb}: {'bit-identical' if ok else 'DIFFERENT'}", flush=True)
    del single, many
    torch.cuda.empty_cache()
print(f"# {ncases} cases, {bad} different; (single-source kernel, batched kernel) -> cases: {kinds}")

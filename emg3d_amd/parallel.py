"""Sharding of independent (source, frequency) solves over the GPUs of one node.

The reference parallelises exclusively over source-frequency pairs with a process pool
(reference emg3d/_multiprocessing.py:33-153, emg3d/simulations.py:835-880, 1453-1464): one
pair = one complete ``solve``; nothing is exchanged inside a solve. The MI355X equivalent:

* one process per GPU (``python -m torch.distributed.run --nproc-per-node 8 ...``);
* the model's property arrays are broadcast ONCE from rank 0 (RCCL over xGMI on GPUs; gloo
  on CPU for the tests); every rank builds eta(f), zeta and its sources locally;
* pairs are assigned statically (longest-processing-time first when costs are known,
  round-robin otherwise); each rank runs its pairs one after the other on its GPU;
* small per-pair results (info dicts, receiver responses) are gathered on request; the
  fields stay where they were computed.

There is no collective inside a solve and no all-reduce on the forward path.
"""
import itertools
import os

import numpy as np

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')    # dmabuf IPC for RCCL on these hosts

__all__ = ['init', 'finalize', 'srcfreq_pairs', 'shard', 'broadcast_model', 'solve',
           'compute', 'gather_objects']


def _dist():
    import torch.distributed as dist
    return dist


def init(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    (as set by torch.distributed.run). Returns (rank, world, device).

    backend: 'nccl' (= RCCL on ROCm) when a GPU is visible, else 'gloo'."""
    import torch
    dist = _dist()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    use_gpu = torch.cuda.is_available()
    if backend is None:
        backend = 'nccl' if use_gpu else 'gloo'
    device = torch.device('cpu')
    if use_gpu and backend == 'nccl':
        # (more ranks than visible devices -- a dry run of the multi-rank path on a smaller box -- share them)
        local = int(os.environ.get('LOCAL_RANK', rank)) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        device = torch.device('cuda', local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29512')
        kw = {'device_id': device} if device.type == 'cuda' else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, device


def finalize():
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def rank_and_world():
    """(rank, world size) of the process group; (0, 1) without one."""
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def srcfreq_pairs(sources, frequencies):
    """All (source, frequency) keys in the reference's order: product(sources, frequencies)
    (emg3d/simulations.py:1453-1464)."""
    return list(itertools.product(list(sources), list(frequencies)))


def shard(n_items, rank, world, costs=None):
    """Indices of the items rank `rank` computes.

    Without costs: round-robin (item i -> rank i % world). With costs: longest-processing
    -time-first greedy assignment, deterministic and identical on every rank."""
    if costs is None:
        return list(range(rank, n_items, world))
    order = sorted(range(n_items), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world
    mine = []
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += float(costs[i])
        if r == rank:
            mine.append(i)
    return sorted(mine)


def _bcast_tensor(t, src, device):
    dist = _dist()
    t = t.to(device)
    dist.broadcast(t, src)
    return t


def broadcast_model(model, src=0, device=None):
    """Make rank `src`'s ``Model`` available on every rank.

    Metadata (grid widths, origin, mapping, which properties exist) goes as one small
    pickled object; each property array (float64, nx*ny*nz) is ONE broadcast of the raw
    buffer. `model` may be None on ranks != src. With a CUDA `device` the received arrays also
    stay in HBM (``_device_props`` of the RETURNED model, which ``VolumeModel.device_arrays``
    starts from): a snapshot of the properties at the time of the call -- the returned object is
    meant for the solves of that call (``compute``); replacing one of its property arrays drops
    the array's device copy, edits in place are not seen."""
    import torch
    from emg3d_amd import meshes, models
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    rank = dist.get_rank()
    device = device or torch.device('cpu')
    names = ('property_x', 'property_y', 'property_z', 'mu_r', 'epsilon_r')
    meta = [None]
    if rank == src:
        meta = [{'h': [h.copy() for h in model.grid.h], 'origin': model.grid.origin.copy(),
                 'mapping': model.mapping,
                 'present': [getattr(model, n) is not None for n in names]}]
    dist.broadcast_object_list(meta, src)
    meta = meta[0]
    grid = model.grid if rank == src else meshes.TensorMesh(meta['h'], meta['origin'])
    kw, dev = {}, {}
    on_gpu = device.type == 'cuda'
    for n, present in zip(names, meta['present']):
        if not present:
            continue
        if rank == src:
            t = torch.from_numpy(np.ascontiguousarray(getattr(model, n).ravel('F')))
        else:
            t = torch.empty(grid.n_cells, dtype=torch.float64)
        t = _bcast_tensor(t, src, device)
        if on_gpu:
            dev[n] = t            # stays in HBM: VolumeModel.device_arrays starts from it, no second upload
        if rank != src:           # the host-side Model of a receiving rank needs its own copy
            kw[n] = t.cpu().numpy().reshape(grid.shape_cells, order='F')
    if rank != src:
        model = models.Model(grid, mapping=meta['mapping'], **kw)
    else:
        # the caller's model is not touched: the HBM copies are a snapshot of its arrays NOW, and
        # an update of the model (an inversion step, an air layer written in place) followed by a
        # direct solve must see the host arrays. The snapshot travels on a shallow copy that
        # shares the arrays and lives as long as the caller keeps it (``compute``: one call).
        import copy
        model = copy.copy(model)
    model.__dict__['_device_props'] = dev
    return model


def solve(inp):
    """Worker: one source-frequency pair, the contract of the reference's
    ``_multiprocessing.solve`` (emg3d/_multiprocessing.py:72-153). ``inp`` has the keys
    [model, sfield, efield, solver_opts] or [model, grid, source, frequency, efield,
    solver_opts]; always returns (efield, info_dict)."""
    from emg3d_amd import fields, models, solver
    opts = dict(inp['solver_opts'])
    if 'sfield' in inp:
        sfield = inp['sfield']
        grid = sfield.grid
    else:
        grid = inp['grid']
        sfield = fields.get_source_field(grid, inp['source'], inp['frequency'])
        opts['_sparse_source'] = True     # made here, not modified: its few non-zeros go up, not 100 MB
    model = inp['model'].interpolate_to_grid(grid)
    cache = inp.get('hierarchies')
    if cache is not None:
        # pairs of one worker that share model, grid and frequency share the device-resident
        # levels, line factorisations and graphs (not in the reference: its workers are
        # separate processes that rebuild everything per pair)
        key = (id(inp['model']), id(grid), complex(sfield.sval))
        hit = cache.get(key)
        # an id can be recycled once its object is gone: the entry keeps the objects and is only
        # valid for the very same ones
        if hit is None or hit[1] is not inp['model'] or hit[2] is not grid:
            cache.pop(key, None)
            while len(cache) >= 2:            # bounded: a hierarchy is ~1.3 kB per cell
                cache.pop(next(iter(cache)))
            hit = cache[key] = (solver.Hierarchy(models.VolumeModel(model, sfield)), inp['model'], grid)
        opts['hierarchy'] = hit[0]
    rec = inp.get('receivers')
    if rec is None:
        return solver.solve(model=model, sfield=sfield, efield=inp.get('efield'),
                            return_info=True, always_return=True, **opts)
    # responses at the receivers, from the field while it is still in HBM when the solver keeps
    # it there (multigrid, BiCGSTAB): without `keep_field` nothing but the responses comes back
    on_device = opts.get('hierarchy') is not None and inp.get('efield') is None
    if on_device and not inp.get('keep_field', True):
        opts['_download'] = False
    efield, info = solver.solve(model=model, sfield=sfield, efield=inp.get('efield'),
                                return_info=True, always_return=True, **opts)
    dev_e = opts['hierarchy'].top.e if on_device else None
    info['responses'] = fields.get_responses(sfield if efield is None else efield, dev_e, rec,
                                             inp.get('receiver_method', 'cubic'), magnetic=inp.get('magnetic'),
                                             mu_r=getattr(model, 'mu_r', None), efield=None if on_device else efield)
    if opts.get('_download') is False:
        efield = None
    return efield, info



def gather_objects(obj, dst=0):
    """Gather one small picklable object per rank on `dst` (list there, None elsewhere)."""
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(obj, out, dst)
    return out


def compute(model, grid, sources, frequencies, solver_opts=None, costs=None, solve_fn=None,
            keep_fields=True, per_gpu=1, reuse=True, receivers=None, receiver_method='cubic', batch=1,
            magnetic=None):
    """Solve all source-frequency pairs, sharded over the ranks of the process group.

    model: on rank 0 (None elsewhere is fine; it is broadcast). sources: dict name ->
    source coordinates; frequencies: dict name -> Hz. Returns, on every rank, a dict
    {(src, freq): (efield or None, info)} for the pairs THIS rank computed; rank 0
    additionally gets key '_all_info': {(src, freq): info} gathered from all ranks.

    receivers: ``(x, y, z, azimuth, elevation)`` (any form ``fields.get_receiver`` takes), or a
    dict source name -> such; every pair's ``info['responses']`` then holds the field at the
    receivers, interpolated on the device from the solution while it is still in HBM. With
    ``keep_fields=False`` only the responses leave the GPU (no field download). ``magnetic``: boolean
    sequence (or dict source name -> such) marking the receivers that are magnetic point receivers:
    their responses are the magnetic field (``get_magnetic_field``, formed on the device from the
    solution) at the point [A/m]; pairs with magnetic receivers are not batched.

    batch > 1: up to that many of the rank's pairs that share a frequency are solved TOGETHER by
    ``solver.solve_batch`` (right-hand sides as one more grid dimension of every launch:
    multigrid: bit-identical fields; BiCGSTAB + multigrid: the Krylov iteration of every source
    with shared preconditioner / operator applications). cgs / gcrotmk: pair by pair.

    reuse: pairs of one worker with the same frequency share one device-resident level
    hierarchy (model, coarse levels, line factorisations, graphs); results are bit-identical.

    per_gpu > 1: the rank works on that many of its pairs at a time, each in its own host
    thread on its own HIP stream (the max_workers of the reference's process pool, but
    inside one GPU): a multigrid cycle leaves most of an MI355X idle on its coarse levels,
    and 3-4 concurrent solves raise the throughput by ~1.5x.
    """
    rank, world, device = init()
    model = broadcast_model(model, 0, device)
    pairs = srcfreq_pairs(sources, frequencies)
    mine = shard(len(pairs), rank, world, costs)
    solve_fn = solve_fn or solve
    if reuse:       # pairs of one frequency next to each other
        forder = {f: n for n, f in enumerate(frequencies)}
        mine = sorted(mine, key=lambda i: (forder[pairs[i][1]], i))
    out = {}

    the_grid = grid or model.grid

    def job(i, stream=None, hierarchies=None):
        s, f = pairs[i]
        inp = {'model': model, 'grid': the_grid, 'source': sources[s],
               'frequency': frequencies[f], 'efield': None, 'solver_opts': solver_opts or {}}
        if hierarchies is not None and solve_fn is solve:
            inp['hierarchies'] = hierarchies
        if receivers is not None:
            inp['receivers'] = receivers[s] if isinstance(receivers, dict) else receivers
            inp['receiver_method'] = receiver_method
            inp['keep_field'] = keep_fields
            if magnetic is not None:
                inp['magnetic'] = magnetic[s] if isinstance(magnetic, dict) else magnetic
        if stream is None:
            efield, info = solve_fn(inp)
        else:
            import torch
            with torch.cuda.stream(stream):
                efield, info = solve_fn(inp)
                stream.synchronize()
        return (s, f), (efield if keep_fields else None, info)

    # batched: multigrid, or BiCGSTAB + multigrid (the default of `solve`); cgs / gcrotmk run
    # pair by pair (on the device as well)
    if (batch > 1 and solve_fn is solve and magnetic is None and
            dict(solver_opts or {}).get('sslsolver', True) in (True, False, None, 'bicgstab') and
            dict(solver_opts or {}).get('cycle', 'F') is not None):
        from emg3d_amd import fields as _fields, solver as _solver
        opts = dict(solver_opts or {})
        opts.setdefault('sslsolver', True)
        by_freq, hiers = {}, {}
        for i in mine:
            by_freq.setdefault(pairs[i][1], []).append(i)
        the_model = model.interpolate_to_grid(the_grid)
        for f, idx in by_freq.items():
            for i0 in range(0, len(idx), batch):
                chunk = idx[i0:i0 + batch]
                sfs = []
                for i in chunk:
                    sf = _fields.get_source_field(the_grid, sources[pairs[i][0]], frequencies[f])
                    sf._trust_sparse = True
                    sfs.append(sf)
                rec = None
                if receivers is not None:
                    rec = [receivers[pairs[i][0]] if isinstance(receivers, dict) else receivers for i in chunk]
                hkey = (f, len(chunk))
                if reuse and hkey not in hiers:
                    hiers.clear()                 # one frequency / batch size at a time stays resident
                    from emg3d_amd import models as _models
                    hiers[hkey] = _solver.Hierarchy(_models.VolumeModel(the_model, sfs[0]), batch=len(chunk))
                res = _solver.solve_batch(the_model, sfs, receivers=rec, receiver_method=receiver_method,
                                          keep_fields=keep_fields, hierarchy=hiers.get(hkey), **opts)
                for i, (ef, info) in zip(chunk, res):
                    out[pairs[i]] = (ef, info)
    elif per_gpu <= 1 or len(mine) <= 1:
        hierarchies = {} if reuse else None
        for i in mine:
            k, v = job(i, None, hierarchies)
            out[k] = v
        del hierarchies
    else:
        import queue
        import threading
        todo = queue.Queue()
        for i in mine:
            todo.put(i)
        errors = []

        def worker():
            stream = None
            hierarchies = {} if reuse else None      # per thread: a hierarchy serves one solve at a time
            if device.type == 'cuda':
                import torch
                torch.cuda.set_device(device)
                stream = torch.cuda.Stream(device)
            while True:
                try:
                    i = todo.get_nowait()
                except queue.Empty:
                    return
                try:
                    k, v = job(i, stream, hierarchies)
                    out[k] = v
                except BaseException as exc:      # surfaced after the join
                    errors.append(exc)
                    return
        from emg3d_amd import _cycle
        threads = [threading.Thread(target=worker) for _ in range(min(per_gpu, len(mine)))]
        _cycle.CONCURRENT += 1            # no stream capture while several threads solve
        try:
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        finally:
            _cycle.CONCURRENT -= 1
        if errors:
            raise errors[0]
    small = {k: {kk: vv for kk, vv in v[1].items() if kk not in ('log',)} for k, v in out.items()}
    gathered = gather_objects(small, 0)
    if rank == 0:
        allinfo = {}
        for d in gathered:
            allinfo.update(d)
        out['_all_info'] = allinfo
    return out

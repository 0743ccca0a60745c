"""Multi-process path on CPU: world_size-2 gloo process group (the GPU job uses the same
code with backend nccl = RCCL). Checks model broadcast, static sharding of the
source-frequency pairs (every pair exactly once), the worker's input contract and the
gather of per-pair results. The solves themselves need a GPU; here the worker is replaced
by a stand-in that exercises everything up to the kernel calls."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    import emg3d_amd as emg3d
    from emg3d_amd import parallel
    try:
        r, w, device = parallel.init('gloo')
        assert (r, w) == (rank, world) and device.type == 'cpu'
        model = None
        h = [np.array([10., 12, 14, 16]), np.array([20., 20]), np.array([5., 6, 7, 8, 9, 10])]
        if rank == 0:
            grid = emg3d.TensorMesh(h, (1, 2, 3))
            rng = np.random.default_rng(0)
            model = emg3d.Model(grid, rng.uniform(1, 2, grid.shape_cells), None,
                                rng.uniform(2, 3, grid.shape_cells), mapping='Conductivity')
        sources = {f'S{i}': (12. + i, 10., 20., 0., 0.) for i in range(5)}
        freqs = {'f1': 0.5, 'f2': 1.0}

        def fake_solve(inp):
            # same input contract as parallel.solve / the reference's _multiprocessing.solve
            assert set(inp) == {'model', 'grid', 'source', 'frequency', 'efield', 'solver_opts'}
            sf = emg3d.get_source_field(inp['grid'], inp['source'], inp['frequency'])
            vm = emg3d.models.VolumeModel(inp['model'].interpolate_to_grid(inp['grid']), sf)
            info = {'exit': 0, 'it_mg': 3, 'chk': float(np.abs(vm.eta_z).sum() +
                                                        np.linalg.norm(sf.field)), 'log': 'x'}
            return sf, info
        out = parallel.compute(model, None, sources, freqs, {'sslsolver': False},
                               solve_fn=fake_solve)
        mine = sorted(k for k in out if k != '_all_info')
        res = {'rank': rank, 'mine': mine, 'case': None, 'all': None}
        # several pairs at a time per rank (host threads): same pairs, same results
        out3 = parallel.compute(model, None, sources, freqs, {'sslsolver': False},
                                solve_fn=fake_solve, per_gpu=3)
        res['threads_ok'] = (sorted(k for k in out3 if k != '_all_info') == mine and
                             all(out3[k][1]['chk'] == out[k][1]['chk'] for k in mine))
        if rank == 0:
            res['all'] = {k: v['chk'] for k, v in out['_all_info'].items()}
        # LPT sharding is deterministic and balanced
        costs = [5, 1, 1, 1, 4, 1, 3, 1, 1, 2]
        res['lpt'] = parallel.shard(10, rank, world, costs)
        # broadcast result usable on every rank
        m2 = parallel.broadcast_model(model if rank == 0 else None, 0)
        res['case'] = m2.case
        res['sum'] = float(m2.property_x.sum() + m2.property_z.sum())
        res['mapping'] = m2.mapping
        q.put(res)
        parallel.finalize()
    except Exception as e:   # pragma: no cover
        q.put({'rank': rank, 'error': repr(e)})
        raise


@pytest.mark.timeout(300)
def test_two_process_gloo_sharding_and_broadcast():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all('error' not in r for r in results), results
    assert all(r['threads_ok'] for r in results)
    res = {r['rank']: r for r in results}
    pairs = [(f'S{i}', f) for i in range(5) for f in ('f1', 'f2')]
    assert sorted(res[0]['mine'] + res[1]['mine']) == sorted(pairs)      # each pair once
    assert not set(res[0]['mine']) & set(res[1]['mine'])
    assert len(res[0]['mine']) == len(res[1]['mine']) == 5
    assert set(res[0]['all']) == set(pairs)                              # gathered on rank 0
    assert res[0]['case'] == res[1]['case'] == 'VTI'
    assert res[0]['sum'] == res[1]['sum'] and res[1]['mapping'] == 'Conductivity'
    lpt0, lpt1 = res[0]['lpt'], res[1]['lpt']
    costs = [5, 1, 1, 1, 4, 1, 3, 1, 1, 2]
    assert sorted(lpt0 + lpt1) == list(range(10))
    assert abs(sum(costs[i] for i in lpt0) - sum(costs[i] for i in lpt1)) <= 1


def _worker8(rank, world, port, q):
    """One rank of the real rank count of BASELINE.json configs 4 and 5 (8 GPUs of one node): the product's
    process-group setup, the model broadcast, and the pair each rank takes."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    import emg3d_amd as emg3d
    from emg3d_amd import parallel
    from bench import PAIR_COSTS
    try:
        r, w, device = parallel.init('gloo')
        assert (r, w) == (rank, world)
        model = None
        if rank == 0:
            grid = emg3d.TensorMesh([np.full(6, 10.), np.full(4, 20.), np.full(4, 5.)], (0, 0, 0))
            model = emg3d.Model(grid, np.random.default_rng(1).uniform(1, 2, grid.shape_cells))
        m2 = parallel.broadcast_model(model, 0)
        costs = PAIR_COSTS['salt384']
        sources = {f'S{i}': (12. + i, 10., 20., 0., 0.) for i in range(8)}

        def fake_solve(inp):
            sf = emg3d.get_source_field(inp['grid'], inp['source'], inp['frequency'])
            return sf, {'exit': 0, 'it_mg': 1, 'chk': float(np.linalg.norm(sf.field)), 'log': ''}
        out = parallel.compute(model, None, sources, {'f': 1.0}, {'sslsolver': False}, solve_fn=fake_solve)
        q.put({'rank': rank, 'sum': float(m2.property_x.sum()), 'case': m2.case,
               'config5': parallel.shard(len(costs), rank, world, costs),      # LPT: cycles to tolerance by frequency
               'config4': parallel.shard(8, rank, world),                      # round-robin: one source per GPU
               'mine': sorted(k for k in out if k != '_all_info'),
               'all': sorted(out['_all_info']) if rank == 0 else None})
        parallel.finalize()
    except Exception as e:   # pragma: no cover
        q.put({'rank': rank, 'error': repr(e)})
        raise


@pytest.mark.timeout(600)
def test_eight_process_gloo_one_pair_per_rank():
    """The multi-rank path at the REAL rank count (world size 8, gloo on CPU): the model reaches all eight ranks,
    config 5's eight (source, frequency) pairs go one to a rank under the longest-processing-time sharding with
    bench.PAIR_COSTS (ranks 0 / 1 take the two 0.25 Hz pairs, ...), config 4's eight sources one to a rank
    round-robin, `parallel.compute` solves every pair exactly once and rank 0 gathers all eight infos."""
    world = 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all('error' not in r for r in results), results
    res = {r['rank']: r for r in results}
    assert len({res[r]['sum'] for r in range(world)}) == 1 and all(res[r]['case'] == 'isotropic' for r in range(world))
    assert [res[r]['config5'] for r in range(world)] == [[i] for i in range(world)]    # costs are sorted: pair r -> rank r
    assert [res[r]['config4'] for r in range(world)] == [[i] for i in range(world)]
    assert all(len(res[r]['mine']) == 1 for r in range(world))
    assert sorted(sum((res[r]['mine'] for r in range(world)), [])) == sorted((f'S{i}', 'f') for i in range(8))
    assert res[0]['all'] == sorted((f'S{i}', 'f') for i in range(8))


def test_shard_round_robin_and_pairs_order():
    from emg3d_amd import parallel
    assert parallel.srcfreq_pairs(['a', 'b'], [1, 2]) == [('a', 1), ('a', 2), ('b', 1), ('b', 2)]
    got = [parallel.shard(8, r, 8) for r in range(8)]
    assert got == [[i] for i in range(8)]                 # config 4: one source per GPU
    got = [parallel.shard(10, r, 4) for r in range(4)]
    assert sorted(sum(got, [])) == list(range(10))

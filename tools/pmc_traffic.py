"""Turn the two PMC passes of a bench run into profiles/<round>_pmc_traffic.json (round: env PMC_ROUND, default r05).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/f -o run -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/w -o run -- python bench.py ...
    python tools/pmc_traffic.py WORKLOAD out/f/run_counter_collection.csv out/w/run_counter_collection.csv

Per kernel the launches with the LARGEST grid (= the finest level) are averaged. Units and
corrections as MI355X_MICROARCH.md (HBM section) prescribes: the counters are KiB; on gfx950
FETCH_SIZE reports half of the bytes of wide streaming reads and is doubled; WRITE_SIZE is
taken as is (calibrated exact against a torch fill in profiles/r01_v4_pmc_hbm_traffic_256.txt).
The bench names a level-0 smoother call after its direction; its bytes are the sum of the
kernels one colour pass launches (fused: k_line_colour; else rhs + forward + backward).
"""
import csv
import json
import os
import sys
from collections import defaultdict


def per_kernel(path, counter):
    rows = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] != counter:
                continue
            grid = int(r['Grid_Size']) if 'Grid_Size' in r else int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1))
            rows[r['Kernel_Name']].append((grid, float(r['Counter_Value'])))
    out = {}
    for k, v in rows.items():
        g = max(x[0] for x in v)
        vals = [x[1] for x in v if x[0] == g]
        out[k] = (g, sum(vals) / len(vals), len(vals))
    return out


def main():
    wl, fpath, wpath = sys.argv[1:4]
    fetch, write = per_kernel(fpath, 'FETCH_SIZE'), per_kernel(wpath, 'WRITE_SIZE')
    res = {}
    for d, name in ((0, 'k_gs_line<x>'), (1, 'k_gs_line<y>'), (2, 'k_gs_line<z>')):
        parts = {}
        for k in fetch:
            for tag in (f'k_line_colour<emg::cplx, {d},', f'k_line_stream<emg::cplx, {d},', f'k_line_backward<emg::cplx, {d}>',
                        f'k_line_rhs<emg::cplx, {d}>', 'k_line_rhs_xt<emg::cplx>' if d == 0 else None):
                if tag and tag in k:
                    cand = {'read': 2 * fetch[k][1] * 1024, 'write': write[k][1] * 1024,
                            'grid': fetch[k][0], 'launches': fetch[k][2], 'kernel': k.split('(emg::Level')[0][-40:]}
                    # several instantiations (LDS modes) share a tag: the finest level's moves most
                    if tag not in parts or cand['read'] + cand['write'] > parts[tag]['read'] + parts[tag]['write']:
                        parts[tag] = cand
        if not parts:
            continue
        gmax = max(p['grid'] for p in parts.values())
        # fused and unfused kernels never serve the same level: keep the finest level's set
        # the finest level is served either by k_line_stream (384-thread workgroups) or by k_line_colour
        stream = {k: p for k, p in parts.items() if 'stream' in k}
        fused = stream or {k: p for k, p in parts.items() if 'colour' in k}
        use = fused if fused and max(p['grid'] for p in fused.values()) * 2 >= gmax else \
            {k: p for k, p in parts.items() if 'colour' not in k and 'stream' not in k}
        total = sum(p['read'] + p['write'] for p in use.values())
        res[name] = {'bytes_per_launch': total, 'kernels': use}
    pt = [k for k in fetch if 'k_gs_point_tile' in k]
    if pt:
        k = max(pt, key=lambda k: fetch[k][1])
        res['k_gs_point_tile'] = {'bytes_per_launch': 2 * fetch[k][1] * 1024 + write.get(k, (0, 0, 0))[1] * 1024,
                                  'kernels': {'k_gs_point_tile': {'read': 2 * fetch[k][1] * 1024, 'write': write.get(k, (0, 0, 0))[1] * 1024,
                                                                  'grid': fetch[k][0], 'launches': fetch[k][2]}}}
    rnd = os.environ.get('PMC_ROUND', 'r05')
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles', f'{rnd}_pmc_traffic.json')
    try:
        with open(path) as f:
            allres = json.load(f)
    except OSError:
        allres = {}
    allres[wl] = res
    import datetime
    meta = allres.setdefault('_meta', {})
    meta['date'] = datetime.date.today().isoformat()
    meta['passes'] = 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, one pass each, tools/profile_bench.sh'
    if os.environ.get('PMC_COMMIT'):
        meta['library_commit'] = os.environ['PMC_COMMIT']
    # what the counters were measured on: the hash of the library's sources (bench.py flags a summary whose hash is not
    # that of the sources it runs with -- there is no git on the GPU box)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from bench import csrc_sha16
    meta['csrc_sha16'] = csrc_sha16()
    with open(path, 'w') as f:
        json.dump(allres, f, indent=1, sort_keys=True)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()

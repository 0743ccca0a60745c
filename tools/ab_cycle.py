"""Same-box A/B of two builds of the library on the bench's timed region (through gpurun): the boxes of the pool
differ by up to 8 % with one binary, so a change is only visible when both builds run on ONE box, interleaved.
    python tools/ab_cycle.py emg3d_amd/lib/libemg3d_amd_r03.so [workload] [rounds]
The other build is loaded in a child process with the entry points it lacks stubbed (older builds have no
emg3d_options_generation / emg3d_line_kernel_name). Built libraries are not in the history; an older one is made from
its commit, e.g. the library of the end of round 3:
    mkdir /tmp/r03 && git archive 57cc31a emg3d_amd/csrc include | tar x -C /tmp/r03 && \
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared /tmp/r03/emg3d_amd/csrc/kernels.hip -o emg3d_amd/lib/libemg3d_amd_r03.so
AB_VARIANTS="opt=value;opt2=value ..." adds runs of the current build under library options."""
import json, os, subprocess, sys
root = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r'''
import os, sys, json, ctypes, time
root, libpath, wlname, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
sys.path.insert(0, root)
from emg3d_amd import _lib
if libpath != 'default':
    _lib.LIBPATH = os.path.join(root, libpath)
    cd = ctypes.CDLL(_lib.LIBPATH)
    for name in list(_lib.SIGNATURES):
        if not hasattr(cd, name):
            del _lib.SIGNATURES[name]
    if 'emg3d_options_generation' not in _lib.SIGNATURES:
        _lib.options_fingerprint = lambda: tuple(_lib.lib().emg3d_get_option(_lib.lib().emg3d_option_name(i))
                                                 for i in range(_lib.lib().emg3d_option_count()))
for o in os.environ.get('AB_OPTS', '').split():
    k, v = o.split('=')
    assert _lib.lib().emg3d_set_option(k.encode(), int(v)) == 0, o
import torch
import emg3d_amd as emg3d
import bench
wl = bench.workload(wlname)
model = emg3d.Model(emg3d.TensorMesh(wl['h'], wl['origin']), **wl['res'])
b = bench.Bench(wl, model, torch.device('cuda', 0))
b.cycles((b.solver._GRAPH_AFTER + 1) * b.var.maxcycle)
b.cycles(2)
b.recording = True
torch.cuda.synchronize(); t0 = time.perf_counter()
b.cycles(steps)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
b.recording = False
st = b.kernel_stats()
print(json.dumps({'ms_per_cycle': dt / steps * 1e3, 'level0_ms_per_launch': {k: v['ms'] / v['launches'] for k, v in st.items()}}))
'''


def run(lib, wl, steps, opts=''):
    r = subprocess.run([sys.executable, '-c', CHILD, root, lib, wl, str(steps)], capture_output=True, text=True,
                       env=dict(os.environ, AB_OPTS=opts))
    line = [l for l in r.stdout.splitlines() if l.startswith('{')]
    if not line:
        raise SystemExit(r.stdout[-2000:] + r.stderr[-2000:])
    return json.loads(line[-1])


def main():
    other = sys.argv[1]
    wl = sys.argv[2] if len(sys.argv) > 2 else 'triaxial256'
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    for i in range(rounds):
        variants = [('other  ', other, ''), ('current', 'default', '')]
        variants += [('current ' + o, 'default', o) for o in os.environ.get('AB_VARIANTS', '').split(';') if o]
        for name, lib, opts in variants:
            res = run(lib, wl, 20, opts)
            print(name, wl, 'ms per cycle %.2f' % res['ms_per_cycle'],
                  'level-0 launches (x / y / z) ' + ' / '.join('%.3f' % res['level0_ms_per_launch'].get(str(k), float('nan')) for k in (1, 2, 3)), flush=True)


if __name__ == '__main__':
    main()

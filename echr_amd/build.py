"""In-tree build of libechr_hip.so: hipcc cross-compiles the gfx950 code objects without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libechr_hip.so')
SOURCES = ['gemm.hip', 'core.hip', 'decoder.hip', 'persist.hip', 'tsrm.hip', 'sst.hip', 'proposals.hip', 'step.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-munsafe-fp-atomics', '-Wall', '-Wno-unused-function']


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, 'obj')
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(os.path.dirname(HERE), 'include', 'echr_hip.h'))
    objs, procs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace('.hip', '.o'))
        objs.append(obj)
        if force or _newer(obj, [src] + headers):
            cmd = [hipcc] + FLAGS + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on %s' % s)
    if force or procs or _newer(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))

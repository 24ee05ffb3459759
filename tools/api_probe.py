"""Probe (development aid): does the placement of the output buffer relative to the input change the kernel time?  It does not -- what the
probe shows instead is the clock ramp: whichever placement is measured first in a fresh process runs ~10 % slower (1.12 ms against 0.975 ms
for every later batch of 20 launches), which is why bench.py and the micro-benchmarks load the device for 300 ms before they time anything.
    python tools/api_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import cosmoprimo_amd as cp
from cosmoprimo_amd import _lib
from oracle.workloads import pk_eh_default
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
k, pk = pk_eh_default(2048)
nb = 100000
f = cp.PowerToCorrelation(k, ell=0, device=dev)
plan = f._get_plan(dev); lib = _lib.load(); stream = torch.cuda.current_stream(dev)
pool = torch.empty(6 * nb * 2048 + (1 << 24), dtype=torch.float64, device=dev)    # one slab: explicit placement of every buffer
base = pool.data_ptr()
def view(offset_bytes):
    o = offset_bytes // 8
    return pool[o:o + nb * 2048].view(nb, 2048)
rows = view(0)
rows.copy_((torch.rand(nb, 1, device=dev, dtype=torch.float64) + 0.5) * torch.as_tensor(pk, device=dev)[None, :])
size = nb * 2048 * 8
def run(outs, n=20):
    for o in outs: lib.cp_fftlog_execute(plan.handle, rows.data_ptr(), o.data_ptr(), nb, 0, 0., 0, 0., 0, stream.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        o = outs[i % len(outs)]
        lib.cp_fftlog_execute(plan.handle, rows.data_ptr(), o.data_ptr(), nb, 0, 0., 0, 0., 0, stream.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print('rows at %#x' % base)
for rep in range(2):
    for name, offs in [('same out, adjacent', [size]), ('two outs alternating', [size, 2 * size]), ('three outs', [size, 2 * size, 3 * size]),
                       ('same out +4 KB', [size + 4096]), ('same out +64 KB', [size + 65536]), ('same out +1 MB', [size + (1 << 20)]),
                       ('same out +2 MB + 192 KB', [size + (1 << 21) + 196608]), ('same out far (4 sizes)', [4 * size]),
                       ('in place (out = in)', [0])]:
        if name.startswith('in place') and rep == 0:
            continue
        print('%-28s %.4f ms' % (name, run([view(o) for o in offs])))

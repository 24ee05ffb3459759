"""
Headline benchmark (BASELINE.json): batched FFTLog P(k) -> xi(r) transforms/sec at N=2048, fp64, with the achieved
fraction of the HBM roofline.

    python bench.py [--gpus N --steps K --warmup W]          # N=1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload: per GPU, 100 000 rows x 2048 log-k bins, rows = A_b (k/0.05)^dn_b P_EH(k),
A~U(0.5,2), dn~U(-0.1,0.1), default_rng(rank); PowerToCorrelation(k, ell=0), defaults (lowring, extrap=0) -> Np=4096.
One "step" = one pass of the fused kernel over the resident batch (inputs already in HBM).  Rows are independent, so
ranks shard with no data-path collective (weak scaling: per-GPU work fixed); `value` = rows of all ranks / max time.

Extra objects on the JSON line: "roofline" (algorithmic bytes 2*8*N per row / kernel time measured with HIP events on
the launch stream, against 8 TB/s) and, at N=1, "cpu_baseline" (the numpy oracle -- same math as the reference's
numpy path -- on a bounded sample, all host cores through a process pool).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_K = 2048
ROWS_PER_GPU = 100000
HBM_PEAK_GBS = 8000.          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_ROW = 2 * 8 * N_K   # read N f64 + write N f64 (tables are batch-shared, excluded)


def _cpu_chunk(args):
    """One worker: run oracle transforms of a (rows, N) chunk until `deadline` (numpy rfft/irfft path, reference fftlog.py:228-241)."""
    seed, rows, seconds = args
    os.environ['OMP_NUM_THREADS'] = '1'
    from oracle import fftlog as ofl
    from oracle.workloads import pk_eh_default, config2_rows
    k, pk = pk_eh_default(N_K)
    t = ofl.power_to_correlation(k, ell=0)
    x = config2_rows(k, pk, seed * rows, (seed + 1) * rows)[:, None, :]
    ofl.apply(t, x[:64])
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        ofl.apply(t, x)
        done += rows
    return done, time.perf_counter() - t0


def _host_cores():
    """Usable host cores: scheduler affinity capped by the cgroup CPU quota (containers often expose more CPUs than they may use)."""
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            cores = max(1, min(cores, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(seconds=10.):
    """Oracle ("port" of the reference's numpy path) on all usable host cores for about `seconds` of wall time."""
    import multiprocessing as mp
    cores = _host_cores()
    rows = 256   # 256 x 4096 f64 temporaries stay in cache (faster per core than larger chunks)
    ctx = mp.get_context('fork')   # before any GPU initialisation in this process
    with ctx.Pool(cores) as pool:
        tic = time.perf_counter()
        res = pool.map(_cpu_chunk, [(i % 300, rows, seconds) for i in range(cores)])
        wall = time.perf_counter() - tic
    total = sum(r[0] for r in res)
    return {'value': total / wall, 'unit': 'transforms/s', 'cores': cores, 'kind': 'port',
            'sample': '%d-row chunks of the config-2 batch (N=%d) repeated for %.0f s on each of %d processes, numpy oracle (%d transforms)'
                      % (rows, N_K, seconds, cores, total),
            'per_core_value': float(np.mean([r[0] / r[1] for r in res]))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--rows', type=int, default=ROWS_PER_GPU, help='rows per GPU (default: the config-2 batch)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--gather', action='store_true', help='also time the final RCCL all_gather of the result shards (reported separately)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    distributed = world > 1

    def log(msg):
        if rank == 0:
            print('[bench %.1fs] %s' % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)

    T0 = time.perf_counter()
    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        log('cpu baseline on %d cores ...' % _host_cores())
        cpu = cpu_baseline()   # before the GPU is touched (fork-safe)
        log('cpu baseline: %.0f transforms/s' % cpu['value'])

    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), 'bench.py needs a ROCm GPU; there is no CPU path to measure'
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    if distributed:
        dist.init_process_group('nccl', device_id=dev)

    import cosmoprimo_amd as cp
    from oracle.workloads import pk_eh_default
    k, pk = pk_eh_default(N_K)
    rng = np.random.default_rng(rank)
    nb = args.rows
    amp, dn = rng.uniform(0.5, 2., nb), rng.uniform(-0.1, 0.1, nb)
    tk, tpk = torch.as_tensor(k, device=dev), torch.as_tensor(pk, device=dev)
    rows = (torch.as_tensor(amp, device=dev)[:, None] * (tk[None, :] / 0.05) ** torch.as_tensor(dn, device=dev)[:, None] * tpk[None, :]).contiguous()
    f = cp.PowerToCorrelation(k, ell=0, device=dev)
    log('inputs resident: %d x %d f64' % (nb, N_K))

    # product call path = FFTlog.__call__ -> cp_fftlog_execute; for timing, call the C ABI directly on preallocated buffers
    from cosmoprimo_amd import _lib
    lib = _lib.load()
    plan = f._get_plan(dev)
    out = torch.empty_like(rows)
    stream = torch.cuda.current_stream(dev)

    def step():
        _lib.check(lib.cp_fftlog_execute(plan.handle, rows.data_ptr(), out.data_ptr(), nb, 0, 0., 0, 0., 0, stream.cuda_stream))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    log('warmup done')
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    tic = time.perf_counter()
    for i in range(args.steps):
        starts[i].record(stream)   # HIP events on the stream the kernel is launched on
        step()
        ends[i].record(stream)
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - tic
    kernel_ms = float(np.mean([s.elapsed_time(e) for s, e in zip(starts, ends)]))

    # the same K steps through the product API (cp.PowerToCorrelation.__call__ on the resident tensor: output allocation, plan
    # lookup and the ctypes call included, no host synchronisation inside)
    for _ in range(max(1, args.warmup)):
        f(rows)
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    tic_api = time.perf_counter()
    for i in range(args.steps):
        s_api, xi_api = f(rows)
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    elapsed_api = time.perf_counter() - tic_api
    assert xi_api.shape == rows.shape and bool(torch.equal(xi_api, out)), 'API path and C-ABI path disagree'

    gather_ms = None
    if distributed:
        tapi = torch.tensor([elapsed_api], device=dev, dtype=torch.float64)
        dist.all_reduce(tapi, op=dist.ReduceOp.MAX)
        elapsed_api = float(tapi[0])
        tmax = torch.tensor([elapsed, kernel_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(tmax[0]), float(tmax[1])
        if args.gather:
            full = torch.empty((world * nb, N_K), dtype=out.dtype, device=dev)
            dist.all_gather_into_tensor(full, out)
            torch.cuda.synchronize(dev)
            dist.barrier()
            t0 = time.perf_counter()
            dist.all_gather_into_tensor(full, out)
            torch.cuda.synchronize(dev)
            dist.barrier()
            gather_ms = (time.perf_counter() - t0) * 1e3

    # sanity: the timed output is the real transform (spot-check one row against the oracle on rank 0)
    if rank == 0:
        from oracle import fftlog as ofl
        t = ofl.power_to_correlation(k, ell=0)
        ref = ofl.apply(t, rows[7].cpu().numpy())[0]
        w = t.y[0] ** 1.5
        err = float(np.abs((out[7].cpu().numpy() - ref) * w).max() / np.abs(ref * w).max())
        assert err < 1e-13, 'bench output failed parity spot-check: %g' % err

        total_rows = nb * world * args.steps
        value = total_rows / elapsed
        achieved = BYTES_PER_ROW * nb / (kernel_ms * 1e-3) / 1e9   # GB/s per GPU, dominant (only) kernel
        grid = __import__('ctypes').c_int()
        block = __import__('ctypes').c_int()
        lds = __import__('ctypes').c_int()
        lib.cp_fftlog_plan_info(plan.handle, nb, grid, block, lds)
        traffic, traffic_src = None, None
        import glob
        prof = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_summary.json')))
        if prof and nb == ROWS_PER_GPU:   # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command
            with open(prof[-1]) as fh:
                traffic = json.load(fh)['hbm_bytes_per_launch']['total']
            traffic_src = os.path.relpath(prof[-1], ROOT)
        line = {
            'metric': 'batched FFTLog P(k)->xi(r) transforms/sec (N=2048)', 'value': value, 'unit': 'transforms/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'config 2: %d rows x N_k=%d per GPU, PowerToCorrelation(ell=0), Np=4096, extrap=0' % (nb, N_K),
                       'rows_per_gpu': nb, 'n_k': N_K, 'padded_size': 4096, 'parallelism': 'rows sharded over %d GPU(s), no collective' % world,
                       'grid': grid.value, 'block': block.value, 'lds_bytes': lds.value},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'traffic_source': traffic_src, 'kernel': 'fftlog_kernel<4096,16,IN_HALF_ZERO,OUT_HALF>', 'kernel_ms': kernel_ms,
                         'algorithmic_bytes_per_launch': BYTES_PER_ROW * nb},
            'parity_spot_check_tilted_err': err,
            'value_api': nb * world * args.steps / elapsed_api,   # same batch through cp.PowerToCorrelation.__call__ (resident tensors)
        }
        if cpu is not None:
            line['cpu_baseline'] = cpu
        if gather_ms is not None:
            line['gather_ms'] = gather_ms
            line['value_with_gather'] = nb * world / (elapsed / args.steps + gather_ms * 1e-3)
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""GPU: the batch pipelines run without any host synchronisation or host-to-device copy once they have run once -- the property that lets
the host queue launches ahead of the device.  The check is mechanical: such a function can be recorded into a HIP graph (``torch.cuda.graph``
refuses synchronisations, pageable copies and allocations outside its pool), and replaying the graph on new parameter values in the same
buffers gives exactly what the eager call gives."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def capture_and_compare(torch, fn, inputs, fresh):
    """fn() reads the tensors ``inputs`` (dict) and returns a device tensor.  Run it eagerly, capture it, replay it on ``fresh`` values."""
    dev = next(iter(inputs.values())).device
    for _ in range(2):
        fn()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = fn()
    for name, value in fresh.items():
        inputs[name].copy_(value)
    graph.replay()
    torch.cuda.synchronize(dev)
    replayed = out.clone()
    eager = fn()
    assert bool(torch.isfinite(eager).all())
    assert torch.equal(replayed, eager)


def eh_parameters(n, seed, torch, dev):
    rng = np.random.default_rng(seed)
    par = dict(Omega_m=rng.uniform(.25, .40, n), Omega_b=rng.uniform(.04, .06, n), h=rng.uniform(.6, .8, n), n_s=rng.uniform(.92, 1., n))
    return {name: torch.as_tensor(v, device=dev) for name, v in par.items()}


@pytest.mark.parametrize('engine', ['wallish2018', 'brieden2022'])
def test_bao_filter_chunk(engine):
    import torch
    import cosmoprimo_amd as cp
    from cosmoprimo_amd.bao_filter import PowerSpectrumBAOFilter
    dev = torch.device('cuda', 0)
    static, fresh = eh_parameters(2048, 1, torch, dev), eh_parameters(2048, 2, torch, dev)
    state = {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fid = cp.Cosmology(engine='eisenstein_hu')

        def chunk():
            # (fresh views of the parameter arrays on every call, as a loop over chunks of a larger batch makes them: a host copy cached by tensor
            # identity would hide a device-to-host transfer)
            cosmo = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **{name: v[:] for name, v in static.items()})
            interp = cosmo.get_fourier().pk_interpolator(z=np.array([0.]))
            kw = dict(cosmo_fid=fid, cosmo=cosmo) if engine == 'brieden2022' else {}
            if 'filter' not in state:
                state['filter'] = PowerSpectrumBAOFilter(interp, engine=engine, **kw)
            else:
                state['filter'](interp, cosmo=cosmo if kw else None)
            return state['filter']._pknow_rows

        capture_and_compare(torch, chunk, static, fresh)


def test_sigma_rz_of_a_batch_of_cosmologies():
    import torch
    import cosmoprimo_amd as cp
    dev = torch.device('cuda', 0)
    static, fresh = eh_parameters(1000, 3, torch, dev), eh_parameters(1000, 4, torch, dev)
    r, z = torch.as_tensor(np.geomspace(1., 100., 64), device=dev), torch.as_tensor(np.linspace(0., 2., 16), device=dev)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        capture_and_compare(torch, lambda: cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **{name: v[:] for name, v in static.items()}).get_fourier().pk_interpolator().sigma_rz(r, z),
                            static, fresh)


def test_fftlog_and_distances():
    import torch
    import cosmoprimo_amd as cp
    from cosmoprimo_amd import background
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(5)
    k = np.geomspace(1e-4, 1e2, 512)
    f = cp.PowerToCorrelation(k, ell=[0, 2], device=dev)
    rows = {'pk': torch.as_tensor(rng.uniform(0.5, 2., (300, 2, 512)) * k**-1.2, device=dev)}
    capture_and_compare(torch, lambda: f(rows['pk'])[1], rows, {'pk': torch.as_tensor(rng.uniform(0.5, 2., (300, 2, 512)) * k**-0.8, device=dev)})
    samples = {name: torch.as_tensor(v, device=dev) for name, v in dict(om=rng.uniform(0.1, 0.5, 5000), w0=rng.uniform(-1.5, -0.5, 5000),
                                                                         zz=rng.uniform(0., 3., 5000)).items()}
    fresh = {name: torch.as_tensor(v, device=dev) for name, v in dict(om=rng.uniform(0.1, 0.5, 5000), w0=rng.uniform(-1.5, -0.5, 5000),
                                                                       zz=rng.uniform(0., 3., 5000)).items()}
    capture_and_compare(torch, lambda: background.distance('comoving_radial_distance', samples['zz'][:, None], dict(w0_fld=samples['w0']),
                                                           Omega_m=samples['om'], per_cosmology_z=True), samples, fresh)


def test_sigma_rz_of_a_batch_of_tables(golden):
    import torch
    import cosmoprimo_amd as cp
    dev = torch.device('cuda', 0)
    g = golden('sigma')
    k, z = g['table_k'], g['table_z']
    rng = np.random.default_rng(6)
    tables = {'pk': torch.as_tensor(rng.uniform(0.5, 2., (50, 1, 1)) * g['table_pk'][None], device=dev)}
    fresh = {'pk': torch.as_tensor(rng.uniform(0.5, 2., (50, 1, 1)) * g['table_pk'][None], device=dev)}
    r, zq = torch.as_tensor(np.geomspace(1., 100., 32), device=dev), torch.as_tensor(np.linspace(0.1, 2.5, 8), device=dev)
    capture_and_compare(torch, lambda: cp.PowerSpectrumInterpolator2D(k, z, tables['pk']).sigma_rz(r, zq), tables, fresh)


def test_upload_cache():
    """Small host arrays are kept on the device by content (dtype and shape included); larger ones are copied every time."""
    import torch
    from cosmoprimo_amd import _device as dv
    dev = torch.device('cuda', 0)
    a = np.linspace(0., 1., 100)
    t1, t2 = dv.upload(a, dev), dv.upload(a.copy(), dev)
    assert t1.data_ptr() == t2.data_ptr() and t1.dtype == torch.float64 and tuple(t1.shape) == (100,)
    assert dv.upload(a.reshape(10, 10), dev).data_ptr() != t1.data_ptr() and tuple(dv.upload(a.reshape(10, 10), dev).shape) == (10, 10)
    assert dv.upload(a.astype('f4'), dev).dtype == torch.float32
    mask = a > 0.5
    tm = dv.upload(mask, dev)
    assert tm.dtype == torch.bool and bool((tm.cpu().numpy() == mask).all())
    idx = np.flatnonzero(mask)
    assert dv.upload(idx, dev).dtype == torch.int64 and np.array_equal(dv.upload(idx, dev).cpu().numpy(), idx)
    assert tuple(dv.upload(3.5, dev).shape) == () and float(dv.upload(3.5, dev)) == 3.5
    b = a.copy()
    b[3] = 7.
    assert dv.upload(b, dev).data_ptr() != t1.data_ptr() and float(dv.upload(b, dev)[3]) == 7. and float(t1[3]) == a[3]
    big = np.arange(100000.)
    u1, u2 = dv.upload(big, dev), dv.upload(big, dev)
    assert u1.data_ptr() != u2.data_ptr() and np.array_equal(u1.cpu().numpy(), big)
    assert dv.to_device(np.arange(5), dev).dtype == torch.float64
    t = torch.arange(4., device=dev)
    assert dv.upload(t, dev) is t

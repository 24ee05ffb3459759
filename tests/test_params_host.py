"""Host-side parameter compilation (no GPU): massive-neutrino inputs against values produced by the reference in this container
(cosmology.py:960-1140): masses from Omega_ncdm / omega_ncdm (Newton), neutrino_hierarchy splitting, N_ur from N_eff."""
import numpy as np
import pytest


def test_ncdm_inputs():
    from cosmoprimo_amd.cosmology import _compile_params, CosmologyInputError
    np.testing.assert_allclose(_compile_params(dict(Omega_ncdm=[0.0014, 0.003], h=0.68))['m_ncdm'], [0.060294267293085214, 0.1292070888252256], rtol=1e-13)
    np.testing.assert_allclose(_compile_params(dict(Omega_ncdm=0.0014))['m_ncdm'], [0.0638935], rtol=1e-6)
    np.testing.assert_allclose(_compile_params(dict(omega_ncdm=0.00064))['m_ncdm'], [0.0596087], rtol=1e-6)
    assert _compile_params(dict(Omega_ncdm=0.))['m_ncdm'] == [0.]
    p = _compile_params(dict(m_ncdm=0.12, neutrino_hierarchy='normal'))
    np.testing.assert_allclose(p['m_ncdm'], [0.030108750535617665, 0.031311928379070764, 0.05857932108531181], rtol=1e-14)
    assert len(p['T_ncdm_over_cmb']) == 3
    np.testing.assert_allclose(p['N_ur'], 3.044 - 3 * 0.71611**4 * (4. / 11.)**(-4. / 3.), rtol=0, atol=1e-15)   # sum of three equal terms (cancellation)
    assert _compile_params(dict(m_ncdm=[0.06], N_ur=2.03))['N_ur'] == 2.03
    assert _compile_params({})['m_ncdm'] == [] and _compile_params({})['N_ur'] == 3.044
    with pytest.raises(CosmologyInputError):
        _compile_params(dict(m_ncdm=0.06, Omega_ncdm=0.001))
    with pytest.raises(TypeError):
        _compile_params(dict(m_ncdm=[0.06, 0.1], T_ncdm_over_cmb=[0.7]))
    with pytest.raises(CosmologyInputError):
        _compile_params(dict(w0_fld=-0.5, wa_fld=1.))


def test_param_tables_and_conflicts(golden):
    """Input names, defaults and exclusions (reference tests/test_cosmology.py::test_params, test_error) against the reference's own tables."""
    import cosmoprimo_amd as cp
    from cosmoprimo_amd import Cosmology, CosmologyError, CosmologyInputError
    g = golden('cosmology_api')
    names = set(Cosmology.get_default_params())
    # 'omk' is an alias the reference documents but loses (a duplicate key in its alias table); accepted here
    assert names - set(g['default_names'].tolist()) == {'omk'} and set(g['default_names'].tolist()) <= names
    assert set(Cosmology.get_default_params(include_conflicts=False)) == set(g['default_names_noconflicts'].tolist())
    assert set(Cosmology.get_default_params(of='cosmology')) - {'omk'} == set(g['default_cosmology_names'].tolist())
    with pytest.raises(CosmologyInputError):
        Cosmology.get_default_params(of='nothing')
    cosmo = Cosmology()
    assert cosmo.get_default_params() == Cosmology.get_default_params()
    for bad in [dict(sigma8=1., A_s=1e-9), dict(tau=0.05, tau_reio=0.06), dict(Omega_m=-0.1), dict(h=-0.7), dict(YHe='nope'), dict(w0_fld=0.5),
                dict(H0=70., h=0.7), dict(omega_b=0.02, ombh2=0.02), dict(Omega_m=0.3, omch2=0.1)]:
        with pytest.raises(CosmologyInputError):
            Cosmology(**bad)
    assert issubclass(CosmologyInputError, CosmologyError)
    cosmo = Cosmology(Omega_cdm=0.3, Omega_b=0.02, h=0.8, n_s=0.96)
    assert cosmo['omega_cdm'] == 0.3 * 0.8**2 and cosmo['sigma8'] == 0.8
    assert len(cosmo['z_pk']) == 30 and np.array_equal(cosmo['z_pk'], g['z_pk'])
    assert cosmo.batch_size is None      # the array of output redshifts is not a batch of cosmologies
    assert np.allclose(Cosmology(z_pk=[1., 0.5])['z_pk'], [0., 0.5, 1.])
    cosmo = Cosmology(ombh2=0.05, omch2=0.1)
    assert np.allclose(cosmo['omega_b'], 0.05) and np.allclose(cosmo['omega_cdm'], 0.1)
    cosmo = Cosmology(Omega_g=5e-5, omega_ur=1.7e-5)
    np.testing.assert_allclose([cosmo['T_cmb'], cosmo['N_ur'], cosmo['N_eff']], g['from_Omega_g_ur'], rtol=1e-13)
    cosmo = Cosmology(r=0.1)
    np.testing.assert_allclose([cosmo['n_t'], cosmo['alpha_t']], g['tensor_defaults'], rtol=1e-13)
    params = cosmo.get_params(of='cosmology')
    assert params['r'] == 0.1 and 'kmax_pk' not in params and cosmo.get_params(of='calculation')['kmax_pk'] == 10.
    assert cosmo.get_params(of='extra') == {} and set(cosmo.get_params()) == set(cosmo.get_params(of='all'))
    assert cp.cosmology.find_conflicts('tau', conflicts=Cosmology._conflict_parameters) == cp.cosmology.find_conflicts('z_reio', conflicts=Cosmology._conflict_parameters) and cp.cosmology.find_conflicts('n_s', conflicts=Cosmology._conflict_parameters) == ('n_s', 'ns') and cp.cosmology.find_conflicts('n_s') == ()


def test_clone_equality_persistence(tmp_path, golden):
    """clone in both bases, equality, state round trips to json / npy (reference test_params, test_clone) -- no engine, host only."""
    from cosmoprimo_amd import Cosmology, CosmologyInputError
    g = golden('cosmology_api')
    cosmo = Cosmology(omega_cdm=0.2)
    for base, ref in [('internal', g['clone_internal']), ('input', g['clone_input'])]:
        clone = cosmo.clone(base=base, h=cosmo['h'] * 1.1)
        np.testing.assert_allclose([clone['Omega_m'], clone['Omega_cdm'], clone['omega_cdm']], ref, rtol=1e-13)
    assert cosmo.clone(base=None, h=0.6)['Omega_cdm'] == cosmo['Omega_cdm']
    with pytest.raises(CosmologyInputError):
        cosmo.clone(base='other')
    with pytest.raises(CosmologyInputError):
        cosmo.clone(h=0.6, H0=60.)
    assert cosmo.clone(sigma8=0.9)['sigma8'] == 0.9 and 'A_s' not in cosmo.clone(sigma8=0.9).get_params()
    assert 'sigma8' not in cosmo.clone(A_s=2e-9).get_params()                # a new name removes what it excludes
    assert cosmo == cosmo.clone() and cosmo != cosmo.clone(h=0.6) and cosmo != 1
    m_ncdm = [0.01, 0.02, 0.05]
    cosmo = Cosmology(m_ncdm=m_ncdm, Omega_m=np.array([0.3, 0.31]))
    for name in ['cosmo.json', 'sub/cosmo.npy']:
        fn = str(tmp_path / name)
        cosmo.write(fn)
        back = Cosmology.read(fn)
        assert back == cosmo and np.allclose(back['m_ncdm'], m_ncdm) and back.engine is None and back.batch_size == 2
    with pytest.warns(DeprecationWarning):
        cosmo.save(str(tmp_path / 'old.npy'))
    with pytest.warns(DeprecationWarning):
        assert Cosmology.load(str(tmp_path / 'old.npy')) == cosmo
    with pytest.raises(AttributeError):
        cosmo.comoving_radial_distance
    assert 'tau_reio' not in dir(cosmo) and 'comoving_radial_distance' not in dir(cosmo)


def test_reference_helper_names(golden):
    """The small names of the reference a caller may import (cosmology.py:188-200, 848-852, 1940-1952; utils.py:51-138), by name: the knots against the
    golden grids of the reference, compute_ncdm_momenta against the reference's tabulated density, the decorators by behaviour."""
    import warnings
    import numpy as np
    from cosmoprimo_amd import Cosmology
    from cosmoprimo_amd.cosmology import get_default_z_interp, compute_ncdm_momenta
    from cosmoprimo_amd import utils
    g = golden('ncdm')
    np.testing.assert_allclose(get_default_z_interp('rho_ncdm'), g['ncdm_knots'], rtol=1e-15)
    np.testing.assert_allclose(get_default_z_interp('p_ncdm'), g['ncdm_knots'], rtol=1e-15)
    zc = get_default_z_interp('comoving_radial_distance')
    assert zc.shape == (119,) and zc[0] == 0. and abs(zc[-1] - 9999.) < 1e-8 and get_default_z_interp('time').shape == (400,)
    with pytest.raises(ValueError):
        get_default_z_interp('nope')
    # the tabulated density of the last golden cosmology: rho(z) = momenta / (1 + z)^3 / h^2 on the knots (cosmology.py:441-442)
    from oracle.gen_golden import NCDM_PARAMS
    par = NCDM_PARAMS[-1]
    T = 2.7255 * np.atleast_1d(par['T_ncdm_over_cmb'])[0]
    zk = g['ncdm_knots']
    rho = compute_ncdm_momenta(T, np.atleast_1d(par['m_ncdm'])[0], zk, out='rho') / (1 + zk)**3 / par.get('h', 0.7)**2
    np.testing.assert_allclose(rho, g['rho_ncdm_table'][:, 0], rtol=1e-13)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        assert Cosmology.get_default_parameters() == Cosmology.get_default_params()
    assert any('deprecated' in str(w.message) for w in caught)

    @utils.addproperty('a', 'b')
    class Holder(utils.BaseClass):
        def __init__(self):
            self._a, self._b = 1, [2]

        @utils.flatarray(iargs=[0, 1], dtype=None)
        def add(self, x, y, scale=1.):
            assert x.ndim == 1 and y.ndim == 1
            return {'sum': scale * (x + y), 'pair': np.array([x, y])}

    h = Holder()
    c = h.copy()
    assert (h.a, h.b) == (1, [2]) and c is not h and c.b is h.b
    out = h.add(np.ones((2, 3), dtype='f4'), np.full((2, 3), 2., dtype='f4'), scale=2.)
    assert out['sum'].shape == (2, 3) and out['sum'].dtype == np.float32 and out['pair'].shape == (2, 2, 3) and np.all(out['sum'] == 6.)
    with pytest.raises(ValueError):
        h.add(np.ones(3), np.ones(4))

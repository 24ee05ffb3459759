"""ctypes binding of libcosmoprimo_amd.so (C ABI declared in include/cosmoprimo_amd.h).

The library is the product: there is no Python/CPU fallback.  If it is missing (or its ABI does
not match) importing any compute entry point raises, loudly.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libcosmoprimo_amd.so')
ABI_VERSION = 4

CP_OK, CP_EINVAL, CP_EUNSUPPORTED, CP_EDEVICE, CP_ENOMEM = range(5)
EXTRAP_CONSTANT, EXTRAP_EDGE, EXTRAP_LOGLOG = range(3)
(KERNEL_BESSEL_J, KERNEL_SPHERICAL_BESSEL_J, KERNEL_TOPHAT, KERNEL_TOPHAT_SQ, KERNEL_GAUSSIAN, KERNEL_GAUSSIAN_SQ) = range(6)
KERNEL_CUSTOM = 100


class FFTlogSpec(ctypes.Structure):
    """cp_fftlog_spec (include/cosmoprimo_amd.h): kernel, tilt and convention of one transform of a plan."""
    _fields_ = [('kind', ctypes.c_int), ('param', ctypes.c_double), ('q', ctypes.c_double), ('xy', ctypes.c_double), ('pre_power', ctypes.c_double),
                ('pre_const', ctypes.c_double), ('post_sign', ctypes.c_double)]


_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int_p = ctypes.POINTER(ctypes.c_int)

# name -> (restype, argtypes); every symbol of include/cosmoprimo_amd.h
SIGNATURES = {
    'cp_abi_version': (ctypes.c_int, []),
    'cp_last_error': (ctypes.c_char_p, []),
    'cp_device_count': (ctypes.c_int, []),
    'cp_loggamma': (ctypes.c_int, [_c_double_p, _c_double_p, ctypes.c_longlong]),
    'cp_gamma': (ctypes.c_int, [_c_double_p, _c_double_p, ctypes.c_longlong]),
    'cp_kernel_eval': (ctypes.c_int, [ctypes.c_int, ctypes.c_double, _c_double_p, _c_double_p, ctypes.c_longlong]),
    'cp_fftlog_padded_size': (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    'cp_fftlog_tables': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _c_double_p, ctypes.POINTER(FFTlogSpec), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       _c_double_p, _c_double_p] + [_c_double_p] * 8),
    'cp_fftlog_plan_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_double_p,
                                            _c_double_p, _c_double_p, ctypes.c_int]),
    'cp_fftlog_execute': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_double,
                                        ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p]),
    'cp_fftlog_execute_window': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_double,
                                               ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cp_fftlog_plan_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'cp_fftlog_plan_info': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_longlong, _c_int_p, _c_int_p, _c_int_p]),
    'cp_background_distance': (ctypes.c_int, [ctypes.c_longlong, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cp_derived_parameters': (ctypes.c_int, [ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    'cp_background_knots': (ctypes.c_int, [_c_double_p, ctypes.c_int]),
    'cp_background_init': (ctypes.c_int, [ctypes.c_int]),
    'cp_distance_from_radial': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    'cp_ncdm_knots': (ctypes.c_int, [_c_double_p, ctypes.c_int]),
    'cp_growth_ode_knots': (ctypes.c_int, [_c_double_p, ctypes.c_int]),
    'cp_growth_ode_tables': (ctypes.c_int, [ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                           ctypes.c_void_p]),
    'cp_ncdm_tables': None,        # filled below (takes cp_param by value)
    'cp_background_eval': (ctypes.c_int, [ctypes.c_longlong, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cp_power_workspace_bytes': (ctypes.c_longlong, [ctypes.c_longlong]),
    'cp_power_eval': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    'cp_power_eval_variants': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    'cp_rfft_plan_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int]),
    'cp_rfft_plan_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'cp_rfft_forward': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]),
    'cp_rfft_backward': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]),
    'cp_eh_scalars': (ctypes.c_int, [ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_void_p]),
    'cp_variants_scalars': (ctypes.c_int, [ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    'cp_sigma_rz_workspace_bytes': (ctypes.c_longlong, [ctypes.c_longlong, ctypes.c_int]),
    'cp_sigma_rz_fused_available': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    'cp_geospline_plan_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), _c_double_p, ctypes.c_int, _c_double_p, ctypes.c_int, ctypes.c_int]),
    'cp_geospline_plan_create_prefiltered': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, _c_double_p, _c_double_p, _c_double_p,
                                                           _c_double_p, _c_double_p, ctypes.c_int, ctypes.c_int]),
    'cp_geospline_basis': (ctypes.c_int, [ctypes.c_double, _c_double_p]),
    'cp_geospline_plan_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'cp_geospline_plan_info': (ctypes.c_int, [ctypes.c_void_p, _c_int_p, _c_int_p, _c_int_p]),
    'cp_fftlog_geospline_execute': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_void_p]),
    'cp_fftlog_spline_execute': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]),
    'cp_sigma_rz_analytic': (ctypes.c_int, [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cp_sigma_rz_analytic_prefiltered': (ctypes.c_int, [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    'cp_sigma_rz_functional': (ctypes.c_int, [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    'cp_spline_plan_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _c_double_p, ctypes.c_int, _c_double_p, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'cp_linop_plan_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, _c_double_p, ctypes.c_int]),
    'cp_spline_apply': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]),
    'cp_spline_apply_grouped': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]),
    'cp_spline_apply_outer': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int,
                                            ctypes.c_double, ctypes.c_void_p]),
    'cp_linop_apply_mid': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_int, ctypes.c_double,
                                         ctypes.c_void_p]),
    'cp_tables_rows_available': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    'cp_tables_rows': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]),
    'cp_spline_plan_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'cp_spline_plan_info': (ctypes.c_int, [ctypes.c_void_p, _c_int_p, _c_int_p, _c_int_p]),
    'cp_spline_plan_columns': (ctypes.c_int, [ctypes.c_void_p, _c_int_p, _c_int_p]),
    'cp_spline_columns_scratch_doubles': (ctypes.c_longlong, [ctypes.c_longlong, ctypes.c_int]),
    'cp_spline_rows_at_queries': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cp_spline_columns': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    'cp_wallish_box': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_int, ctypes.c_void_p]),
    'cp_wallish_dd_box': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    'cp_splice_plan_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _c_double_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                            ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_int, _c_double_p, ctypes.c_int]),
    'cp_splice_apply': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p]),
    'cp_splice_plan_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'cp_splice_plan_scheme': (ctypes.c_int, [ctypes.c_void_p]),
    'cp_math_eval': (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]),
    'cp_splice_plan_set_scheme': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'cp_spline_rows_plan_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_double_p,
                                                 ctypes.c_int]),
    'cp_spline_rows_plan_info': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                               ctypes.POINTER(ctypes.c_int)]),
    'cp_spline_rows_apply': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_void_p]),
    'cp_spline_rows_second_derivatives': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]),
    'cp_spline_rows_pairs': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]),
    'cp_spline_rows_plan_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'cp_tables_rows_direct': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int,
                                            ctypes.c_double, ctypes.c_void_p]),
    'cp_gap_spline': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cp_wallish_finish': (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cp_brieden_ratio': (ctypes.c_int, [ctypes.c_void_p] * 7 + [ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cp_brieden_knots': (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_void_p]),
    'cp_brieden_finish': (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cp_brieden_resample': (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cp_brieden_smooth': (ctypes.c_int, [ctypes.c_void_p] * 7 + [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cp_dst_plan_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _c_double_p, ctypes.c_int]),
    'cp_dst_execute': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cp_bilinear_pairs': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_void_p]),
    'cp_interp_linear': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int,
                                        ctypes.c_void_p]),
    'cp_interp_table_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_longlong, _c_double_p, _c_double_p, ctypes.c_int]),
    'cp_interp_table_law': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_longlong)]),
    'cp_interp_table_apply': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]),
    'cp_interp_table_apply_f32': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]),
    'cp_interp_table_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'cp_spline_points': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cp_rows_screen': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                      ctypes.c_void_p]),
    'cp_dst_plan_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'cp_dst_forward_analytic_workspace_bytes': (ctypes.c_longlong, [ctypes.c_longlong]),
    'cp_wallish_full': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'cp_wallish_tail': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'cp_dst_forward_analytic_box': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_void_p]),
    'cp_dst_forward_analytic': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    'cp_spline_operator': (ctypes.c_int, [ctypes.c_int, _c_double_p, ctypes.c_int, _c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_double_p,
                                         _c_int_p]),
}


class cp_param(ctypes.Structure):
    """A per-cosmology parameter: device array (ptr) or broadcast value (include/cosmoprimo_amd.h)."""
    _fields_ = [('ptr', ctypes.c_void_p), ('value', ctypes.c_double)]


class cp_ncdm(ctypes.Structure):
    """Massive-neutrino spline tables of the background kernels (include/cosmoprimo_amd.h)."""
    _fields_ = [('nspecies', ctypes.c_int), ('species', ctypes.c_int), ('tab', ctypes.c_void_p)]


SIGNATURES['cp_ncdm_tables'] = (ctypes.c_int, [ctypes.c_longlong, ctypes.c_int, cp_param, cp_param, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                               _c_double_p, _c_double_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p])
SIGNATURES['cp_sigma8_normalise'] = (ctypes.c_int, [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_void_p, cp_param, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_void_p])
NCDM_NKNOTS = 119
GROWTH_NKNOTS = 201

BG_PARAMS = ('h', 'Omega_cdm', 'Omega_b', 'Omega_k', 'T_cmb', 'N_ur', 'w0_fld', 'wa_fld')
SPLINE_BC = {'natural': 0, 'clamped': 1, 'not-a-knot': 2}
PK_PARAMS = ('A_s', 'n_s', 'alpha_s', 'beta_s', 'k_pivot')
ENGINES = {'eisenstein_hu': 0, 'eisenstein_hu_nowiggle': 1, 'bbks': 2}
PK_WHAT = {'matter': 0, 'transfer': 1, 'primordial': 2, 'log_k_matter': 3}
DERIVED_VALUES = ('_h2', 'H0', 'Omega_g', 'T_ur', 'Omega_ur', 'Omega_r', 'Omega_m', 'Omega_de', 'K', 'omega_b', 'omega_cdm', 'omega_m', 'omega_g', 'omega_ur',
                  'omega_r', 'omega_k', 'omega_de')      # enum cp_derived_value
VARIANTS_SCALARS = ('omega_b', 'omega_m', 'frac_b', 'frac_cdm', 'frac_cb', 'frac_ncdm', 'theta_cmb', 'z_eq', 'k_eq', 'z_drag', 'rs_drag', 'p_c', 'p_cb',
                    'gamma_ncdm', 'beta_c')      # enum cp_variants_scalar
MATH_FUNCTIONS = {'exp_mid': 0, 'exp_tab': 1, 'log_pos': 2, 'log_tab': 3, 'exp10_mid': 4, 'exp10_tab': 5, 'sin_bounded': 6, 'recip': 7, 'rsqrt_pos': 8}      # enum cp_math_function
EH_SCALARS = ('rs_drag', 'z_drag', 'z_eq', 'k_eq', 'r_drag', 'r_eq', 'k_silk', 'alpha_c', 'beta_c', 'alpha_b', 'beta_node', 'beta_b', 'alpha_gamma',
              'gamma')
BG_KINDS = {'comoving_radial_distance': 0, 'comoving_transverse_distance': 1, 'angular_diameter_distance': 2, 'luminosity_distance': 3,
            'efunc': 4, 'hubble_function': 5, 'growth_cpt': 6, 'growth_rate': 7, 'rho_crit': 8, 'Omega_m': 9, 'Omega_de': 10,
            'rho_g': 11, 'rho_b': 12, 'rho_ur': 13, 'rho_cdm': 14, 'rho_k': 15, 'rho_Lambda': 16, 'rho_fld': 17, 'rho_de': 18, 'rho_tot': 19,
            'rho_m': 20, 'rho_r': 21, 'T_cmb': 22, 'time': 23, 'age': 24, 'rho_ncdm': 25, 'p_ncdm': 26, 'rs': 27, 'rs_cosmomc': 28}
BG_AS_FRACTION = 32
for _name in ('g', 'b', 'ur', 'cdm', 'k', 'Lambda', 'fld', 'r', 'ncdm'):     # Omega_x(z) = rho_x(z) / rho_crit(z)
    BG_KINDS['Omega_' + _name] = BG_KINDS['rho_' + _name] | BG_AS_FRACTION
BG_KINDS['pfrac_ncdm'] = BG_KINDS['p_ncdm'] | BG_AS_FRACTION     # p_ncdm(z) / rho_crit(z)

_lib = None


class LibraryError(RuntimeError):
    """The HIP library is missing or unusable."""


def load():
    """Load (once) and return the shared library; raise :class:`LibraryError` if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH
    variant = os.environ.get('COSMOPRIMO_AMD_LIBRARY')      # measurements only (tools/variant_lib.sh: a variant of ONE source linked beside the shipped library,
    if variant:                                             # which is never rebuilt in place): an explicit path, which must exist
        if not os.path.isfile(variant):
            raise LibraryError('COSMOPRIMO_AMD_LIBRARY = {} does not exist'.format(variant))
        path = variant
    if not os.path.isfile(path):
        raise LibraryError('{} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` or '
                           '`make -C cosmoprimo_amd/csrc -j8` (hipcc, gfx950). There is no CPU fallback.'.format(LIB_PATH))
    try:
        # torch ships its own libamdhip64.so.7; import it first so the HIP runtime is shared with torch's allocator/streams
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise LibraryError('{} does not export {} (stale build?)'.format(path, name))
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.cp_abi_version() != ABI_VERSION:
        raise LibraryError('ABI version mismatch: library {}, bindings {}'.format(lib.cp_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


_background_ready = set()


def background_init(device_index):
    """The knot tables of the background kernels on device ``device_index`` (``cp_background_init``: one allocation and synchronous uploads, once per
    device), so that the entry points that read them stay asynchronous and allocation-free from their first call on."""
    if device_index not in _background_ready:
        check(load().cp_background_init(int(device_index)))
        _background_ready.add(device_index)


def check(status):
    """Map a status code onto the exception class the reference raises for the same condition."""
    if status == CP_OK:
        return
    msg = load().cp_last_error().decode('utf-8', 'replace')
    if status == CP_EINVAL:
        raise ValueError(msg)
    if status == CP_EUNSUPPORTED:
        raise NotImplementedError(msg)
    if status == CP_ENOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def as_double_p(array):
    return array.ctypes.data_as(_c_double_p)


def _complex_map(fn_name, z, *head):
    z = np.ascontiguousarray(z, dtype='c16')
    out = np.empty_like(z)
    fn = getattr(load(), fn_name)
    check(fn(*head, as_double_p(z.view('f8')), as_double_p(out.view('f8')), z.size))
    return out


def loggamma(z):
    """Principal branch of log Gamma on complex ``z`` (host; replaces scipy.special.loggamma in table setup)."""
    return _complex_map('cp_loggamma', z)


def gamma(z):
    return _complex_map('cp_gamma', z)


def kernel_eval(kind, param, z):
    """Mellin kernel U_K(z) (reference cosmoprimo/fftlog.py:666-766) on complex ``z``."""
    return _complex_map('cp_kernel_eval', z, int(kind), float(param))

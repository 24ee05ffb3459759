"""CPU: the plain-C++ part of the product under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on the pool):
special functions and FFTLog table setup behind the C ABI, and the per-thread phases of the fused FFTLog kernel run by the host emulator with
its LDS and rows as exact-size heap arrays -- every size class, padding mode, odd batches, NaN / Inf / tiny rows; the plan builder of the uniform-stretch spliced spline; the table
interpolation of the 'tabulated' engine (law of the knots, per-sample walk, laws the table does not follow).  See tests/host_san/san_driver.cpp."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_under_asan_ubsan():
    here = os.path.join(ROOT, 'tests', 'host_san')
    csrc = os.path.join(ROOT, 'cosmoprimo_amd', 'csrc')
    sources = [os.path.join(here, 'san_driver.cpp'), os.path.join(ROOT, 'tests', 'host_emu', 'emu_fftlog.cpp'), os.path.join(ROOT, 'tests', 'host_emu', 'emu_splice.cpp'), os.path.join(ROOT, 'tests', 'host_emu', 'emu_interp.cpp'),
               os.path.join(csrc, 'cp_special.cpp'), os.path.join(csrc, 'cp_fftlog_setup.cpp')]
    deps = sources + [os.path.join(csrc, h) for h in os.listdir(csrc) if h.endswith('.h')] + [os.path.join(ROOT, 'include', 'cosmoprimo_amd.h')]
    exe = os.path.join(here, 'san_driver')
    if not os.path.isfile(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(['g++', '-std=c++17', '-O1', '-g', '-ffp-contract=off', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-o', exe] + sources)
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    run = subprocess.run([exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert run.returncode == 0, run.stderr[-4000:]
    assert run.stdout.strip().endswith('ok')
    assert 'runtime error' not in run.stderr and 'AddressSanitizer' not in run.stderr, run.stderr[-4000:]

"""Steady-state summary of the headline kernel from a rocprofv3 --kernel-trace run of bench.py (written by scripts/gpu_profile.sh):

    python scripts/summarize_steady.py <tag>     # gpurun_out/prof_<tag>/trace -> profiles/<tag>_headline_kernel_stats.csv, profiles/<tag>_headline_steady.json

The bench loads the device for 300 ms before its warm-up and timed steps (the first tens of milliseconds after an idle period run ~10 % slower);
the dispatches of that ramp are listed apart from the steady-state ones, which are what `roofline.achieved` of the bench line is measured on.
The bench line printed by the SAME profiled run is kept beside the summary, so the HIP-event time and the profiler's time can be compared on
one run (under the profiler both read about 1-3 % longer than in an unprofiled run)."""
import csv
import glob
import os
import json
import shutil
import sys

tag = sys.argv[1]
src = 'gpurun_out/prof_%s' % tag


def newest(pattern):
    """gpurun merges every call's output into gpurun_out/: several runs of one tag leave several files, the last one counts"""
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)

RAMP_NS = 300e6
ROWS, BYTES_PER_ROW, PEAK = 100000, 2 * 8 * 2048, 8e12

stats = newest(src + '/trace/**/*kernel_stats.csv')
shutil.copy(stats, 'profiles/%s_headline_kernel_stats.csv' % tag)
trace = newest(src + '/trace/**/*kernel_trace.csv')
disp = []
for row in csv.DictReader(open(trace)):
    if 'fftlog_kernel' in row['Kernel_Name']:
        disp.append((int(row['Start_Timestamp']), int(row['End_Timestamp']), int(row['VGPR_Count']), int(row['Grid_Size_X']), int(row['LDS_Block_Size'])))
disp.sort()
t0 = disp[0][0]
dur = lambda rows: [(e - s) * 1e-6 for s, e, *_ in rows]      # noqa: E731  (ms)
ramp = [d for d in disp if d[0] - t0 < RAMP_NS]
steady = [d for d in disp if d[0] - t0 >= RAMP_NS]
# ... up to the first idle period: behind the timed steps the bench checks rows against the oracle on the host (seconds), and what it launches
# after that (the `value_api` measurement) starts from an idle device again -- those dispatches are listed apart
IDLE_NS = 20e6
cut = next((i for i in range(1, len(steady)) if steady[i][0] - steady[i - 1][1] > IDLE_NS), len(steady))
after_idle, steady = steady[cut:], steady[:cut]


def summary(rows):
    d = dur(rows)
    if not d:
        return None
    mean = sum(d) / len(d)
    return {'n': len(d), 'mean_ms': mean, 'min_ms': min(d), 'max_ms': max(d), 'frac_of_hbm_peak_at_mean': ROWS * BYTES_PER_ROW / (mean * 1e-3) / PEAK}


bench_line = None
for line in open(src + '/trace.log'):
    if line.startswith('{'):
        bench_line = json.loads(line)


def kernel_resources():
    """Registers and scratch of the headline kernel from its code object metadata (hipcc -S of the size group it is built in, the Makefile's flags):
    the profiler's trace columns VGPR_Count / LDS_Block_Size report the allocation granule of the architected half (128) and the STATIC LDS (0: the
    kernel's 69 920 bytes are dynamic), not what the kernel uses."""
    import re
    import subprocess
    import tempfile
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'cosmoprimo_amd', 'csrc')
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, 'g3.s')
        cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-mllvm', '-amdgpu-mfma-vgpr-form', '-DCP_INST_GROUP=3', '-S', '--cuda-device-only', '-o', asm,
               os.path.join(csrc, 'cp_fftlog_inst.hip')]
        try:
            subprocess.run(cmd, check=True, capture_output=True, timeout=900)
        except Exception as exc:      # no compiler here: say so instead of printing the profiler's columns
            return {'error': 'hipcc -S failed: %s' % exc}
        text = open(asm).read()
    # metadata entries: .name: <mangled>, then .vgpr_count / .vgpr_spill_count / .private_segment_fixed_size / .sgpr_count
    for block in text.split('  - .agpr_count:')[1:]:
        name = re.search(r'\.name:\s+(\S+)', block)
        if name and 'fftlog_kernelILi4096ELi16ELi3ELi1E' in name.group(1):
            get = lambda key: int(re.search(r'\.%s:\s+(\d+)' % key, block).group(1))      # noqa: E731
            return {'vgpr_count': get('vgpr_count'), 'vgpr_spill_count': get('vgpr_spill_count'), 'sgpr_count': get('sgpr_count'),
                    'scratch_bytes_per_lane': get('private_segment_fixed_size'), 'static_lds_bytes': get('group_segment_fixed_size'), 'source': 'code object metadata (hipcc -S)'}
    return {'error': 'kernel not found in the metadata'}


resources = kernel_resources()
out = {
    'command': 'rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary',
    'kernel': 'cpfft::fftlog_kernel<4096, 16, 3, 1>', 'rows_per_launch': ROWS, 'algorithmic_bytes_per_launch': ROWS * BYTES_PER_ROW,
    'kernel_resources': resources, 'lds_bytes_dynamic': None if bench_line is None else bench_line['config'].get('lds_bytes'),
    'grid_threads': disp[0][3], 'profiler_trace_columns': {'VGPR_Count': disp[0][2], 'LDS_Block_Size': disp[0][4],
                                                           'note': 'allocation granule of the architected registers and static LDS only: not the kernel\'s use'},
    'all_dispatches': summary(disp), 'ramp_first_300ms': summary(ramp), 'steady_state': summary(steady), 'after_an_idle_period_of_the_process': summary(after_idle),
    'same_run_bench_line': None if bench_line is None else {'kernel_ms_hip_events': bench_line['roofline']['kernel_ms'], 'frac': bench_line['roofline']['frac'],
                                                            'ms_per_step': bench_line['ms_per_step'], 'value': bench_line['value']},
}
if bench_line is not None and out['steady_state']:
    out['events_over_rocprof_steady_mean'] = bench_line['roofline']['kernel_ms'] / out['steady_state']['mean_ms']
json.dump(out, open('profiles/%s_headline_steady.json' % tag, 'w'), indent=1)
print(json.dumps(out, indent=1))

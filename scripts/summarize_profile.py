"""Turn gpurun_out/prof_<tag>/ (written by scripts/gpu_profile.sh on the GPU box) into the committed summaries
profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc_summary.json.   python scripts/summarize_profile.py <tag>"""
import collections
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

stats = newest(src + '/trace/**/*kernel_stats.csv')
shutil.copy(stats, 'profiles/%s_kernel_stats.csv' % tag)


def collect(folder, match):
    acc = collections.defaultdict(list)
    for f in [newest('%s/%s/**/*counter_collection.csv' % (src, folder))]:
        for row in csv.DictReader(open(f)):
            if match in row.get('Kernel_Name', ''):
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
    return {k: {'n_dispatches': len(v), 'mean': sum(v) / len(v)} for k, v in sorted(acc.items())}


counters = {}
for name in ('pmc_fetch', 'pmc_write', 'pmc_sq', 'pmc_sq2', 'pmc_grbm'):
    counters.update(collect(name, 'fftlog_kernel'))
# calibration of the memory counters on copies of exactly 2^30 bytes with 8- and 16-byte accesses (tools/fetch_calibration.hip)
cal = {}
for width in ('copy8', 'copy16'):
    c = {}
    c.update(collect('cal_fetch', width))
    c.update(collect('cal_write', width))
    cal[width] = {k: v['mean'] * 1024 / 2**30 for k, v in c.items()}      # counter (KB) x 1024 / bytes moved
rows = 100000
f8, w8 = cal['copy8'].get('FETCH_SIZE', 0.5), cal['copy8'].get('WRITE_SIZE', 1.)
read = counters['FETCH_SIZE']['mean'] * 1024 / f8
write = counters['WRITE_SIZE']['mean'] * 1024 / w8
out = {
    'command': 'rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary '
               '(separate passes: FETCH_SIZE | WRITE_SIZE | SQ_* | SQ_* | GRBM_GUI_ACTIVE; every fftlog_kernel dispatch of the run, device ramp included)',
    'kernel': 'cpfft::fftlog_kernel<4096, 16, 3, 1>',
    'rows_per_launch': rows,
    'counters': counters,
    'counter_calibration': {'what': 'FETCH_SIZE / WRITE_SIZE (KB x 1024) per byte actually moved, copies of 2^30 bytes (tools/fetch_calibration.hip)', **cal},
    'hbm_bytes_per_launch': {
        'read': read, 'write': write,
        'note': 'the kernel moves its rows with 8-byte-per-lane accesses: FETCH_SIZE and WRITE_SIZE (KB) are corrected with the factors measured on '
                'the 8-byte copy in the same profiling session (MI355X_MICROARCH.md, HBM section: calibrate other access widths on a known byte count)',
        'total': read + write, 'algorithmic': rows * 2048 * 8 * 2},
}
if 'GRBM_GUI_ACTIVE' in counters:
    # the clock the chip ran the kernel at: the counter is summed over the 8 XCDs; the duration of every dispatch comes from the SAME pass (the timestamps
    # of its own counter rows), dispatch by dispatch
    clocks, durations = [], []
    for row in csv.DictReader(open(newest('%s/pmc_grbm/**/*counter_collection.csv' % src))):
        if 'fftlog_kernel' in row.get('Kernel_Name', '') and row['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            ns = int(row['End_Timestamp']) - int(row['Start_Timestamp'])
            if ns > 0:
                durations.append(ns * 1e-6)
                clocks.append(float(row['Counter_Value']) / 8 / (ns * 1e-9))
    if clocks:
        out['kernel_ms_profiled'] = sum(durations) / len(durations)
        out['effective_clock_GHz'] = sum(clocks) / len(clocks) / 1e9
    out['note_clock'] = ('effective clock = GRBM_GUI_ACTIVE / 8 XCDs / the dispatch\'s own duration in the same pass, mean over its dispatches (reads high on '
                         'dispatches shorter than 0.3 ms: MI355X_MICROARCH.md, DVFS give-back)')
json.dump(out, open('profiles/%s_pmc_summary.json' % tag, 'w'), indent=1)
print(json.dumps({k: out[k] for k in ('counter_calibration', 'hbm_bytes_per_launch')}, indent=1))
for k, v in counters.items():
    print('%-26s n=%3d mean=%.6g' % (k, v['n_dispatches'], v['mean']))

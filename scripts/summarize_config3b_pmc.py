"""gpurun_out/prof3b_<tag>/ (scripts/profile_config3b_pmc.sh) -> profiles/<tag>_config3b_kernel_stats.csv and profiles/<tag>_config3b_traffic.json:
HBM bytes per sigma_rz call of config 3B, kernel by kernel, next to the algorithmic bytes.   python scripts/summarize_config3b_pmc.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = 'gpurun_out/prof3b_%s' % tag
NCOSMO, ALGORITHMIC = 10000, 10000 * (500 * 30 * 8 + 256 * 64 * 8)


def newest(pattern):
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


shutil.copy(newest(src + '/trace/**/*kernel_stats.csv'), 'profiles/%s_config3b_kernel_stats.csv' % tag)


def per_kernel(folder):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(newest('%s/%s/**/*counter_collection.csv' % (src, folder)))):
        acc[(row['Kernel_Name'], row['Counter_Name'])].append(float(row['Counter_Value']))
    return acc


def calibration(folder, counter):
    acc = per_kernel(folder)
    return {k[0].replace('void ', '').split('(')[0]: sum(v) / len(v) * 1024 / 2**30 for k, v in acc.items() if k[1] == counter}


cal_f, cal_w = calibration('cal_fetch', 'FETCH_SIZE'), calibration('cal_write', 'WRITE_SIZE')
f8 = [v for k, v in cal_f.items() if 'copy8' in k][0]
w8 = [v for k, v in cal_w.items() if 'copy8' in k][0]
# the kernels of one sigma_rz call, by the names the trace gives them; per-call bytes = mean over the dispatches of the timed region and its ramp
mine = ('tables_rows', 'fftlog', 'linop', 'spline')
fetch, write = per_kernel('pmc_fetch'), per_kernel('pmc_write')
kernels = {}
for (name, counter), values in list(fetch.items()) + list(write.items()):
    short = name.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
    if not any(m in short for m in mine):
        continue
    entry = kernels.setdefault(short, {'dispatches': len(values)})
    entry['read_bytes' if counter == 'FETCH_SIZE' else 'write_bytes'] = sum(values) / len(values) * 1024 / (f8 if counter == 'FETCH_SIZE' else w8)
# kernels that run once per table set (second derivatives of the tables) are set-up, not part of a call: they have few dispatches
calls = max(v['dispatches'] for v in kernels.values())
per_call = {k: v for k, v in kernels.items() if v['dispatches'] >= calls // 2}
total = sum(v.get('read_bytes', 0.) + v.get('write_bytes', 0.) for v in per_call.values())
out = {'command': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 tools/profile_secondary.py 3b',
       'workload': 'config 3B: sigma_rz 256 r x 64 z of %d tabulated P(k, z) of 500 k x 30 z' % NCOSMO,
       'counter_calibration': {'what': 'FETCH_SIZE / WRITE_SIZE (KB x 1024) per byte actually moved by 8-byte-per-lane copies of 2^30 bytes in the same session '
                                       '(tools/fetch_calibration.hip); the kernels here mix 8- and 16-byte accesses, so bytes are good to the ratio of the two rows',
                               'fetch': cal_f, 'write': cal_w},
       'kernels_per_call': per_call, 'kernels_per_table_set': {k: v for k, v in kernels.items() if k not in per_call},
       'hbm_bytes_per_call': total, 'algorithmic_bytes_per_call': ALGORITHMIC, 'traffic_over_algorithmic': total / ALGORITHMIC}
json.dump(out, open('profiles/%s_config3b_traffic.json' % tag, 'w'), indent=1)
print(json.dumps({k: out[k] for k in ('kernels_per_call', 'hbm_bytes_per_call', 'traffic_over_algorithmic')}, indent=1))

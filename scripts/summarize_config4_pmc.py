"""gpurun_out/prof4pmc_<tag>/ (scripts/profile_config4_pmc.sh) -> profiles/<tag>_config4_traffic.json: HBM bytes per vector of config 4, kernel by kernel and
filter by filter (the package's own kernels; framework and runtime kernels summed apart), next to the 16 384 algorithmic bytes of a vector.
    python scripts/summarize_config4_pmc.py <tag>"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench      # noqa: E402  (the chunk size the runs used)

tag = sys.argv[1]
src = 'gpurun_out/prof4pmc_%s' % tag
VECTORS = (1 + bench.CONFIG4_PROFILE_CHUNKS) * bench.CONFIG4_CHUNK      # tools/profile_secondary.py 4w / 4b: one untimed chunk and the timed ones


def newest(pattern):
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


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
out = {'command': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 tools/profile_secondary.py 4w | 4b',
       'vectors_per_run': VECTORS, 'chunk': bench.CONFIG4_CHUNK, 'library_sha256_16': bench._library_sha(),
       'counter_calibration': {'what': 'FETCH_SIZE / WRITE_SIZE (KB x 1024) per byte moved by 8-byte-per-lane copies of 2^30 bytes in the same session (tools/fetch_calibration.hip)',
                               'fetch_copy8': f8, 'write_copy8': w8},
       'algorithmic_bytes_per_vector': 16384}
for which, engine in (('4w', 'wallish2018'), ('4b', 'brieden2022')):
    kernels, other = {}, {'read_bytes_per_vector': 0., 'write_bytes_per_vector': 0.}
    for folder, counter, key, cal in (('%s_fetch' % which, 'FETCH_SIZE', 'read_bytes_per_vector', f8), ('%s_write' % which, 'WRITE_SIZE', 'write_bytes_per_vector', w8)):
        for (name, cname), values in per_kernel(folder).items():
            if cname != counter:
                continue
            nbytes = sum(values) * 1024 / cal / VECTORS
            if 'at::native' in name or 'rocclr' in name or 'elementwise' in name:
                other[key] += nbytes
                continue
            short = name.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
            entry = kernels.setdefault(short, {'dispatches': len(values)})
            entry[key] = entry.get(key, 0.) + nbytes
    kernels = dict(sorted(kernels.items(), key=lambda kv: -(kv[1].get('read_bytes_per_vector', 0.) + kv[1].get('write_bytes_per_vector', 0.))))
    total = sum(v.get('read_bytes_per_vector', 0.) + v.get('write_bytes_per_vector', 0.) for v in kernels.values())
    out[engine] = {'kernels': kernels, 'framework_and_runtime_kernels': other, 'hbm_bytes_per_vector': total, 'traffic_over_algorithmic': total / 16384.}
json.dump(out, open('profiles/%s_config4_traffic.json' % tag, 'w'), indent=1)
for engine in ('wallish2018', 'brieden2022'):
    print(engine, '%.0f B per vector' % out[engine]['hbm_bytes_per_vector'])
    for k, v in list(out[engine]['kernels'].items())[:8]:
        print('   %-50s read %8.0f  write %8.0f' % (k[:50], v.get('read_bytes_per_vector', 0.), v.get('write_bytes_per_vector', 0.)))

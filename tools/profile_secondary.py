"""One of the secondary configs of bench.py on its own, for a kernel trace (development aid):
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c3 -- python3 tools/profile_secondary.py 3
Configs: 3, 3b, 4 (both filters, chunks of bench.CONFIG4_CHUNK vectors), 4w / 4b (one filter, for counter collection), 5."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    import torch
    import bench
    import cosmoprimo_amd as cp
    which = sys.argv[1] if len(sys.argv) > 1 else '3'
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    if which == '3':
        out = bench.config3(cp, torch, dev, reps=20)
    elif which == '3b':
        out = bench.config3b(cp, torch, dev, reps=10)
    elif which in ('4w', '4b'):      # one filter only (counter collection per filter)
        engine = 'wallish2018' if which == '4w' else 'brieden2022'
        bench.RAMP_S = 0.      # one untimed chunk, then the timed ones
        n = bench.CONFIG4_PROFILE_CHUNKS * bench.CONFIG4_CHUNK      # (steady state: the first chunk builds the filter's plans and operators -- uploads that happen once per filter object)
        out = bench.config4(cp, torch, dev, bench.eh_parameters(n, 2, torch, dev), engines=(engine,), spot_check=False)
        out['vectors_through_the_filter_in_this_process'] = bench.CONFIG4_CHUNK + n
        out['chunk'] = bench.CONFIG4_CHUNK
    elif which == '4':
        out = bench.config4(cp, torch, dev, bench.eh_parameters(2 * bench.CONFIG4_CHUNK, 2, torch, dev))
    else:
        out = bench.config5(torch, dev, *bench.config5_samples(1250000, 3, torch, dev), reps=20)
    print(json.dumps(out))


if __name__ == '__main__':
    main()

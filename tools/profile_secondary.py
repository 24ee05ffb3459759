"""One of the secondary configs of bench.py on its own, for a kernel trace (development aid):
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c3 -- python3 tools/profile_secondary.py 3
Configs: 3, 3b, 4 (both filters, one 16 384-vector chunk timed 4 times), 5."""
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
        bench.RAMP_S = 0.      # one untimed chunk, then the four timed ones: 5 x 16 384 vectors go through the filter in this process
        out = bench.config4(cp, torch, dev, bench.eh_parameters(4 * 16384, 2, torch, dev), engines=(engine,))
        out['vectors_through_the_filter_in_this_process'] = 5 * 16384
    elif which == '4':
        out = bench.config4(cp, torch, dev, bench.eh_parameters(4 * 16384, 2, torch, dev))
    else:
        out = bench.config5(torch, dev, *bench.config5_samples(1250000, 3, torch, dev), reps=20)
    print(json.dumps(out))


if __name__ == '__main__':
    main()

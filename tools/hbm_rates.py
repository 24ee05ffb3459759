"""What the memory system of the box gives plain streams of 5.24 GB (torch kernels): fill, copy, reduction.    python tools/hbm_rates.py"""
import torch, time
n = 5242880000 // 8
x = torch.empty(n, dtype=torch.float64, device='cuda')
y = torch.empty(n, dtype=torch.float64, device='cuda')
def timeit(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
t = timeit(lambda: x.fill_(1.5)); print('fill 5.24 GB: %.3f ms -> %.2f TB/s write' % (t, 5.24288 / t))
t = timeit(lambda: y.copy_(x)); print('copy 5.24 GB: %.3f ms -> %.2f TB/s read + %.2f TB/s write' % (t, 5.24288 / t, 5.24288 / t))
t = timeit(lambda: x.sum()); print('sum 5.24 GB: %.3f ms -> %.2f TB/s read' % (t, 5.24288 / t))

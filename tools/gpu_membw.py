"""Device memory ceilings on this box (fill / copy / read-reduce), for judging roofline fractions."""
import torch, time
dev = "cuda"
n = 1 << 30  # 1 GiB
a = torch.empty(n, dtype=torch.uint8, device=dev)
b = torch.empty(n, dtype=torch.uint8, device=dev)
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
t = timeit(lambda: a.fill_(7)); print("fill  1 GiB: %.3f ms -> %.2f TB/s (write only)" % (t * 1e3, n / t / 1e12))
t = timeit(lambda: b.copy_(a)); print("copy  1 GiB: %.3f ms -> %.2f TB/s (read+write bytes)" % (t * 1e3, 2 * n / t / 1e12))
af = a.view(torch.int32)
t = timeit(lambda: af.sum()); print("sum   1 GiB: %.3f ms -> %.2f TB/s (read only)" % (t * 1e3, n / t / 1e12))

"""Achievable HBM bandwidth of this device for plain streaming kernels (read + write bytes / time): the practical ceiling the
FFT passes are compared with, next to the 8 TB/s peak of the data sheet."""
import time
import torch

dev = torch.device("cuda", 0)
n = 2 ** 31  # floats: 8.6 GB, the size of a C3 volume
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t = timed(lambda: b.copy_(a))
print(f"copy   (read 8.6 GB + write 8.6 GB): {t * 1e3:.2f} ms, {2 * n * 4 / t / 1e12:.2f} TB/s", flush=True)
t = timed(lambda: a.mul_(1.0001))
print(f"scale in place (read + write):        {t * 1e3:.2f} ms, {2 * n * 4 / t / 1e12:.2f} TB/s", flush=True)
t = timed(lambda: torch.add(a, b, out=b))
print(f"add    (2 reads + 1 write):           {t * 1e3:.2f} ms, {3 * n * 4 / t / 1e12:.2f} TB/s", flush=True)
t = timed(lambda: a.sum())
print(f"sum    (read only):                   {t * 1e3:.2f} ms, {n * 4 / t / 1e12:.2f} TB/s", flush=True)
t = timed(lambda: b.fill_(1.0))
print(f"fill   (write only):                  {t * 1e3:.2f} ms, {n * 4 / t / 1e12:.2f} TB/s", flush=True)

"""What this box's HBM sustains for plain streaming (torch kernels, 4 GiB arrays): read (sum), write (fill), copy - the yardstick for
the memory-bound launches of the cooperative gradient (the D-sized products over tiles read ~ 4 TB/s)."""
import json, torch
dev = torch.device("cuda:0")
n = 1 << 30   # floats: 4 GiB
x = torch.empty(n, device=dev, dtype=torch.float32).normal_()
y = torch.empty_like(x)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
gb = 4 * n / 1e9
out = {"read_sum_TBps": gb / t(lambda: x.sum()) / 1e3, "write_fill_TBps": gb / t(lambda: y.fill_(1.0)) / 1e3,
       "copy_TBps_read_plus_write": 2 * gb / t(lambda: y.copy_(x)) / 1e3, "add_TBps_2r1w": 3 * gb / t(lambda: torch.add(x, y, out=y)) / 1e3}
print(json.dumps({k: round(v, 2) for k, v in out.items()}))

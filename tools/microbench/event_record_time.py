"""Host cost of torch.cuda.Event.record() behind a batch of small launches, with and without a device -> pinned-host
write in front of it (the step engine's numerics word is written by a kernel into device-mapped host memory)."""
import time, torch
dev = "cuda"
x = torch.zeros(1024, device=dev)
pin = torch.zeros(1, dtype=torch.int32).pin_memory()
flag = torch.zeros(1, dtype=torch.int32, device=dev)
evs = [torch.cuda.Event() for _ in range(2)]
def run(pinned, n=300, launches=25, after=0):
    t_rec = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        for _ in range(launches): x.add_(1.0)
        if pinned: pin.copy_(flag, non_blocking=True)
        a = time.perf_counter(); evs[i & 1].record(); t_rec += time.perf_counter() - a
        for _ in range(after): x.add_(1.0)
        time.sleep(0.0002)  # host busy elsewhere (the rest of the step)
    torch.cuda.synchronize()
    return t_rec / n * 1e6, (time.perf_counter() - t0) / n * 1e6
for pinned in (False, True):
    for after in (0, 25):
        r, tot = run(pinned, after=after)
        print(f"pinned write {pinned}, {after} launches behind the record: record() {r:6.1f} us, iteration {tot:7.1f} us")

import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/proc/loadavg"):
    try: print(f, open(f).read().strip())
    except OSError as e: print(f, "n/a")
x = torch.randn(2, 64, 256, 256); w = torch.randn(64, 64, 3, 3)
for t in (4, 8, 16, 32, 64, 128):
    torch.set_num_threads(t)
    torch.nn.functional.conv2d(x, w, padding=1)
    t0 = time.perf_counter()
    for _ in range(5): torch.nn.functional.conv2d(x, w, padding=1)
    print("threads", t, "conv ms", (time.perf_counter() - t0) / 5 * 1e3)

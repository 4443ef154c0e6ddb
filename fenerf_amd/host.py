"""What the host side of this package may assume about its CPUs.  A container often shows every logical CPU of the machine while its cgroup
grants a fraction of them (the GPU boxes of this project: 256 logical CPUs, `cpu.max 1600000 100000` = 16): torch sizes its intra-op thread
pool by the former, and every small CPU tensor operation then costs milliseconds (tools/exp/callers_timing.py, dump_timing.py:
torch.argmax of one 256 x 256 label image 50-70 ms, one `/ 255.` 5-15 ms, save_image of a 128 x 128 image 4.7 ms).  The inference callers
avoid such operations (numpy on host images); respect_cpu_quota() is for everything else a script does on the CPU."""
import os


def effective_host_cores():
    """The host cores this process may actually use: the smallest of os.cpu_count(), the scheduler affinity mask and the cgroup CPU quota
    (v2 cpu.max / v1 cpu.cfs_quota_us).  The GPU boxes of this pool show 256 logical CPUs with a quota of 16 (`cpu.max 1600000 100000`):
    256 OpenMP threads on 16 cores' worth of time ran the torch-CPU oracle 16 x SLOWER than 16 threads (tools/exp/cpu_threads_probe.py,
    round 6: 118 vs 1,765 rays/s at 64x64x24+24)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota:
        n = min(n, max(1, int(quota + 0.5)))
    return n


def respect_cpu_quota():
    """torch.set_num_threads(effective_host_cores()) if torch's intra-op pool is larger than what this process may use.  Process-wide, hence
    never called on import: the command-line front ends under tools/ call it first thing.  -> the thread count in force."""
    import torch
    n = effective_host_cores()
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()

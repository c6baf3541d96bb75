"""Host-thread placement for HOST-BOUND callers (opt-in; nothing here is called by the library itself).

A decoder step launched call by call from Python -- an unchanged caller of ``DecoderSplattingCUDA`` -- is two host threads
taking turns: the caller's and autograd's device thread.  Where the kernel puts them is luck, and it matters: on a
two-socket EPYC host the same command ran the BASELINE config-2 step in 0.358 ms per step with both threads in one L3
group (one CCD), 0.376 - 0.380 spread over one socket and 0.420 with one thread on each socket (round 6,
``tools/child_probe.sh``; the GPU's kernels take 0.354).  ``bind_to_gpu_l3`` pins the calling process to ONE L3 group
of the NUMA node the GPU hangs off -- what ``numactl --physcpubind`` would do from outside.  It binds the CALLING THREAD
(Linux affinity is per thread) and whatever that thread creates afterwards: call it before the first backward pass
(autograd's thread then inherits it) but AFTER torch's CPU thread pool has done its first parallel work -- a pool created
under the binding puts its 100+ threads on one CCD (bench.py did that for one run: 15 times the wall time of its input
generation) -- and widen it again (``os.sched_setaffinity(0, saved)``) around CPU-heavy work such as starting a
DataLoader's workers.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Optional

_SYS = Path("/sys")


def parse_cpulist(text: str) -> list[int]:
    """'0-7,128-135' -> [0, ..., 7, 128, ..., 135] (the kernel's cpulist format)."""
    out: list[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def format_cpulist(cpus) -> str:
    cpus = sorted(set(cpus))
    runs, start = [], None
    for i, c in enumerate(cpus):
        if start is None:
            start = c
        if i + 1 == len(cpus) or cpus[i + 1] != c + 1:
            runs.append(f"{start}-{c}" if c != start else f"{c}")
            start = None
    return ",".join(runs)


def l3_groups(cpus, sys_root: Path = _SYS) -> list[list[int]]:
    """The distinct L3 groups (``cache/index3/shared_cpu_list``) the given CPUs belong to, each cut down to those CPUs,
    in the order of their lowest CPU."""
    want, seen, groups = set(cpus), set(), []
    for c in sorted(want):
        if c in seen:
            continue
        f = sys_root / "devices" / "system" / "cpu" / f"cpu{c}" / "cache" / "index3" / "shared_cpu_list"
        try:
            grp = [x for x in parse_cpulist(f.read_text()) if x in want]
        except (OSError, ValueError):
            grp = [c]
        grp = grp or [c]
        seen.update(grp)
        groups.append(sorted(grp))
    return groups


def pci_address(device_index: int) -> str:
    import torch
    p = torch.cuda.get_device_properties(device_index)
    return f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"


def choose_group(local: list[int], allowed, k: int, n: int, sys_root: Path = _SYS) -> Optional[list[int]]:
    """The L3 group for the k-th of the n GPUs of one NUMA node: the node's groups are dealt out evenly, so the ranks of a
    node do not sit on each other's cores.  `allowed`: the process's current affinity (a cpuset may be narrower than the
    node).  None when nothing is left."""
    cpus = [c for c in local if c in set(allowed)]
    if not cpus:
        return None
    groups = l3_groups(cpus, sys_root)
    return groups[(k * len(groups)) // max(n, 1) % len(groups)]


def bind_to_gpu_l3(device_index: Optional[int] = None, sys_root: Path = _SYS) -> Optional[str]:
    """Pin the calling thread (and the threads it creates from now on) to one L3 group of the CPUs next to HIP
    device `device_index` (default: the current device).  Returns the cpulist it was bound to, or None when the topology
    cannot be read (no sysfs entry, a cpuset without local CPUs, no ``sched_setaffinity``) -- nothing is changed then."""
    try:
        import torch
        i = torch.cuda.current_device() if device_index is None else int(device_index)
        dev = sys_root / "bus" / "pci" / "devices"
        mine = dev / pci_address(i)
        local = parse_cpulist((mine / "local_cpulist").read_text())
        node = (mine / "numa_node").read_text().strip()
        same = []                                   # the GPUs of this NUMA node, in device order
        for j in range(torch.cuda.device_count()):
            try:
                if (dev / pci_address(j) / "numa_node").read_text().strip() == node:
                    same.append(j)
            except OSError:
                pass
        k, n = (same.index(i), len(same)) if i in same else (0, 1)
        grp = choose_group(local, os.sched_getaffinity(0), k, n, sys_root)
        if not grp:
            return None
        os.sched_setaffinity(0, grp)
        return format_cpulist(grp)
    except Exception:                               # noqa: BLE001  (placement is an optimisation: never an error)
        return None

"""What the launchers need to know about the node BEFORE any process touches a GPU, read from sysfs (the KFD
topology): how many GPUs there are, and which host NUMA node each hangs off -- so that `--gpus N` can be refused
when N devices are not there (a message and exit 2, not a rendezvous that hangs), and a rank's host-side stages
(parser threads, page-locked staging, the uploads) can be pinned next to its GPU.  No HIP call, no torch import."""
import glob
import os

KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"


def _props(path):
    out = {}
    try:
        with open(path) as f:
            for line in f:
                k, _, v = line.strip().partition(" ")
                if k and v:
                    try:
                        out[k] = int(v)
                    except ValueError:
                        pass
    except OSError:
        pass
    return out


def kfd_gpus(root=KFD_NODES):
    """[{node, bdf, numa}] of the GPU nodes (simd_count > 0) in KFD order = HIP device order; [] when the topology
    is not readable (no amdgpu driver, a container without /sys)."""
    gpus = []
    for d in sorted(glob.glob(os.path.join(root, "*")), key=lambda p: int(os.path.basename(p)) if os.path.basename(p).isdigit() else 1 << 30):
        p = _props(os.path.join(d, "properties"))
        if p.get("simd_count", 0) <= 0:
            continue
        loc, dom = p.get("location_id", 0), p.get("domain", 0)
        bdf = f"{dom:04x}:{(loc >> 8) & 0xFF:02x}:{(loc >> 3) & 0x1F:02x}.{loc & 7:x}"
        numa = -1
        try:
            numa = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip())
        except (OSError, ValueError):
            pass
        gpus.append({"node": os.path.basename(d), "bdf": bdf, "numa": numa})
    return gpus


def visible_gpus(root=KFD_NODES, env=None):
    """Number of GPUs a new process would see (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES narrow
    it), or None when it cannot be told without touching the GPU."""
    env = os.environ if env is None else env
    gpus = kfd_gpus(root)
    if not gpus:
        return None
    n = len(gpus)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(var)
        if v is not None:
            ids = [x for x in v.split(",") if x.strip() != ""]
            n = min(n, len(ids))
    return n


def numa_cpus(node):
    """The CPUs of a host NUMA node as a set (empty when unknown)."""
    try:
        txt = open(f"/sys/devices/system/node/node{int(node)}/cpulist").read().strip()
    except (OSError, ValueError):
        return set()
    cpus = set()
    for part in txt.split(","):
        a, _, b = part.partition("-")
        if a.strip().isdigit():
            cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def kfd_index_of(local_rank, n_gpus, env=None):
    """Index into the KFD GPU list of HIP device `local_rank` of a process started with this environment:
    ROCR_VISIBLE_DEVICES selects and orders the agents, HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES then index into that
    selection.  None when it cannot be told (entries that are not plain indices, e.g. GPU UUIDs; an index out of range)."""
    env = os.environ if env is None else env
    order = list(range(n_gpus))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(var)
        if v is None:
            continue
        ids = [x.strip() for x in v.split(",") if x.strip() != ""]
        if not all(x.isdigit() for x in ids):
            return None
        picked = []
        for x in ids:
            if int(x) >= len(order):
                break       # the runtime stops at the first invalid entry
            picked.append(order[int(x)])
        order = picked
    return order[local_rank] if 0 <= local_rank < len(order) else None


def pin_to_gpu_numa(local_rank, root=KFD_NODES):
    """Pin the calling thread (and the threads it starts later: call it before the process group, RCCL and the parser
    pool exist) to the CPUs of the NUMA node GPU `local_rank` hangs off:
    eight ranks on one host otherwise share one memory system for their parser pools and uploads
    (profiles/r03_host_stage_probes.txt).  Best effort: returns what was done, never raises.  LRB_NUMA_PIN=0 turns
    it off."""
    if os.environ.get("LRB_NUMA_PIN", "1") == "0":
        return {"pinned": False, "why": "LRB_NUMA_PIN=0"}
    gpus = kfd_gpus(root)
    if not gpus:
        return {"pinned": False, "why": "KFD topology not readable"}
    # HIP device `local_rank` is KFD GPU `local_rank` only while no *_VISIBLE_DEVICES reorders or subsets the node
    idx = kfd_index_of(local_rank, len(gpus))
    if idx is None:
        return {"pinned": False, "why": "cannot map the local rank through *_VISIBLE_DEVICES"}
    local_rank = idx
    node = gpus[local_rank]["numa"]
    if node < 0:
        return {"pinned": False, "why": "the GPU reports no NUMA node", "bdf": gpus[local_rank]["bdf"]}
    cpus = numa_cpus(node)
    try:
        allowed = os.sched_getaffinity(0)
        want = cpus & allowed
        if not want or want == allowed:
            return {"pinned": False, "why": "one NUMA node" if want else "no CPU of that node allowed", "numa": node}
        os.sched_setaffinity(0, want)
        return {"pinned": True, "numa": node, "cpus": len(want), "bdf": gpus[local_rank]["bdf"]}
    except (AttributeError, OSError) as e:
        return {"pinned": False, "why": f"{type(e).__name__}: {e}", "numa": node}


def cpu_budget():
    """CPUs this process may actually use: the affinity mask, cut by the cgroup's CPU quota (cpu.max of cgroup v2,
    cfs_quota_us / cfs_period_us of v1) -- a container that shows 256 CPUs may be given the time of 16, and a thread pool
    sized from the former is throttled as a whole."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(p)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)

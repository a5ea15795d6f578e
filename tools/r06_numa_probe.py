"""Developer probe (round 6): NUMA placement of the reader threads against the read + host-to-device pipeline's rate."""
import glob, os, subprocess, sys, tempfile, threading, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import torch
print(subprocess.run("lscpu | grep -i 'numa\\|socket\\|model name'", shell=True, capture_output=True, text=True).stdout)
p = torch.cuda.get_device_properties(0)
bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
node = open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip() if os.path.exists("/sys/bus/pci/devices/%s/numa_node" % bdf) else "?"
print("GPU", bdf, "numa_node", node, "| this process may run on", len(os.sched_getaffinity(0)), "cpus")
cpus = {}
for d in glob.glob("/sys/devices/system/node/node*"):
    lst = open(d + "/cpulist").read().strip()
    s = set()
    for part in lst.split(","):
        a, _, b = part.partition("-")
        s.update(range(int(a), int(b or a) + 1))
    cpus[int(d.rsplit("node", 1)[1])] = s
print({k: (min(v), max(v), len(v)) for k, v in cpus.items()})
root = tempfile.mkdtemp(prefix="probe_")
n, size = 20, 100 * 64 * 64 * 15 * 4
data = np.random.default_rng(0).random(size // 4, dtype=np.float32)
paths = []
for i in range(n):
    q = os.path.join(root, "f%d.bin" % i)
    with open(q, "wb") as f:
        f.write(data.tobytes())
    paths.append(q)
dev = torch.device("cuda")
dst = torch.empty(n * size // 4, dtype=torch.float32, device=dev)
def run(th, affinity, n_streams=2, slice_mb=8):
    pinned = {}
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    def init():
        if affinity is not None:
            os.sched_setaffinity(0, affinity)
    ex = ThreadPoolExecutor(th, initializer=init)
    def work(i):
        tid = threading.get_ident()
        if tid not in pinned:
            pinned[tid] = [torch.empty(size // 4, dtype=torch.float32).pin_memory() for _ in range(2)] + [0, [None, None], streams[len(pinned) % n_streams]]
        st = pinned[tid]
        k = st[2]; st[2] ^= 1
        if st[3][k] is not None:
            st[3][k].synchronize()
        buf = st[k]
        mv = memoryview(buf.numpy()).cast("B")
        sl = slice_mb << 20
        with open(paths[i], "rb", buffering=0) as f, torch.cuda.stream(st[4]):
            for o in range(0, size, sl):
                e = min(size, o + sl)
                f.readinto(mv[o:e])
                dst[i * (size // 4) + o // 4:i * (size // 4) + e // 4].copy_(buf[o // 4:e // 4], non_blocking=True)
            ev = torch.cuda.Event(); ev.record(st[4]); st[3][k] = ev
    best = 1e9
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        list(ex.map(work, range(n)))
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    ex.shutdown()
    return best * 1e3
allc = os.sched_getaffinity(0)
print("no affinity, 8 threads: %.1f ms" % run(8, None))
for k, v in sorted(cpus.items()):
    a = v & allc
    if a:
        print("threads on node %d (%d cpus), 8 threads: %.1f ms; 12 threads: %.1f ms" % (k, len(a), run(8, a), run(12, a)))
print("no affinity, 8 threads, 4 streams: %.1f ms; 16 threads: %.1f ms" % (run(8, None, 4), run(16, None)))

"""H2D copy rate from page-locked buffers allocated (first-touched) on each NUMA node.   python tools/debug/numa_probe.py"""
import glob, os, time
import torch

dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
props = torch.cuda.get_device_properties(0)
print("GPU pci:", getattr(props, "pci_bus_id", None), getattr(props, "pci_device_id", None), getattr(props, "pci_domain_id", None))
for p in sorted(glob.glob("/sys/class/drm/card*/device/numa_node")):
    print(p, open(p).read().strip(), open(os.path.join(os.path.dirname(p), "vendor")).read().strip())
nodes = sorted(int(os.path.basename(p)[4:]) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))
print("nodes", nodes)


def cpus(node):
    out = []
    for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


stage = torch.empty((3000, 4000), dtype=torch.float32, device=dev)
full = os.sched_getaffinity(0)
for node in nodes:
    os.sched_setaffinity(0, cpus(node))
    time.sleep(0.01)
    bufs = [torch.zeros((3000, 4000), dtype=torch.float32).pin_memory() for _ in range(8)]
    os.sched_setaffinity(0, full)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in bufs:
            stage.copy_(b, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"pinned buffers first-touched on node {node}: {8 * 48e6 / dt / 1e9:.1f} GB/s")

#!/usr/bin/env python3
"""Where page-locked blocks land and what a device-to-host copy into them reaches (round 6, the bimodal one-file decode): N blocks of 52 MB from
hipHostMalloc, taken one by one while the calling thread is moved from CPU to CPU (first CPU of every NUMA node in turn), the NUMA node of
each block's pages (/proc/self/numa_maps) and the GB/s of a 46 MB hipMemcpy D2H into it (best of 5).  usage (GPU box): python tools/pinned_probe.py"""
import ctypes as C, json, os, re, time
hip = C.CDLL("libamdhip64.so")
def chk(rc):
    assert rc == 0, rc
N, SZ = 46 << 20, 52 << 20
chk(hip.hipSetDevice(0))
d = C.c_void_p(); chk(hip.hipMalloc(C.byref(d), N))
nodes = sorted(int(m.group(1)) for m in (re.match(r"node(\d+)$", x) for x in os.listdir("/sys/devices/system/node")) if m)
first_cpu = {}
for nd in nodes:
    lst = open(f"/sys/devices/system/node/node{nd}/cpulist").read().strip()
    first_cpu[nd] = int(re.split(r"[-,]", lst)[0])
allowed = os.sched_getaffinity(0)
bus = C.create_string_buffer(64); chk(hip.hipDeviceGetPCIBusId(bus, 64, 0))
gpu_node = open("/sys/bus/pci/devices/%s/numa_node" % bus.value.decode().lower()).read().strip()
def node_of(addr):
    for line in open("/proc/self/numa_maps"):
        if line.startswith("%x " % addr):
            return dict(kv.split("=") for kv in line.split()[2:] if "=" in kv and kv.startswith("N"))
    return None
rows = []
for i in range(12):
    nd = nodes[i % len(nodes)]
    cpu = first_cpu[nd]
    if cpu in allowed:
        os.sched_setaffinity(0, {cpu})
    time.sleep(0.01)
    p = C.c_void_p(); chk(hip.hipHostMalloc(C.byref(p), SZ, 0))
    best = 0.0
    for _ in range(5):
        t0 = time.perf_counter(); chk(hip.hipMemcpy(p, d, N, 2)); dt = time.perf_counter() - t0
        best = max(best, N / dt / 1e9)
    rows.append({"block": i, "thread_on_cpu": cpu if cpu in allowed else None, "thread_node": nd, "pages_on": node_of(p.value), "d2h_gb_s": round(best, 1)})
    os.sched_setaffinity(0, allowed)
print(json.dumps({"gpu_numa_node": gpu_node, "numa_nodes": nodes, "cpus_allowed": len(allowed), "blocks": rows}, indent=1))

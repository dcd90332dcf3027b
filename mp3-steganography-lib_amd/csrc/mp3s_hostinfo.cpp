// What this process may use of the host it shares with the other ranks (one process per GPU): how many CPUs
// (affinity mask and cgroup quota), how many ranks they are shared with, and which of them sit on the GPU's NUMA node.
// The reference has no counterpart (one Python thread); this decides how many threads a rank walks / scans with and where
// its page-locked staging lives.
#include <sched.h>
#include <unistd.h>

#include <cctype>
#include <fstream>
#include <sstream>

#include "mp3s_internal.h"

static std::string read_line(const std::string &path)
{
    std::ifstream f(path);
    std::string s;
    if (f) std::getline(f, s);
    return s;
}

int host_cpus_allowed()
{
    cpu_set_t set;
    int n = 0;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::max(1L, sysconf(_SC_NPROCESSORS_ONLN));
    // cgroup v2: "max 100000" or "<quota> <period>"; v1: cpu.cfs_quota_us / cpu.cfs_period_us
    double quota = -1, period = 100000;
    {
        std::istringstream is(read_line("/sys/fs/cgroup/cpu.max"));
        std::string q;
        if (is >> q >> period && q != "max") quota = atof(q.c_str());
    }
    if (quota < 0) {
        const std::string q = read_line("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), p = read_line("/sys/fs/cgroup/cpu/cpu.cfs_period_us");
        if (!q.empty() && !p.empty() && atof(q.c_str()) > 0) { quota = atof(q.c_str()); period = atof(p.c_str()); }
    }
    if (quota > 0 && period > 0) n = std::min(n, std::max(1, (int)(quota / period + 0.5)));
    return n;
}

int local_world_size()
{
    const char *v = getenv("LOCAL_WORLD_SIZE");
    const int n = v ? atoi(v) : 1;
    return n > 0 ? n : 1;
}

int default_scan_threads(const mp3s_ctx *c)
{
    if (c && c->opt[MP3S_OPT_SCAN_THREADS] > 0) return (int)c->opt[MP3S_OPT_SCAN_THREADS];
    // the frame walk takes 0.27 ms of a 0.77 ms batch: one thread, and a second one only makes the two take turns at the issue
    // (0.76 against 0.80 ms per batch); the byte-level scan of round 2 (0.75 ms per batch) wants up to three
    if (!c || c->opt[MP3S_OPT_DEVICE_PARSE]) return 1;
    const int share = host_cpus_allowed() / local_world_size();
    return std::min(3, std::max(1, share - 1));
}

std::vector<int> gpu_node_cpus(int device)
{
    std::vector<int> out;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) return out;
    for (char *p = bus; *p; p++) *p = (char)tolower(*p);
    const std::string node = read_line(std::string("/sys/bus/pci/devices/") + bus + "/numa_node");
    if (node.empty() || atoi(node.c_str()) < 0) return out;
    const std::string list = read_line("/sys/devices/system/node/node" + std::to_string(atoi(node.c_str())) + "/cpulist");
    cpu_set_t allowed;
    if (list.empty() || sched_getaffinity(0, sizeof allowed, &allowed) != 0) return out;
    // "0-63,128-191"
    std::istringstream is(list);
    std::string part;
    while (std::getline(is, part, ',')) {
        int a = 0, b = 0;
        if (sscanf(part.c_str(), "%d-%d", &a, &b) == 2) { }
        else if (sscanf(part.c_str(), "%d", &a) == 1) b = a;
        else continue;
        for (int cpu = a; cpu <= b && cpu < CPU_SETSIZE; cpu++) if (CPU_ISSET(cpu, &allowed)) out.push_back(cpu);
    }
    // a node this process may hardly use (a narrow affinity mask elsewhere) is no place to bind to
    if ((int)out.size() < 2) out.clear();
    return out;
}

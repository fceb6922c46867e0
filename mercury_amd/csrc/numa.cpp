// Host placement for the multi-GPU driver (include/mercury_gpu.h: mgpu_device_props_get / mgpu_alloc_host_near; include/mercury_pool.h): which
// NUMA node a GPU hangs off, which CPUs belong to it, and memory-policy / affinity helpers built on plain Linux system calls (no libnuma in
// the image). Eight GPUs fed from host memory ask the host for ~450 GB/s of reads: a page-locked input buffer on the other
// socket's memory makes every DMA cross the inter-socket link, so a context's staging buffers are allocated on its GPU's node and a pool's
// worker thread runs there. The caller mirrored is the reference's single-threaded RX loop (telecom_system.cc:2266-2390), which has no
// placement to speak of: one process, one sound card.
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "numa.hpp"

namespace mgpu_numa {

static std::string slurp(const std::string& path) {
    std::ifstream f(path);
    std::string s;
    if (f) std::getline(f, s);
    return s;
}

// "0-15,32-47" -> the CPU numbers
std::vector<int> parse_cpulist(const std::string& s) {
    std::vector<int> out;
    size_t i = 0;
    while (i < s.size()) {
        char* end = nullptr;
        const long a = std::strtol(s.c_str() + i, &end, 10);
        if (end == s.c_str() + i) break;
        long b = a;
        i = size_t(end - s.c_str());
        if (i < s.size() && s[i] == '-') {
            b = std::strtol(s.c_str() + i + 1, &end, 10);
            i = size_t(end - s.c_str());
        }
        for (long c = a; c <= b && c - a < 4096; ++c) out.push_back(int(c));
        if (i < s.size() && s[i] == ',') ++i;
    }
    return out;
}

int node_of_pci(const char* pci_bus_id, const char* sysfs_root) {
    if (!pci_bus_id || !*pci_bus_id) return -1;
    std::string id(pci_bus_id);
    for (auto& ch : id) if (ch >= 'A' && ch <= 'F') ch = char(ch - 'A' + 'a');       // sysfs spells the address in lower case
    const std::string s = slurp(std::string(sysfs_root ? sysfs_root : "/sys") + "/bus/pci/devices/" + id + "/numa_node");
    if (s.empty()) return -1;
    const int n = std::atoi(s.c_str());
    return n >= 0 ? n : -1;                                                          // -1: the platform reports no affinity
}

std::vector<int> cpus_of_node(int node, const char* sysfs_root) {
    if (node < 0) return {};
    return parse_cpulist(slurp(std::string(sysfs_root ? sysfs_root : "/sys") + "/devices/system/node/node" + std::to_string(node) + "/cpulist"));
}

bool bind_thread_to_node(int node) {
    const std::vector<int> cpus = cpus_of_node(node, nullptr);
    if (cpus.empty()) return false;
    cpu_set_t now, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof(now), &now) != 0) return false;
    int n = 0;
    for (int c : cpus) if (c < CPU_SETSIZE && CPU_ISSET(c, &now)) { CPU_SET(c, &want); ++n; }     // never widen what the process was given
    return n > 0 && sched_setaffinity(0, sizeof(want), &want) == 0;
}

// set_mempolicy(2) for the calling thread: MPOL_PREFERRED on `node` while the scope lives (pages first touched — and page-locked — inside
// it come from that node when it has room); afterwards the policy the thread HAD (a process started under numactl --interleave /
// --membind, or one that set its own, keeps it: the constructor reads it with get_mempolicy and the destructor puts it back)
static const unsigned long kMaskBits = 1024;                    // nodes 0..1023: the kernel's MAX_NUMNODES on x86-64 distributions
static long set_policy(int mode, const unsigned long* mask) {
#ifdef SYS_set_mempolicy
    return syscall(SYS_set_mempolicy, mode, mask, mask ? kMaskBits + 1 : 0);
#else
    (void)mode; (void)mask;
    return -1;
#endif
}
PreferNode::PreferNode(int node) : active_(false), saved_mode_(0) {
    for (unsigned long& w : saved_mask_) w = 0;
#if defined(SYS_set_mempolicy) && defined(SYS_get_mempolicy)
    if (node < 0 || node >= int(kMaskBits)) return;
    int mode = 0;
    if (syscall(SYS_get_mempolicy, &mode, saved_mask_, kMaskBits + 1, nullptr, 0) != 0) return;     // cannot restore -> do not change
    saved_mode_ = mode;
    unsigned long mask[kMaskBits / (8 * sizeof(unsigned long))] = {0};
    mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
    active_ = set_policy(1 /* MPOL_PREFERRED */, mask) == 0;
#else
    (void)node;
#endif
}
PreferNode::~PreferNode() {
    if (!active_) return;
    bool any = false;
    for (unsigned long w : saved_mask_) any = any || w != 0;
    set_policy(saved_mode_, any ? saved_mask_ : nullptr);
}

}  // namespace mgpu_numa

extern "C" {

int mgpu_host_numa_node_of_pci(const char* pci_bus_id) { return mgpu_numa::node_of_pci(pci_bus_id, std::getenv("MERCURY_SYSFS_ROOT")); }

int mgpu_host_numa_cpus(int node, int* cpus, int max) {
    const std::vector<int> v = mgpu_numa::cpus_of_node(node, std::getenv("MERCURY_SYSFS_ROOT"));
    for (int i = 0; i < int(v.size()) && i < max && cpus; ++i) cpus[i] = v[size_t(i)];
    return int(v.size());
}

}  // extern "C"

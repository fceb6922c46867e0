// NUMA placement helpers (numa.cpp)
#pragma once
#include <string>
#include <vector>

namespace mgpu_numa {
std::vector<int> parse_cpulist(const std::string& s);
int node_of_pci(const char* pci_bus_id, const char* sysfs_root);     // -1: unknown / no affinity
std::vector<int> cpus_of_node(int node, const char* sysfs_root);
bool bind_thread_to_node(int node);                                   // the calling thread; false when the node has no usable CPUs
class PreferNode {                                                    // memory policy of the calling thread while in scope
public:
    explicit PreferNode(int node);
    ~PreferNode();
    PreferNode(const PreferNode&) = delete;
    PreferNode& operator=(const PreferNode&) = delete;
    bool active() const { return active_; }
private:
    bool active_;
    int saved_mode_;
    unsigned long saved_mask_[1024 / (8 * sizeof(unsigned long)) + 1];   // the policy the thread had, put back by the destructor
};
}  // namespace mgpu_numa

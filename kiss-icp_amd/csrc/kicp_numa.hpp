// kicp_numa.hpp -- where the HOST side of a pipeline lives: the pinned staging slots a scan is copied into, and the helper
// threads that copy, belong on the NUMA node the GPU hangs off.  An 8-GPU MI355X box has two sockets (four GPUs each): a
// staging slot on the other socket makes every zero-copy read of the front kernels (12 B per point over PCIe) cross the
// socket link first, and a helper thread over there writes its share of the scan across it.
// Host-only, no HIP, no libnuma (not in the image): sysfs for the topology, the raw mbind / move_pages system calls for the
// memory.  Every function degrades to "unknown" (-1 / false / nullptr) -- containers may hide sysfs nodes or refuse the
// system calls -- and the callers then do what they did before.  Tested on the CPU by tests/test_numa_host.py.
#pragma once
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace kicp {
namespace numa {

// NUMA node of a PCI function ("0000:05:00.0", any case) from sysfs; -1: unknown or the platform reports none
inline int pci_numa_node(const char *bdf, const char *sysfs_root = "/sys") {
    if (!bdf || !*bdf) return -1;
    char low[64];
    size_t k = 0;
    for (; bdf[k] && k + 1 < sizeof low; ++k) low[k] = (char)tolower((unsigned char)bdf[k]);
    low[k] = 0;
    char path[256];
    snprintf(path, sizeof path, "%s/bus/pci/devices/%s/numa_node", sysfs_root, low);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

// "0-3,8,10-11\n" -> set; false on a malformed list (set is then unspecified)
inline bool parse_cpulist(const char *text, cpu_set_t *set) {
    CPU_ZERO(set);
    const char *p = text;
    while (*p) {
        while (*p == ',' || isspace((unsigned char)*p)) ++p;
        if (!*p) break;
        char *end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p || a < 0) return false;
        long b = a;
        p = end;
        if (*p == '-') {
            ++p;
            b = strtol(p, &end, 10);
            if (end == p || b < a) return false;
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) CPU_SET((int)c, set);
        if (*p && *p != ',' && !isspace((unsigned char)*p)) return false;
    }
    return true;
}

// the CPUs of `node` that this process may run on (its affinity mask); false: unknown node or none of them allowed
inline bool node_cpus(int node, cpu_set_t *out, const char *sysfs_root = "/sys") {
    if (node < 0) return false;
    char path[256], text[4096];
    snprintf(path, sizeof path, "%s/devices/system/node/node%d/cpulist", sysfs_root, node);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    const size_t got = fread(text, 1, sizeof text - 1, f);
    fclose(f);
    text[got] = 0;
    cpu_set_t of_node, mine;
    if (!parse_cpulist(text, &of_node)) return false;
    if (sched_getaffinity(0, sizeof mine, &mine) != 0) return false;
    CPU_AND(out, &of_node, &mine);
    return CPU_COUNT(out) > 0;
}

// restrict a thread to the node's CPUs (a no-op returning false when they are unknown)
inline bool bind_thread_to_node(pthread_t t, int node, const char *sysfs_root = "/sys") {
    cpu_set_t set;
    if (!node_cpus(node, &set, sysfs_root)) return false;
    return pthread_setaffinity_np(t, sizeof set, &set) == 0;
}

// the node a (touched) page lies on: move_pages in its query form; -1: unknown (not touched, call refused)
inline int node_of_address(const void *p) {
#ifdef SYS_move_pages
    const long page = sysconf(_SC_PAGESIZE);
    void *pages[1] = {(void *)((unsigned long)p & ~(unsigned long)(page - 1))};
    int status[1] = {-1};
    if (syscall(SYS_move_pages, 0, 1ul, pages, nullptr, status, 0) != 0) return -1;
    return status[0] >= 0 ? status[0] : -1;
#else
    (void)p;
    return -1;
#endif
}

// page-aligned anonymous memory whose pages PREFER `node` (soft: a full node falls back to its neighbours instead of failing),
// touched so that they exist before the runtime pins them.  nullptr: the node is unknown, or mmap / mbind refused.
inline void *alloc_on_node(size_t bytes, int node) {
#ifdef SYS_mbind
    if (node < 0 || node >= 1024 || bytes == 0) return nullptr;
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) return nullptr;
    unsigned long mask[1024 / (8 * sizeof(unsigned long))] = {0};
    mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
    constexpr int kPreferred = 1;  // MPOL_PREFERRED (<linux/mempolicy.h>)
    if (syscall(SYS_mbind, p, bytes, kPreferred, mask, 1024ul + 1, 0u) != 0) {
        munmap(p, bytes);
        return nullptr;
    }
    const long page = sysconf(_SC_PAGESIZE);
    for (size_t off = 0; off < bytes; off += (size_t)page) ((volatile char *)p)[off] = 0;
    return p;
#else
    (void)bytes;
    (void)node;
    return nullptr;
#endif
}
inline void free_on_node(void *p, size_t bytes) {
    if (p) munmap(p, bytes);
}

}  // namespace numa
}  // namespace kicp

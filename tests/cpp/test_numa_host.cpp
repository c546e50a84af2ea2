// test_numa_host.cpp -- host-only checks of kicp_numa.hpp (no GPU, no HIP): the cpulist grammar, the sysfs look-ups against a
// made-up tree (argv[1]: a scratch directory), thread binding, and -- where the kernel allows the calls -- that memory asked
// for on a node lies there.  Run by tests/test_numa_host.py.
#include <sys/stat.h>

#include <cstdio>
#include <string>

#include "kicp_numa.hpp"

static int g_fail = 0;
#define CHECK(cond)                                                     \
    do {                                                                \
        if (!(cond)) {                                                  \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); \
            ++g_fail;                                                   \
        }                                                               \
    } while (0)

static void put(const std::string &path, const std::string &text) {
    FILE *f = fopen(path.c_str(), "w");
    if (!f) {
        std::printf("cannot write %s\n", path.c_str());
        ++g_fail;
        return;
    }
    fputs(text.c_str(), f);
    fclose(f);
}
static void mkdirs(const std::string &path) {
    for (size_t i = 1; i <= path.size(); ++i)
        if (i == path.size() || path[i] == '/') mkdir(path.substr(0, i).c_str(), 0755);
}

int main(int argc, char **argv) {
    using namespace kicp::numa;
    if (argc < 2) return 2;
    const std::string root = argv[1];
    cpu_set_t s;
    // ---- the list grammar of /sys/devices/system/node/nodeN/cpulist --------------------------------------------------
    CHECK(parse_cpulist("0-3,8,10-11\n", &s) && CPU_COUNT(&s) == 7 && CPU_ISSET(0, &s) && CPU_ISSET(3, &s) && !CPU_ISSET(4, &s) && CPU_ISSET(8, &s) &&
          CPU_ISSET(11, &s));
    CHECK(parse_cpulist("5", &s) && CPU_COUNT(&s) == 1 && CPU_ISSET(5, &s));
    CHECK(parse_cpulist("", &s) && CPU_COUNT(&s) == 0);
    CHECK(parse_cpulist("\n", &s) && CPU_COUNT(&s) == 0);
    CHECK(parse_cpulist("0-63,128-191", &s) && CPU_COUNT(&s) == 128);
    CHECK(!parse_cpulist("3-1", &s));
    CHECK(!parse_cpulist("a-b", &s));
    CHECK(!parse_cpulist("1;2", &s));
    CHECK(parse_cpulist("100000", &s) && CPU_COUNT(&s) == 0);  // beyond CPU_SETSIZE: ignored, not an overflow
    // ---- a made-up sysfs: a GPU on node 1, whose CPUs are the first CPU this process may use (+ one it may not) --------------
    cpu_set_t mine;
    CHECK(sched_getaffinity(0, sizeof mine, &mine) == 0);
    int first = -1, foreign = -1;
    for (int c = 0; c < CPU_SETSIZE; ++c) {
        if (CPU_ISSET(c, &mine) && first < 0) first = c;
        if (!CPU_ISSET(c, &mine) && foreign < 0) foreign = c;
    }
    CHECK(first >= 0);
    mkdirs(root + "/bus/pci/devices/0000:c5:00.0");
    put(root + "/bus/pci/devices/0000:c5:00.0/numa_node", "1\n");
    mkdirs(root + "/bus/pci/devices/0000:05:00.0");
    put(root + "/bus/pci/devices/0000:05:00.0/numa_node", "-1\n");  // (what a single-node or virtualised platform says)
    mkdirs(root + "/devices/system/node/node1");
    put(root + "/devices/system/node/node1/cpulist", std::to_string(first) + (foreign >= 0 ? "," + std::to_string(foreign) : "") + "\n");
    mkdirs(root + "/devices/system/node/node2");
    put(root + "/devices/system/node/node2/cpulist", foreign >= 0 ? std::to_string(foreign) + "\n" : "\n");
    CHECK(pci_numa_node("0000:C5:00.0", root.c_str()) == 1);  // (hipDeviceGetPCIBusId prints upper-case hex, sysfs lower-case)
    CHECK(pci_numa_node("0000:c5:00.0", root.c_str()) == 1);
    CHECK(pci_numa_node("0000:05:00.0", root.c_str()) == -1);
    CHECK(pci_numa_node("0000:99:00.0", root.c_str()) == -1);
    CHECK(pci_numa_node("", root.c_str()) == -1 && pci_numa_node(nullptr, root.c_str()) == -1);
    CHECK(node_cpus(1, &s, root.c_str()) && CPU_COUNT(&s) == 1 && CPU_ISSET(first, &s));  // the foreign CPU is not offered
    CHECK(!node_cpus(2, &s, root.c_str()));                                               // nothing this process may use
    CHECK(!node_cpus(7, &s, root.c_str()) && !node_cpus(-1, &s, root.c_str()));
    // ---- binding: the calling thread ends up on exactly that CPU; an unknown node changes nothing ---------------------------
    CHECK(!bind_thread_to_node(pthread_self(), 7, root.c_str()));
    cpu_set_t now;
    CHECK(pthread_getaffinity_np(pthread_self(), sizeof now, &now) == 0 && CPU_EQUAL(&now, &mine));
    CHECK(bind_thread_to_node(pthread_self(), 1, root.c_str()));
    CHECK(pthread_getaffinity_np(pthread_self(), sizeof now, &now) == 0 && CPU_COUNT(&now) == 1 && CPU_ISSET(first, &now));
    CHECK(pthread_setaffinity_np(pthread_self(), sizeof mine, &mine) == 0);
    // ---- memory on a node of the REAL machine (node 0 exists everywhere; the calls may be refused in a container) ------------
    CHECK(alloc_on_node(1 << 20, -1) == nullptr && alloc_on_node(0, 0) == nullptr);
    void *p = alloc_on_node((size_t)1 << 20, 0);
    if (p) {
        const int where = node_of_address((char *)p + 4096 * 3);
        std::printf("alloc_on_node(1 MiB, 0): page on node %d\n", where);
        CHECK(where == 0 || where == -1);
        ((char *)p)[12345] = 7;
        free_on_node(p, (size_t)1 << 20);
    } else {
        std::printf("alloc_on_node refused here (mbind not permitted): the callers keep the runtime's own placement\n");
    }
    int on_stack = 0;
    const int sn = node_of_address(&on_stack);
    std::printf("the stack lies on node %d\n", sn);
    CHECK(sn >= -1);
    std::printf(g_fail ? "test_numa_host: %d FAILED\n" : "test_numa_host: all checks passed\n", g_fail);
    return g_fail ? 1 : 0;
}

// pinning.hpp -- optional placement of the oracle's worker threads (test infrastructure, like everything under oracle/).
// The CPU baseline of bench.py is timed on whatever box the GPU sits in, typically two sockets: with `orc_set_thread_pinning(1)`
// worker t of a parallel_search runs on the t-th logical CPU of a NUMA-ordered list (all CPUs of node 0, then node 1, ...), so a
// run on T threads uses as few memory domains as T allows and the by-thread-count figures are reproducible.
#pragma once
#include <sched.h>
#include <unistd.h>
#include <sys/syscall.h>

#include <cstdlib>

#include <atomic>
#include <cstdio>
#include <string>
#include <vector>

namespace oracle_pin {

// The index of the CPU baseline is built by ONE thread (orc_load, FlatBaseline's constructor); with the kernel's default
// first-touch policy every page of it then lives on that thread's NUMA node, and the 128-256 search threads of a two-socket
// host all pull from one node's memory controllers (round 5: the baseline got SLOWER beyond 64 threads).  While an
// InterleavedAllocations object lives, the pages this thread touches for the first time are spread round-robin over all
// nodes (set_mempolicy(MPOL_INTERLEAVE), the raw system call: no libnuma in the image); ORACLE_NO_INTERLEAVE=1 keeps the default.
struct InterleavedAllocations {
    bool on = false;
    InterleavedAllocations() {
#if defined(__linux__) && defined(__x86_64__)
        if (std::getenv("ORACLE_NO_INTERLEAVE")) return;
        int nodes = 0;
        for (; nodes < 64; ++nodes) {
            char path[96];
            std::snprintf(path, sizeof(path), "/sys/devices/system/node/node%d", nodes);
            if (::access(path, F_OK) != 0) break;
        }
        if (nodes < 2) return;
        unsigned long mask = nodes >= 64 ? ~0ul : ((1ul << nodes) - 1ul);
        on = ::syscall(238 /* SYS_set_mempolicy */, 3 /* MPOL_INTERLEAVE */, &mask, (unsigned long)(nodes + 1)) == 0;
#endif
    }
    ~InterleavedAllocations() {
#if defined(__linux__) && defined(__x86_64__)
        if (on) (void)::syscall(238, 0 /* MPOL_DEFAULT */, nullptr, 0ul);
#endif
    }
    InterleavedAllocations(const InterleavedAllocations&) = delete;
    InterleavedAllocations& operator=(const InterleavedAllocations&) = delete;
};

inline std::atomic<int>& mode() {
    static std::atomic<int> m{0};
    return m;
}

// "0-63,128-191" -> the CPUs it names
inline void parse_cpulist(const std::string& s, std::vector<int>& out) {
    size_t i = 0;
    while (i < s.size()) {
        if (s[i] < '0' || s[i] > '9') { ++i; continue; }
        int a = 0;
        while (i < s.size() && s[i] >= '0' && s[i] <= '9') a = a * 10 + (s[i++] - '0');
        int b = a;
        if (i < s.size() && s[i] == '-') {
            ++i;
            b = 0;
            while (i < s.size() && s[i] >= '0' && s[i] <= '9') b = b * 10 + (s[i++] - '0');
        }
        for (int c = a; c <= b; ++c) out.push_back(c);
    }
}

// logical CPUs this process may run on, node by node (sysfs); {} when the topology cannot be read
inline const std::vector<int>& cpu_order() {
    static const std::vector<int> order = [] {
        std::vector<int> o;
        cpu_set_t allowed;
        CPU_ZERO(&allowed);
        if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return o;
        for (int node = 0; node < 64; ++node) {
            char path[96];
            std::snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
            std::FILE* f = std::fopen(path, "r");
            if (!f) break;
            char buf[4096];
            std::string s;
            if (std::fgets(buf, sizeof(buf), f)) s = buf;
            std::fclose(f);
            std::vector<int> cpus;
            parse_cpulist(s, cpus);
            for (int c : cpus)
                if (c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) o.push_back(c);
        }
        if (o.empty())
            for (int c = 0; c < CPU_SETSIZE; ++c)
                if (CPU_ISSET(c, &allowed)) o.push_back(c);
        return o;
    }();
    return order;
}

// RAII: the calling thread runs on the CPU of worker t for the scope (pinning on), and gets its old mask back afterwards
class Pin {
public:
    explicit Pin(int t) {
        if (mode().load() == 0) return;
        const std::vector<int>& o = cpu_order();
        if (o.empty()) return;
        if (sched_getaffinity(0, sizeof(old_), &old_) != 0) return;
        cpu_set_t one;
        CPU_ZERO(&one);
        CPU_SET(o[(size_t)t % o.size()], &one);
        restore_ = sched_setaffinity(0, sizeof(one), &one) == 0;
    }
    ~Pin() { if (restore_) (void)sched_setaffinity(0, sizeof(old_), &old_); }
    Pin(const Pin&) = delete;
    Pin& operator=(const Pin&) = delete;

private:
    cpu_set_t old_;
    bool restore_ = false;
};

}  // namespace oracle_pin

// tests/mock_rccl/mock_rccl_shm.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl whose ranks are PROCESSES (or
// threads) that may all share one GPU.
//
// Why: `bench.py --gpus N` is launched by torch.distributed.run as N processes, one rank each; on a box with one GPU
// the real RCCL refuses a second rank on the same device, so bench.py's world > 1 path (SlabLayout, the unique-id
// broadcast, wv_comm_init, the chain's per-batch agreements, the max-over-ranks timing, the JSON line) could never
// execute before the driver's 8-GPU run.  tests/mock_rccl/mock_rccl.cpp covers csrc/comm.cpp's stream choreography with
// thread ranks; this one trades that fidelity for process ranks: every call is HOST-SYNCHRONOUS (the stream is drained,
// data is staged through POSIX shared memory), which NCCL semantics allow -- a call may block until the peer has made the
// matching one -- but which overlaps nothing.  Loaded by explicit path through wv_comm_use_library (no soname games with
// the real librccl that torch brings into the process).
//
// Entry points: exactly what csrc/comm.cpp resolves with dlsym.  ncclAllReduce: ncclUint64 with ncclSum or ncclMin.
//
//   hipcc -O2 -fPIC -shared tests/mock_rccl/mock_rccl_shm.cpp -o <dir>/libwvmockrccl.so -lrt
#include <fcntl.h>
#include <signal.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxRanks = 16, kMaxWords = 1024;
// how long a call waits for the peer's matching call (WV_MOCK_RCCL_TIMEOUT_S in the environment overrides)
double timeout_seconds() {
    static const double t = [] {
        const char* v = std::getenv("WV_MOCK_RCCL_TIMEOUT_S");
        return v ? std::atof(v) : 180.0;
    }();
    return t;
}

// WV_MOCK_RCCL_STALL=<rank>:<n>: that rank's n-th ncclAllReduce does nothing but put a kernel on the stream that spins (for at
// most a minute) and returns -- what a real RCCL collective with a dead peer looks like from the host: the call returns, the
// stream never drains.  For the engine's watchdog (csrc/comm.cpp, SlabComm::sync).
// WV_MOCK_RCCL_DIE=<rank>:<n>: that rank's process is killed (SIGKILL) in its n-th ncclAllReduce: a rank that dies mid-run.
__global__ void stall_kernel(volatile int* release, long long ticks) {
    const long long t0 = wall_clock64();
    while (!*release && wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(100);
}
// 0: go on, 1: stall, 2: die -- decided per ncclAllReduce of `rank`
int fate_now(int rank) {
    static const char* stall = std::getenv("WV_MOCK_RCCL_STALL");
    static const char* die = std::getenv("WV_MOCK_RCCL_DIE");
    static std::atomic<int> calls{0};
    if (!stall && !die) return 0;
    const int mine = calls.fetch_add(1) + 1;
    int r = -1, n = -1;
    if (stall && std::sscanf(stall, "%d:%d", &r, &n) == 2 && r == rank && n == mine) return 1;
    if (die && std::sscanf(die, "%d:%d", &r, &n) == 2 && r == rank && n == mine) return 2;
    return 0;
}

struct Control {
    std::atomic<int> nranks;  // 0 until the first rank joins
    std::atomic<int> joined, gone;
    std::atomic<int> arrived[2], left[2];
    unsigned long long value[2][kMaxRanks][kMaxWords];
};

struct Header {
    std::atomic<int> ready;
    unsigned long long bytes;
    char pad[48];
};

struct Comm {
    std::string id;
    Control* ctl = nullptr;
    int rank = 0, nranks = 1;
    int bank = 0;
    std::map<int, unsigned long long> sent, received;  // per peer: messages so far
};

struct Op {
    bool send;
    void* buf;
    size_t bytes;
    int peer;
    Comm* comm;
    hipStream_t stream;
};
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
std::atomic<unsigned long long> g_next_id{1};

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

size_t dtype_size(int dt) {
    switch (dt) {
        case 0: case 1: return 1;
        case 2: case 3: case 7: return 4;
        case 4: case 5: case 8: return 8;
        case 6: return 2;
        default: return 0;
    }
}

template <typename Ready>
bool wait_until(Ready ready) {
    const double t0 = now();
    int spins = 0;
    while (!ready()) {
        if (++spins > 200) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (now() - t0 > timeout_seconds()) return false;
    }
    return true;
}

void* map_segment(const std::string& name, size_t bytes, bool create) {
    int fd = -1;
    if (create) {
        fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) return nullptr;
    } else {
        const bool ok = wait_until([&] {
            fd = shm_open(name.c_str(), O_RDWR, 0600);
            if (fd < 0) return false;
            struct stat st;
            if (fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes) return true;  // created AND sized
            close(fd);
            fd = -1;
            return false;
        });
        if (!ok) return nullptr;
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    return p == MAP_FAILED ? nullptr : p;
}

std::string message_name(const Comm& c, int from, int to, unsigned long long seq) {
    return "/" + c.id + "-" + std::to_string(from) + "-" + std::to_string(to) + "-" + std::to_string(seq);
}

int flush() {
    std::vector<Op> ops;
    ops.swap(t_ops);
    for (const Op& o : ops)  // what is sent is final, what is overwritten is no longer read
        if (hipStreamSynchronize(o.stream) != hipSuccess) return 1;
    for (const Op& o : ops) {  // 1. every send is posted before anything waits
        if (!o.send) continue;
        Comm& c = *o.comm;
        const std::string name = message_name(c, c.rank, o.peer, c.sent[o.peer]++);
        char* seg = static_cast<char*>(map_segment(name, sizeof(Header) + o.bytes, true));
        if (!seg) return 3;
        Header* h = reinterpret_cast<Header*>(seg);
        h->bytes = o.bytes;
        if (o.bytes && hipMemcpy(seg + sizeof(Header), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        h->ready.store(1, std::memory_order_release);
        munmap(seg, sizeof(Header) + o.bytes);
    }
    for (const Op& o : ops) {  // 2. receives, in issue order per peer
        if (o.send) continue;
        Comm& c = *o.comm;
        const std::string name = message_name(c, o.peer, c.rank, c.received[o.peer]++);
        char* seg = static_cast<char*>(map_segment(name, sizeof(Header) + o.bytes, false));
        if (!seg) return 3;
        Header* h = reinterpret_cast<Header*>(seg);
        if (!wait_until([&] { return h->ready.load(std::memory_order_acquire) == 1; })) return 3;
        int rc = 0;
        if (h->bytes != o.bytes) rc = 2;
        if (!rc && o.bytes && hipMemcpy(o.buf, seg + sizeof(Header), o.bytes, hipMemcpyHostToDevice) != hipSuccess) rc = 1;
        munmap(seg, sizeof(Header) + o.bytes);
        shm_unlink(name.c_str());
        if (rc) return rc;
    }
    return 0;
}

}  // namespace

extern "C" {

int ncclGetUniqueId(char* id128) {
    std::memset(id128, 0, 128);
    const std::string id = "wvmockrccl-" + std::to_string((long long)getpid()) + "-" +
                           std::to_string((long long)(now() * 1e6)) + "-" + std::to_string(g_next_id++);
    std::memcpy(id128, id.c_str(), id.size());
    Control* ctl = static_cast<Control*>(map_segment("/" + id, sizeof(Control), true));
    if (!ctl) return 3;
    munmap(ctl, sizeof(Control));  // (zero-filled by ftruncate: nranks 0, nobody joined)
    return 0;
}

struct IdByValue {
    char internal[128];
};

int ncclCommInitRank(void** comm, int nranks, IdByValue id, int rank) {
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return 4;
    Comm* c = new Comm;
    c->id.assign(id.internal, strnlen(id.internal, 128));
    c->rank = rank;
    c->nranks = nranks;
    c->ctl = static_cast<Control*>(map_segment("/" + c->id, sizeof(Control), false));
    if (!c->ctl) {
        delete c;
        return 3;
    }
    int expected = 0;
    if (!c->ctl->nranks.compare_exchange_strong(expected, nranks) && expected != nranks) return 4;
    c->ctl->joined.fetch_add(1);
    if (!wait_until([&] { return c->ctl->joined.load() >= nranks; })) return 3;  // collective, like the real one
    *comm = c;
    return 0;
}

int ncclCommDestroy(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return 0;
    if (c->ctl->gone.fetch_add(1) + 1 == c->nranks) shm_unlink(("/" + c->id).c_str());
    munmap(c->ctl, sizeof(Control));
    delete c;
    return 0;
}

int ncclCommAbort(void* comm) {  // what the engine's watchdog calls: nothing of this stand-in's runs on the device
    Comm* c = static_cast<Comm*>(comm);
    if (c) {
        munmap(c->ctl, sizeof(Control));
        delete c;
    }
    return 0;
}

int ncclGroupStart(void) {
    ++t_depth;
    return 0;
}

int ncclGroupEnd(void) {
    if (t_depth <= 0) return 5;
    if (--t_depth == 0) return flush();
    return 0;
}

int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
    t_ops.push_back(Op{true, const_cast<void*>(buf), count * dtype_size(dtype), peer, static_cast<Comm*>(comm), stream});
    return t_depth ? 0 : flush();
}

int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
    t_ops.push_back(Op{false, buf, count * dtype_size(dtype), peer, static_cast<Comm*>(comm), stream});
    return t_depth ? 0 : flush();
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) {
    if (dtype != 5 || (op != 0 && op != 3) || count > (size_t)kMaxWords) return 4;  // ncclUint64; ncclSum / ncclMin
    Comm* c = static_cast<Comm*>(comm);
    const int fate = fate_now(c->rank);
    if (fate == 1) {
        int* release = nullptr;
        if (hipMalloc((void**)&release, sizeof(int)) != hipSuccess || hipMemset(release, 0, sizeof(int)) != hipSuccess) return 1;
        hipLaunchKernelGGL(stall_kernel, dim3(1), dim3(1), 0, stream, release, 60ll * 100000000ll);  // (wall_clock64: 100 MHz)
        return 0;
    }
    if (fate == 2) {
        std::fflush(nullptr);
        kill(getpid(), SIGKILL);
    }
    Control& ctl = *c->ctl;
    const int b = c->bank;
    c->bank ^= 1;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    if (hipMemcpy(ctl.value[b][c->rank], send, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    ctl.arrived[b].fetch_add(1, std::memory_order_acq_rel);
    if (!wait_until([&] { return ctl.arrived[b].load(std::memory_order_acquire) >= c->nranks; })) return 3;
    std::vector<unsigned long long> total(count);
    for (size_t i = 0; i < count; ++i) {
        unsigned long long v = ctl.value[b][0][i];
        for (int r = 1; r < c->nranks; ++r) {
            const unsigned long long w = ctl.value[b][r][i];
            v = op == 0 ? v + w : (w < v ? w : v);
        }
        total[i] = v;
    }
    if (ctl.left[b].fetch_add(1, std::memory_order_acq_rel) + 1 == c->nranks) {  // last one out resets the bank
        ctl.left[b].store(0);
        ctl.arrived[b].store(0, std::memory_order_release);
    }
    if (hipMemcpy(recv, total.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) return 1;
    return 0;
}

const char* ncclGetErrorString(int code) {
    switch (code) {
        case 0: return "success";
        case 1: return "mock rccl (shm): a HIP call failed";
        case 2: return "mock rccl (shm): mismatched sizes between the ranks";
        case 3: return "mock rccl (shm): a peer did not show up in time, or shared memory failed";
        case 4: return "mock rccl (shm): invalid argument";
        case 5: return "mock rccl (shm): ncclGroupEnd without ncclGroupStart";
        default: return "mock rccl (shm): error";
    }
}

}  // extern "C"

// tests/mock_rccl/mock_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl whose "ranks" are threads of one
// process sharing one GPU.
//
// Why: csrc/comm.cpp's multi-rank branch (open chain: rank-1 / rank+1 peers, has_lo / has_hi, the flag all-reduce and
// the per-batch "passes or single steps" word over nranks > 1) needs a second rank to run at all, and the boxes this
// is developed on have one GPU, on which the real RCCL refuses a second rank.  Built as librccl.so.1 into a scratch
// directory that the worker process puts first on LD_LIBRARY_PATH (tests/test_gpu_rccl_chain.py), it gives
// comm.cpp exactly the nine entry points it resolves with dlsym, with RCCL's stream semantics:
//   ncclSend / ncclRecv (inside ncclGroupStart / End): the receive is ordered after the sender's stream at the
//     time of the send, the sender's stream does not run on before the data has been taken; sends and receives
//     between a pair of ranks match in issue order; a rank may send to itself;
//   ncclAllReduce(ncclUint64, ncclSum or ncclMin, in place or not): every rank's stream sees the result.
// Calls block on the host until the peer has made the matching call -- as NCCL may.
//
//   hipcc -O2 -fPIC -shared tests/mock_rccl/mock_rccl.cpp -o <dir>/librccl.so.1
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Post {  // one ncclSend waiting for its ncclRecv
    const void* ptr = nullptr;
    size_t bytes = 0;
    hipEvent_t ready = nullptr;   // sender's stream at the time of the send
    hipEvent_t copied = nullptr;  // receiver's stream after the copy
    bool taken = false;
};

struct Group {
    int nranks = 0;
    int joined = 0, left = 0;
    std::mutex m;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<std::shared_ptr<Post>>> mail;  // (from, to) -> posts in issue order
    // all-reduce rendezvous
    std::vector<unsigned long long> sum;
    int contributed = 0, collected = 0;
    unsigned long long generation = 0;
};

struct Comm {
    std::shared_ptr<Group> g;
    int rank = 0;
};

std::mutex g_registry_mutex;
std::map<std::string, std::shared_ptr<Group>> g_registry;
unsigned long long g_next_id = 1;

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

size_t dtype_size(int dt) {
    switch (dt) {
        case 0: case 1: return 1;
        case 2: case 3: case 7: return 4;
        case 4: case 5: case 8: return 8;
        case 6: return 2;
        default: return 0;
    }
}

int flush() {
    std::vector<Op> ops;
    ops.swap(t_ops);
    std::vector<std::shared_ptr<Post>> mine;
    for (const Op& o : ops) {  // 1. every send is posted before anything waits
        if (!o.send) continue;
        auto p = std::make_shared<Post>();
        p->ptr = o.buf;
        p->bytes = o.bytes;
        if (hipEventCreateWithFlags(&p->ready, hipEventDisableTiming) != hipSuccess) return 1;
        if (hipEventRecord(p->ready, o.stream) != hipSuccess) return 1;
        {
            std::lock_guard<std::mutex> lk(o.comm->g->m);
            o.comm->g->mail[{o.comm->rank, o.peer}].push_back(p);
        }
        o.comm->g->cv.notify_all();
        mine.push_back(p);
    }
    for (const Op& o : ops) {  // 2. receives, in issue order per peer
        if (o.send) continue;
        Group& g = *o.comm->g;
        std::shared_ptr<Post> p;
        {
            std::unique_lock<std::mutex> lk(g.m);
            auto& q = g.mail[{o.peer, o.comm->rank}];
            g.cv.wait(lk, [&] { return !q.empty(); });
            p = q.front();
            q.pop_front();
        }
        if (p->bytes != o.bytes) return 2;  // mismatched send / recv sizes: a bug in the caller
        if (hipStreamWaitEvent(o.stream, p->ready, 0) != hipSuccess) return 1;
        if (hipMemcpyAsync(o.buf, p->ptr, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess) return 1;
        hipEvent_t done;
        if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) return 1;
        if (hipEventRecord(done, o.stream) != hipSuccess) return 1;
        {
            std::lock_guard<std::mutex> lk(g.m);
            p->copied = done;
            p->taken = true;
        }
        g.cv.notify_all();
    }
    size_t k = 0;
    for (const Op& o : ops) {  // 3. a send buffer is free again once the peer's copy has run
        if (!o.send) continue;
        std::shared_ptr<Post> p = mine[k++];
        Group& g = *o.comm->g;
        {
            std::unique_lock<std::mutex> lk(g.m);
            g.cv.wait(lk, [&] { return p->taken; });
        }
        if (hipStreamWaitEvent(o.stream, p->copied, 0) != hipSuccess) return 1;
    }
    return 0;
}

}  // namespace

extern "C" {

int ncclGetUniqueId(char* id128) {
    std::lock_guard<std::mutex> lk(g_registry_mutex);
    std::memset(id128, 0, 128);
    const std::string s = "mock-rccl-" + std::to_string(g_next_id++);
    std::memcpy(id128, s.c_str(), s.size());
    return 0;
}

struct IdByValue {
    char internal[128];
};

int ncclCommInitRank(void** comm, int nranks, IdByValue id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return 4;  // ncclInvalidArgument
    std::shared_ptr<Group> g;
    {
        std::lock_guard<std::mutex> lk(g_registry_mutex);
        const std::string key(id.internal, strnlen(id.internal, 128));
        auto& slot = g_registry[key];
        if (!slot) {
            slot = std::make_shared<Group>();
            slot->nranks = nranks;
        }
        g = slot;
    }
    if (g->nranks != nranks) return 4;
    {
        std::unique_lock<std::mutex> lk(g->m);
        ++g->joined;
        g->cv.notify_all();
        g->cv.wait(lk, [&] { return g->joined >= g->nranks; });  // collective, like the real one
    }
    Comm* c = new Comm;
    c->g = g;
    c->rank = rank;
    *comm = c;
    return 0;
}

int ncclCommDestroy(void* comm) {
    delete static_cast<Comm*>(comm);
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
    if (dtype != 5 || (op != 0 && op != 3)) return 4;  // only what comm.cpp asks for: ncclUint64 with ncclSum or ncclMin
    Comm* c = static_cast<Comm*>(comm);
    Group& g = *c->g;
    std::vector<unsigned long long> mine(count);
    if (hipMemcpyAsync(mine.data(), send, count * 8, hipMemcpyDeviceToHost, stream) != hipSuccess) return 1;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    std::vector<unsigned long long> total;
    {
        std::unique_lock<std::mutex> lk(g.m);
        const unsigned long long gen = g.generation;
        if (g.contributed == 0) g.sum.assign(count, op == 0 ? 0ull : ~0ull);
        if (g.sum.size() != count) return 2;
        for (size_t i = 0; i < count; ++i) g.sum[i] = op == 0 ? g.sum[i] + mine[i] : std::min(g.sum[i], mine[i]);
        if (++g.contributed == g.nranks) g.cv.notify_all();
        g.cv.wait(lk, [&] { return g.contributed == g.nranks || g.generation != gen; });
        total = g.sum;
        if (++g.collected == g.nranks) {  // last one out resets the rendezvous
            g.contributed = 0;
            g.collected = 0;
            ++g.generation;
            g.cv.notify_all();
        } else {
            g.cv.wait(lk, [&] { return g.generation != gen; });
        }
    }
    if (hipMemcpyAsync(recv, total.data(), count * 8, hipMemcpyHostToDevice, stream) != hipSuccess) return 1;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    return 0;
}

const char* ncclGetErrorString(int code) {
    switch (code) {
        case 0: return "success";
        case 1: return "mock rccl: a HIP call failed";
        case 2: return "mock rccl: mismatched sizes between the ranks";
        case 4: return "mock rccl: invalid argument";
        case 5: return "mock rccl: ncclGroupEnd without ncclGroupStart";
        default: return "mock rccl: error";
    }
}

}  // extern "C"

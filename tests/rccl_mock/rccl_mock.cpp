// rccl_mock — a stand-in for librccl.so, TEST INFRASTRUCTURE ONLY (loaded through GPMI_RCCL_LIB by tests/test_gpu_multi_rccl.py).
//
// abstractgps.jl_amd/csrc/multi.hip binds seven RCCL symbols (ncclCommInitAll, ncclCommDestroy, ncclGroupStart, ncclGroupEnd,
// ncclSend, ncclRecv, ncclGetErrorString).  The real library refuses several ranks on one device, and the CI box has one
// GPU — so the library's RCCL transport (staging images, grouped send/recv posting order, the L_kk image hand-over, the
// backward-sweep transfers) could never execute.  This file implements the same seven entry points with RCCL's
// point-to-point semantics — the i-th send of rank a to rank b matches the i-th receive of b from a, element counts must agree,
// both calls complete in stream order on their own streams — by a host rendezvous and hipMemcpyAsync:
//   send  : an event is recorded on the sender's stream (buffer ready) and the transfer is posted to the pair's queue
//   recv  : waits (host, bounded) for the matching post, makes its stream wait for the sender's event, copies, records "done"
//   sender: waits (host, bounded) until the receiver has issued the copy, then makes its stream wait for "done"
// Calls of one group are processed at ncclGroupEnd: all sends are posted when they are called, receives are served next,
// send completions last — so two ranks that send to and receive from each other in one group cannot block each other.
// Unmatched or mismatching operations are errors (with a text), never hangs: every host wait is bounded by
// RCCL_MOCK_TIMEOUT_S (default 60).  rcclMockStats reports the numbers of sends / receives served and of posts never received.
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Xfer {
    const void* sbuf = nullptr;
    size_t bytes = 0;
    hipEvent_t ready = nullptr, done = nullptr;
    int state = 0;  // 0 posted, 1 copy issued by the receiver, 2 failed
};

struct World {
    int n = 0;
    int refs = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<Xfer*>> posted;  // (src, dst) -> sends not yet matched by a receive
    std::vector<Xfer*> all;
};

struct Comm {
    World* w;
    int rank, device;
};

struct Op {
    int kind;  // 0 send (already posted), 1 recv
    void* buf;
    size_t bytes;
    int peer;
    Comm* comm;
    hipStream_t stream;
    Xfer* x;
};

thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
thread_local std::string t_err;
std::mutex g_mu;
long g_sends = 0, g_recvs = 0;
std::vector<World*> g_worlds;

double timeout_s() {
    const char* e = getenv("RCCL_MOCK_TIMEOUT_S");
    return e ? atof(e) : 60.0;
}
size_t dtype_bytes(int dt) {
    static const size_t sz[] = {1, 1, 4, 4, 8, 8, 2, 4, 8, 2};
    return dt >= 0 && dt < 10 ? sz[dt] : 0;
}
int fail(int code, const std::string& msg) {
    t_err = msg;
    fprintf(stderr, "[rccl_mock] %s\n", msg.c_str());
    return code;
}

int post_send(const void* buf, size_t bytes, int peer, Comm* c, hipStream_t s, Xfer** out) {
    Xfer* x = new Xfer();
    x->sbuf = buf;
    x->bytes = bytes;
    if (hipEventCreateWithFlags(&x->ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&x->done, hipEventDisableTiming) != hipSuccess ||
        hipEventRecord(x->ready, s) != hipSuccess)
        return fail(1, "hip event error while posting a send");
    {
        std::lock_guard<std::mutex> l(c->w->mu);
        c->w->posted[{c->rank, peer}].push_back(x);
        c->w->all.push_back(x);
    }
    c->w->cv.notify_all();
    *out = x;
    return 0;
}

int serve_recv(const Op& o) {
    World* w = o.comm->w;
    Xfer* x = nullptr;
    {
        std::unique_lock<std::mutex> l(w->mu);
        auto& q = w->posted[{o.peer, o.comm->rank}];
        if (!w->cv.wait_for(l, std::chrono::duration<double>(timeout_s()), [&] { return !q.empty(); }))
            return fail(2, "rank " + std::to_string(o.comm->rank) + ": receive from rank " + std::to_string(o.peer) + " was never matched by a send");
        x = q.front();
        q.pop_front();
    }
    int rc = 0;
    if (x->bytes != o.bytes) {
        rc = fail(4, "rank " + std::to_string(o.comm->rank) + " receives " + std::to_string(o.bytes) + " bytes from rank " + std::to_string(o.peer) +
                         " whose matching send has " + std::to_string(x->bytes));
    } else if (hipStreamWaitEvent(o.stream, x->ready, 0) != hipSuccess ||
               hipMemcpyAsync(o.buf, x->sbuf, o.bytes, hipMemcpyDefault, o.stream) != hipSuccess || hipEventRecord(x->done, o.stream) != hipSuccess) {
        rc = fail(1, "hip error while serving a receive");
    }
    {
        std::lock_guard<std::mutex> l(w->mu);
        x->state = rc == 0 ? 1 : 2;
    }
    w->cv.notify_all();
    if (rc == 0) {
        std::lock_guard<std::mutex> l(g_mu);
        ++g_recvs;
    }
    return rc;
}

int finish_send(const Op& o) {
    World* w = o.comm->w;
    {
        std::unique_lock<std::mutex> l(w->mu);
        if (!w->cv.wait_for(l, std::chrono::duration<double>(timeout_s()), [&] { return o.x->state != 0; }))
            return fail(2, "rank " + std::to_string(o.comm->rank) + ": send to rank " + std::to_string(o.peer) + " was never matched by a receive");
        if (o.x->state == 2) return fail(4, "rank " + std::to_string(o.comm->rank) + ": the receive matching a send to rank " + std::to_string(o.peer) + " failed");
    }
    if (hipStreamWaitEvent(o.stream, o.x->done, 0) != hipSuccess) return fail(1, "hip error while finishing a send");
    std::lock_guard<std::mutex> l(g_mu);
    ++g_sends;
    return 0;
}

int run_ops(std::vector<Op>& ops) {
    int rc = 0;
    for (auto& o : ops)
        if (o.kind == 1 && rc == 0) rc = serve_recv(o);
    for (auto& o : ops)
        if (o.kind == 0 && rc == 0) rc = finish_send(o);
    ops.clear();
    return rc;
}

}  // namespace

extern "C" {

int ncclCommInitAll(void** comms, int ndev, const int* devlist) {
    if (!comms || ndev < 1) return fail(4, "ncclCommInitAll: bad arguments");
    World* w = new World();
    w->n = ndev;
    w->refs = ndev;
    for (int i = 0; i < ndev; ++i) comms[i] = new Comm{w, i, devlist ? devlist[i] : i};  // duplicate devices are fine here
    std::lock_guard<std::mutex> l(g_mu);
    g_worlds.push_back(w);
    return 0;
}

int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c) return 0;
    World* w = c->w;
    delete c;
    bool last;
    {
        std::lock_guard<std::mutex> l(w->mu);
        last = --w->refs == 0;
    }
    if (last) {
        for (Xfer* x : w->all) {
            if (x->ready) (void)hipEventDestroy(x->ready);
            if (x->done) (void)hipEventDestroy(x->done);
            delete x;
        }
        std::lock_guard<std::mutex> l(g_mu);
        for (auto& p : g_worlds)
            if (p == w) p = nullptr;
        delete w;
    }
    return 0;
}

int ncclGroupStart() {
    ++t_depth;
    return 0;
}

int ncclGroupEnd() {
    if (t_depth <= 0) return fail(5, "ncclGroupEnd without ncclGroupStart");
    if (--t_depth > 0) return 0;
    return run_ops(t_ops);
}

int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    const size_t es = dtype_bytes(dtype);
    if (!c || !es || peer < 0 || peer >= c->w->n) return fail(4, "ncclSend: bad arguments");  // a self pair inside one group is legal (served as a local copy), as in RCCL
    Op o{0, nullptr, count * es, peer, c, stream, nullptr};
    const int rc = post_send(buf, count * es, peer, c, stream, &o.x);
    if (rc != 0) return rc;
    t_ops.push_back(o);
    return t_depth > 0 ? 0 : run_ops(t_ops);
}

int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    const size_t es = dtype_bytes(dtype);
    if (!c || !es || peer < 0 || peer >= c->w->n) return fail(4, "ncclRecv: bad arguments");
    t_ops.push_back(Op{1, buf, count * es, peer, c, stream, nullptr});
    return t_depth > 0 ? 0 : run_ops(t_ops);
}

const char* ncclGetErrorString(int code) {
    static thread_local std::string s;
    static const char* names[] = {"success", "unhandled hip error", "system error (an operation was never matched)", "internal error",
                                  "invalid argument (mismatching transfer)", "invalid usage"};
    s = std::string(code >= 0 && code < 6 ? names[code] : "unknown error") + (t_err.empty() ? "" : " — " + t_err);
    return s.c_str();
}

// test hook: transfers served so far, and sends of live communicators that no receive has taken
int rcclMockStats(long* sends, long* recvs, long* unmatched) {
    std::lock_guard<std::mutex> l(g_mu);
    if (sends) *sends = g_sends;
    if (recvs) *recvs = g_recvs;
    long u = 0;
    for (World* w : g_worlds)
        if (w) {
            std::lock_guard<std::mutex> l2(w->mu);
            for (auto& kv : w->posted) u += (long)kv.second.size();
        }
    if (unmatched) *unmatched = u;
    return 0;
}
}

// multi.hip — multi-device 2D block-cyclic driver of the logpdf + posterior pair, INSIDE libgpmi355.
//
// The reference has no distributed path; its caller is ONE Julia process calling posterior(fx, y)
// (src/exact_gpr_posterior.jl:29-35) / logpdf(fx, y) (src/finite_gp_projection.jl:306-311).  So the N devices of a node are
// driven from that one process: gp_ctx_create_multi builds R = P·Q rank contexts (one device each, or several "virtual" ranks
// sharing a device for tests), and gp_posterior_fit / gp_logpdf on such a ctx run the SPMD schedule below with ONE HOST
// THREAD PER RANK.  Every numeric step is a call into the single-device engine (engine.hpp: eng_*), i.e. the same HIP
// kernels as the 1-GPU path under the block-cyclic predicate (GridMap).
//
// Layout: K + Σy in NB×NB blocks, block (i, j) on rank (i mod P, j mod Q), lower blocks only; the y − m rows ride along as an
// extra block row on process row 0 (forward substitution inside the factorisation).
//
// Schedule per block column k (look-ahead depth d, default 2; three streams per rank: main / panel / comm):
//   panel stream : columns k+1..k+d are kept out of the bulk update; column k+1 gets panel k's update FIRST, is factored
//                  (diagonal owner: Cholesky of the NB×NB block -> L_kk to the P−1 column peers; all owners X ← X L_kk⁻ᵀ)
//                  and published; columns k+2..k+d get panel k's update behind it
//   comm stream  : exchange(k+1) — every rank fetches exactly the panel blocks it consumes: its own process row's piece as
//                  the A operand (rows in local row order) and the blocks of its own process column as the B operand (rows in
//                  local COLUMN order — laid out in consumer order, no gather copy).  Volume per rank ≈ panel·(1/P + 1/Q)
//   main stream  : bulk(k) — ONE MFMA GEMM over all local columns right of the look-ahead window
// Transport ("comm" parameter): 1 = RCCL grouped ncclSend/ncclRecv between distinct peers (every pair has its own xGMI link:
// the transfers of one step run on several links at once — a ring broadcast would be bound by one), 2 = peer copies pulled by
// the consumer (hipMemcpy2DAsync over xGMI), 0 = auto.  GPMI_RCCL_LIB names the library to dlopen (tests: tests/rccl_mock,
// a host-rendezvous stand-in that lets the RCCL branch run with virtual ranks on ONE device).
//
// Every stream operation, event record and event wait of a rank goes through RankRun (op / rec / wait / publish / await):
//   * GPMI_TRACE_SCHEDULE=<file> (or gp_multi_schedule_trace, which runs the SAME fit_rank control flow without a device)
//     writes them as JSON lines with their block footprints — tools/multi_schedule_check.py checks happens-before on that;
//   * "multi_check" (ctx parameter): 1 = every event record is preceded by a marker kernel and every event wait followed by a
//     checker kernel on the waiting stream (a waiter that runs before the recorded point is logged), 2 = every update GEMM
//     first compares its operand buffers with the owners' final panel blocks, 4 = operand buffers poisoned with NaN instead
//     of zeros.  Findings fail the fit with status −1990 and the list in gp_last_error().
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "engine.hpp"

using namespace gpmi;

// ------------------------------------------------------------------------------------------------
// diagnostic kernels ("multi_check")
// ------------------------------------------------------------------------------------------------
__global__ void mk_mark_kernel(int* flag, int seq) { __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// log layout: [0] number of findings, then 4 ints per finding (code, a, b, c), at most 60 kept
__device__ inline void mk_log(int* log, int code, int a, int b, int c) {
    const int i = atomicAdd(log, 1);
    if (i < 60) {
        log[4 + 4 * i] = code;
        log[5 + 4 * i] = a;
        log[6 + 4 * i] = b;
        log[7 + 4 * i] = c;
    }
}
__global__ void mk_check_kernel(const int* flag, int seq, int* log, int owner, int id, int stream) {
    const int v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (v != seq) mk_log(log, 1, owner, id, stream * 65536 + (v & 0xffff));
}
// bitwise comparison of two rows×cols blocks; one finding per launch at most
__global__ void mk_cmp_kernel(const double* a, long lda, const double* b, long ldb, long rows, long cols, int* log, int code, int k, int blk) {
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    const long total = rows * cols;
    int first = -1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols, c = i - r * cols;
        const unsigned long long va = ((const unsigned long long*)a)[r * lda + c], vb = ((const unsigned long long*)b)[r * ldb + c];
        if (va != vb && first < 0) first = (int)i;
    }
    if (first >= 0) atomicAdd(&bad, 1);
    __syncthreads();
    if (threadIdx.x == 0 && bad) mk_log(log, code, k, blk, bad);
}

// dst[r][c] += src[r][c] for a rows×cols block (row-major, own leading dimensions)
__global__ void mk_addmat_kernel(double* dst, long ldd, const double* src, long lds, long rows, long cols) {
    const long total = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols, c = i - r * cols;
        dst[r * ldd + c] += src[r * lds + c];
    }
}

// out[s][row] = Σ_{c < len(row)} A[row][c] · v[s][c] over the local columns of a block-cyclic piece of the LOWER factor: local row
// `row` of global block gi = (row / NB)·P + p reaches over the ncb local block columns left of the diagonal and — on the rank that
// owns the diagonal block — over its lower triangle (the upper triangle of a stored diagonal block is not part of the factor).
// One workgroup per (local row, vector).
__global__ __launch_bounds__(256) void mk_rowdot_kernel(const double* __restrict__ A, long ld, long NB, int P, int p, int Q, int q,
                                                         const double* __restrict__ v, long ldv, double* __restrict__ out, long ldo) {
    const long row = blockIdx.x, s = blockIdx.y;
    const long gi = (row / NB) * P + p;
    const long ncb = gi - 1 >= q ? (gi - 1 - q) / Q + 1 : 0;
    const long len = ncb * NB + ((gi % Q == q) ? (row % NB) + 1 : 0);
    const double* a = A + row * ld;
    const double* x = v + s * ldv;
    double acc = 0;
    for (long c = threadIdx.x; c < len; c += 256) acc = fma(a[c], x[c], acc);
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[s * ldo + row] = red[0];
}

namespace {

// ------------------------------------------------------------------------------------------------
// RCCL, loaded lazily (the single-device path never touches it)
// ------------------------------------------------------------------------------------------------
typedef void* ncclComm_t_;
struct Rccl {
    void* h = nullptr;
    std::string name;
    int (*CommInitAll)(ncclComm_t_*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t_) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string err;
    long live = 0;  // communicators created through the CURRENT library and not yet destroyed (under g_rccl_mu)
    bool load() {
        const char* over = getenv("GPMI_RCCL_LIB");  // tests: a stand-in library (tests/rccl_mock)
        const std::string want = over ? over : "";
        if (h && want == name) return ok;
        if (h && live > 0) {
            // a different library is asked for while communicators of the loaded one are alive: their ncclCommDestroy must come
            // from the library that created them, so the switch is refused (this context uses copies) and the pointers stay
            err = "a different RCCL library (" + (name.empty() ? std::string("librccl") : name) + ") has live communicators";
            return false;
        }
        if (h && h != (void*)1) (void)dlclose(h);
        ok = false;
        err.clear();
        name = want;
        h = nullptr;
        if (over) {
            h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
        } else {
            const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (const char* n : names) {
                h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
                if (h) break;
            }
        }
        if (!h) {
            const char* de = dlerror();
            err = std::string("dlopen(") + (over ? over : "librccl") + "): " + (de ? de : "?");
            h = (void*)1;
            return false;
        }
#define GPMI_SYM(field, sym)                                  \
    field = (decltype(field))dlsym(h, sym);                   \
    if (!field) {                                             \
        err = std::string("librccl lacks ") + sym;            \
        return false;                                         \
    }
        GPMI_SYM(CommInitAll, "ncclCommInitAll");
        GPMI_SYM(CommDestroy, "ncclCommDestroy");
        GPMI_SYM(GroupStart, "ncclGroupStart");
        GPMI_SYM(GroupEnd, "ncclGroupEnd");
        GPMI_SYM(Send, "ncclSend");
        GPMI_SYM(Recv, "ncclRecv");
        GPMI_SYM(GetErrorString, "ncclGetErrorString");
#undef GPMI_SYM
        ok = true;
        return true;
    }
};
constexpr int NCCL_FLOAT64 = 8;  // ncclDataType_t::ncclFloat64 (rccl.h)
Rccl g_rccl;
std::mutex g_rccl_mu;

// ------------------------------------------------------------------------------------------------
// events.  Ev: an event of rank `owner` with index `id` into that rank's marker-flag array (trace name "r<owner>e<id>").
// XEvent (cross-thread): the producer rank records it on one of its streams and publishes the fit sequence number; a consumer
// (another rank's host thread) waits for the publication on the HOST, then makes ITS stream wait for the event on the device.
// ------------------------------------------------------------------------------------------------
struct Ev {
    hipEvent_t ev = nullptr;
    int owner = -1, id = -1;
    const char* tag = "";  // what the event stands for (trace only)
    long k = 0;
};
struct XEvent : Ev {
    std::atomic<long> gen{0};
};

constexpr long RHS_ROWS = 128;
enum { SM = 0, SP = 1, SC = 2, SD = 3 };  // main / panel / comm / diagonal chain ("multi_chain_cus"; otherwise the chain runs on the panel stream)
const char* const SNAME[4] = {"sm", "sp", "sc", "sd"};
constexpr int NXKIND = 7;
const char* const XTAG[NXKIND] = {"ready", "lkk", "accr", "alr", "sx", "sa", "bar"};

// block footprint of an operation (block units; the checker expands it):
//   A   : rank, local block rows [a0, a1) (fl & 2: plus the RHS block row; fl & 4: the RHS block row only), local block columns
//         [b0, b1), fl & 1: only blocks on/below the global diagonal
//   Ab / St : rank, slot a0, local block rows [b0, b1) (fl & 2: plus the RHS block row)
//   Bb  : rank, slot a0, local block columns [b0, b1)
//   Lkk : rank            acc : rank, local block columns [b0, b1)       alb : rank, global block b0       tmp : rank, slot b0
struct Fp {
    const char* buf;
    int rank;
    long a0, a1, b0, b1;
    int fl;
};

struct Trace {
    FILE* f = nullptr;
    std::mutex mu;
    void line(const std::string& s) {
        std::lock_guard<std::mutex> l(mu);
        if (f) {
            fputs(s.c_str(), f);
            fputc('\n', f);
        }
    }
    static std::string fps(std::initializer_list<Fp> v) {
        std::string s = "[";
        bool first = true;
        for (const Fp& p : v) {
            if (!first) s += ",";
            first = false;
            char b[160];
            snprintf(b, sizeof b, "[\"%s\",%d,%ld,%ld,%ld,%ld,%d]", p.buf, p.rank, p.a0, p.a1, p.b0, p.b1, p.fl);
            s += b;
        }
        return s + "]";
    }
};

}  // namespace

struct MRank {
    int r = 0, p = 0, q = 0, device = 0;
    gp_ctx* c = nullptr;       // rank context: main stream c->sm, panel stream c->sp
    hipStream_t sc = nullptr;  // comm stream
    // "multi_chain_cus" = r > 0: the diagonal block's chain on a stream masked to CUs [0, r) (r/8 of every XCD), main and panel work on streams masked
    // to the other CUs (hipExtStreamCreateWithCUMask; masked streams carry no priority)
    hipStream_t sd = nullptr, sm_m = nullptr, sp_m = nullptr;
    ncclComm_t_ comm = nullptr;
    std::vector<XEvent> ready, lkk, accr, alr;  // per block column k (see fit_rank)
    std::vector<XEvent> sx, sa;                  // forward solve on the distributed factor: X_k published / accumulators after step k
    std::vector<XEvent> bar;                     // [0]: "everything this rank issued so far" — the barrier between the sweeps of a distributed solve
    std::vector<hipEvent_t> own;                 // own-thread events (arrived / bulk_done / la_done), pooled
    // per-fit state (device pointers owned through DevBufs of the rank thread; shared with peers for the pulls)
    double* A = nullptr;
    long ld = 0, m_loc = 0, n_loc = 0;
    double* Lkk = nullptr;
    double* Winv = nullptr;      // −inv(L_kk) of the diagonal blocks this rank owns, one NB×LDP slot per block (never rewritten within a fit: peers pull from it)
    double* stage[4] = {nullptr, nullptr, nullptr, nullptr};
    double* acc = nullptr;       // backward sweep: per local column partial sums
    double* alpha_blk = nullptr; // backward sweep: α blocks computed by this rank (diagonal owner), indexed by global block
    double* xs = nullptr;        // scaled inputs [d][npad] (kept until the fit's buffers are released: the self-check reads them)
    double* ver = nullptr;       // self-check: this rank's rows of K·α
    double* sacc = nullptr;      // forward solve: accumulators −Σ_j X_j L_ijᵀ of this rank's block rows  [nsp][nlb_r·NB + 32]
    double* sxown = nullptr;     // forward solve: the solution blocks X_k this rank owns (diagonal owner)  [nsp][n_own·NB + 32]
    long sacc_ld = 0, sxown_ld = 0;
    double* A2 = nullptr;        // sequential update: this rank's piece of the EXTENDED factor (new block rows are filled by the forward solve)
    long ld2 = 0;
    int* flags = nullptr;        // "multi_check": marker flags of this rank's events
    int* log = nullptr;          // "multi_check": findings
    int32_t rc = 0;
    std::string err;
    double gemm_ms = 0, gemm_flops = 0;
    long gemm_launches = 0;
    double t_ms = 0;
};

// R persistent rank threads owned by a multi-device context: a fit, a solve pass, a verification pass or an update hands every rank its
// closure (run(fn): fn(r) on thread r) and waits for all of them.  Until round 5 every such pass created and joined R std::threads
// (irrelevant at C4, measurable at C2-sized fits and in optimiser loops: the round-4 review's item 14).  Errors stay per thread
// (gp_last_error is thread-local; the closures copy the text into their rank's record before they return, as before).
class RankPool {
  public:
    explicit RankPool(int R) : n_(R) {
        for (int r = 0; r < R; ++r) th_.emplace_back([this, r]() { loop(r); });
    }
    RankPool(const RankPool&) = delete;
    ~RankPool() {
        {
            std::lock_guard<std::mutex> l(mu_);
            stop_ = true;
        }
        cv_start_.notify_all();
        for (auto& t : th_) t.join();
    }
    // One pass at a time: every caller holds its ctx's mutex today, but nothing else enforces it — a second caller waits at run_mu_ instead of
    // overwriting the pass in flight (fn_ / left_ / gen_) while the first waits on cv_done_.
    void run(const std::function<void(int)>& fn) {
        std::lock_guard<std::mutex> one(run_mu_);
        std::unique_lock<std::mutex> l(mu_);
        fn_ = &fn;
        left_ = n_;
        ++gen_;
        cv_start_.notify_all();
        cv_done_.wait(l, [&] { return left_ == 0; });
        fn_ = nullptr;
    }

  private:
    void loop(int r) {
        long seen = 0;
        for (;;) {
            const std::function<void(int)>* fn = nullptr;
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_start_.wait(l, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                fn = fn_;
            }
            set_err_text(0, std::string());  // the threads persist across passes: every pass starts with a clean thread-local gp_last_error text
            (*fn)(r);
            {
                std::lock_guard<std::mutex> l(mu_);
                if (--left_ == 0) cv_done_.notify_all();
            }
        }
    }
    int n_;
    std::vector<std::thread> th_;
    std::mutex mu_, run_mu_;
    std::condition_variable cv_start_, cv_done_;
    const std::function<void(int)>* fn_ = nullptr;
    long gen_ = 0;
    int left_ = 0;
    bool stop_ = false;
};

struct gp_multi {
    int P = 1, Q = 1, R = 1;
    long nb = 1024;
    int depth = 2;   // look-ahead depth (columns kept ahead of the bulk update)
    int comm = 2;    // 1 RCCL, 2 peer copies
    bool virt = false;  // several ranks share a device
    std::vector<MRank> ranks;
    std::atomic<int> abort{0};
    long seq = 0;    // fit sequence number (XEvent generations)
    int copy_kernel = 0;  // block copies by copy2d_kernel instead of hipMemcpy2DAsync ("copy_kernel" parameter)
    int debug_sync = 0;   // diagnostic: host-synchronise the rank's streams after every exchange ("multi_debug_sync")
    int check = 0;        // "multi_check" (see the header comment)
    int verify = 1;          // host-side self-check of every fit + one repetition on failure ("multi_verify")
    int dist_predict = 1;    // predictive variances of a multi-device posterior on the distributed factor ("multi_dist_predict"; 0: gather first)
    long solves = 0;         // forward solves on the distributed factor so far
    int inject_fault = 0;    // diagnostic: the next fit hands the self-check a spoiled alpha once ("multi_inject_fault")
    long fits = 0, retries = 0;  // fit attempts / repetitions after a failed self-check (gp_ctx_multi_stats)
    int chain_cus = 0;       // "multi_chain_cus": CUs reserved for the diagonal block's chain (0: the chain shares the machine with the bulk update)
    bool dry_chain = false;  // (schedule trace only: the chain-stream schedule without streams)
    int trsm_inv = 1;        // rows-below solve of a panel as ONE triangular-k GEMM with the diagonal owner's −inv(L_kk) ("multi_trsm_inv"; 0: substitution recursion;
                             // 1: the inverse level by level in batched launches when nb = 64·2^m; 2: the inverse by the restricted-row recursion on the identity)
    bool use_inv = true;     // ... for the fit in flight (multi_fit: trsm_inv and the conditioning bound of the inputs)
    long window = 16;        // block steps a rank thread may queue ahead of its device ("multi_window")
    double timeout_s = 600;  // a rank thread that waits longer than this for a peer or for its own streams fails the fit
    std::string comm_note;
    Trace* tr = nullptr;  // schedule trace of the running fit
    std::unique_ptr<RankPool> pool;  // the rank threads (created with the ranks; dry-run traces on a stack gp_multi spawn their own)
    void run_ranks(const std::function<void(int)>& fn) {
        if (!pool) pool.reset(new RankPool(R));
        pool->run(fn);
    }
};

struct gp_multi_post {
    int P = 1, Q = 1;  // (copied: the posterior may outlive the ctx's gp_multi)
    long n = 0, npad = 0, nblk = 0, nb = 0;
    // real points per block (a prefix of the block; the rest is identity padding): after a fit only the tail of the LAST blocks is
    // padding, after a sequential update on the pieces every batch of observations ends in its own padded blocks
    std::vector<long> valid;
    struct Piece {
        gp_ctx* c;
        void* A;
        long ld, m_loc, n_loc;
    };
    std::vector<Piece> pieces;  // one per rank
};

namespace {

inline long nlb_before(long k, long p, long P) { return k >= p ? (k - p) / P + 1 : 0; }  // #global blocks i <= k with i ≡ p (mod P)

struct Dims {
    long n, npad, nblk, NB, nlb_r, nlb_c, LDP;
    int d;
};

#define MCHK(expr)                                                          \
    do {                                                                    \
        hipError_t e_ = (expr);                                             \
        if (e_ != hipSuccess) return set_hip_err(e_, #expr, __LINE__);      \
    } while (0)

using Clock = std::chrono::steady_clock;

// ---- one rank thread's view of its streams: every operation of the schedule goes through here --------------------------
struct RankRun {
    gp_multi* M;
    MRank* me;
    Dims dm;
    long seq;
    bool dry;      // schedule trace only: no device, no HIP call
    int check;
    Trace* tr;
    hipStream_t st[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t own_used = 0;
    long nflags = 0;

    int32_t own_event(Ev* out, const char* tag = "", long k = 0) {
        out->owner = me->r;
        out->tag = tag;
        out->k = k;
        out->id = (int)(NXKIND * dm.nblk + (long)own_used);
        if (dry) {
            ++own_used;
            out->ev = nullptr;
            return 0;
        }
        if (own_used == me->own.size()) {
            hipEvent_t e;
            MCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            me->own.push_back(e);
        }
        out->ev = me->own[own_used++];
        return 0;
    }
    // a traced stream operation: fn issues the device work on stream s (skipped in a dry run)
    template <class F>
    int32_t op(int s, const char* name, long k0, long k1, std::initializer_list<Fp> rd, std::initializer_list<Fp> wr, F&& fn) {
        if (tr) {
            char b[96];
            snprintf(b, sizeof b, "{\"t\":\"op\",\"r\":%d,\"s\":\"%s\",\"n\":\"%s\",\"k\":[%ld,%ld],\"R\":", me->r, SNAME[s], name, k0, k1);
            tr->line(std::string(b) + Trace::fps(rd) + ",\"W\":" + Trace::fps(wr) + "}");
        }
        if (dry) return 0;
        return fn();
    }
    int32_t rec(int s, const Ev& e) {
        if (tr) {
            char b[160];
            snprintf(b, sizeof b, "{\"t\":\"rec\",\"r\":%d,\"s\":\"%s\",\"e\":\"r%de%d\",\"tag\":\"%s\",\"k\":%ld}", me->r, SNAME[s], e.owner, e.id, e.tag, e.k);
            tr->line(b);
        }
        if (dry) return 0;
        if ((check & 1) && e.id < nflags) {
            hipLaunchKernelGGL(mk_mark_kernel, dim3(1), dim3(1), 0, st[s], M->ranks[(size_t)e.owner].flags + e.id, (int)seq);
            MCHK(hipGetLastError());
        }
        MCHK(hipEventRecord(e.ev, st[s]));
        return 0;
    }
    int32_t wait(int s, const Ev& e) {
        if (e.id < 0) return 0;  // never recorded
        if (tr) {
            char b[160];
            snprintf(b, sizeof b, "{\"t\":\"wait\",\"r\":%d,\"s\":\"%s\",\"e\":\"r%de%d\",\"tag\":\"%s\",\"k\":%ld}", me->r, SNAME[s], e.owner, e.id, e.tag, e.k);
            tr->line(b);
        }
        if (dry) return 0;
        // "multi_debug_sync": 4 = every wait on the host; 32 / 64 / 128 = only the waits for arrived / for a peer's events / for the
        // buffer-reuse events (bulk_done, la_done) on the host — localisation of the first-fit item
        const int dbg = M->debug_sync;
        const bool peer = !strcmp(e.tag, "ready") || !strcmp(e.tag, "lkk") || !strcmp(e.tag, "accr") || !strcmp(e.tag, "alr");
        const bool host = (dbg & 4) || ((dbg & 32) && !strcmp(e.tag, "arrived")) || ((dbg & 64) && peer) ||
                          ((dbg & 128) && (!strcmp(e.tag, "bulk_done") || !strcmp(e.tag, "la_done")));
        if (host) MCHK(hipEventSynchronize(e.ev));
        else MCHK(hipStreamWaitEvent(st[s], e.ev, 0));
        if ((check & 1) && e.id < nflags) {
            hipLaunchKernelGGL(mk_check_kernel, dim3(1), dim3(1), 0, st[s], M->ranks[(size_t)e.owner].flags + e.id, (int)seq, me->log, e.owner,
                               e.id, s);
            MCHK(hipGetLastError());
        }
        return 0;
    }
    int32_t publish(XEvent& x, int s) {
        RC(rec(s, x));
        x.gen.store(seq, std::memory_order_release);
        return 0;
    }
    // make stream s of THIS rank wait for x of another rank's thread
    int32_t await(XEvent& x, int s) {
        long spins = 0;
        const auto t0 = Clock::now();
        while (x.gen.load(std::memory_order_acquire) < seq) {
            if (M->abort.load(std::memory_order_relaxed)) return set_err_text(-1999, "multi-device fit aborted: another rank failed");
            if (++spins > 64) {
                std::this_thread::yield();
                if ((spins & 1023) == 0 && std::chrono::duration<double>(Clock::now() - t0).count() > M->timeout_s) {
                    char b[160];
                    snprintf(b, sizeof b, "rank %d waited more than %.0f s for event r%de%d of a peer (host level)", me->r, M->timeout_s, x.owner, x.id);
                    return set_err_text(-1993, b);
                }
            }
        }
        return wait(s, x);
    }
    // 2-D block copy into one of my buffers from a buffer of rank src (possibly on another device), on my stream s
    int32_t pull(int s, double* dst, long dld, const double* src, long sld, long rows, long cols) {
        if (rows <= 0 || cols <= 0) return 0;
        if (M->copy_kernel && !(cols & 1)) return eng_copy2d(me->c, st[s], dst, dld, src, sld, rows, cols);
        MCHK(hipMemcpy2DAsync(dst, sizeof(double) * dld, src, sizeof(double) * sld, sizeof(double) * cols, rows, hipMemcpyDefault, st[s]));
        return 0;
    }
    // "multi_check" & 2: my copy of a block must equal the owner's final block (bitwise)
    int32_t verify(int s, const double* mine, long mld, const double* theirs, long tld, long rows, long cols, int code, long k, long blk) {
        if (!(check & 2) || dry || rows <= 0 || cols <= 0) return 0;
        hipLaunchKernelGGL(mk_cmp_kernel, dim3((unsigned)std::min<long>(64, (rows * cols + 255) / 256)), dim3(256), 0, st[s], mine, mld, theirs, tld,
                           rows, cols, me->log, code, (int)k, (int)blk);
        MCHK(hipGetLastError());
        return 0;
    }
    // wait (bounded) until stream s has drained
    int32_t drain(int s) {
        if (dry) return 0;
        const auto t0 = Clock::now();
        long polls = 0;
        while (true) {
            const hipError_t e = hipStreamQuery(st[s]);
            if (e == hipSuccess) return 0;
            if (e != hipErrorNotReady) return set_hip_err(e, "hipStreamQuery", __LINE__);
            (void)hipGetLastError();
            if (++polls > 200) std::this_thread::sleep_for(std::chrono::microseconds(50));
            else std::this_thread::yield();
            if ((polls & 255) == 0 && std::chrono::duration<double>(Clock::now() - t0).count() > M->timeout_s) {
                char b[160];
                snprintf(b, sizeof b, "rank %d: stream %s did not drain within %.0f s (device-level wait that never completes)", me->r, SNAME[s],
                         M->timeout_s);
                return set_err_text(-1992, b);
            }
        }
    }
};

int32_t nccl_err(int rc, const char* what) {
    if (rc == 0) return 0;
    return set_err_text(-1998, std::string("RCCL error in ") + what + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"));
}
#define NCHK(expr) RC(nccl_err((expr), #expr))

// trace lines of the point-to-point transfers (RCCL matches the i-th send of a pair with the i-th receive of the peer)
void tr_send(RankRun& rr, int s, int peer, long count, Fp f) {
    if (!rr.tr) return;
    char b[128];
    snprintf(b, sizeof b, "{\"t\":\"send\",\"r\":%d,\"s\":\"%s\",\"to\":%d,\"n\":%ld,\"R\":", rr.me->r, SNAME[s], peer, count);
    rr.tr->line(std::string(b) + Trace::fps({f}) + "}");
}
void tr_group(RankRun& rr, int s, int begin) {  // the transfers between the two lines form ONE group: unordered among themselves
    if (!rr.tr) return;
    char b[96];
    snprintf(b, sizeof b, "{\"t\":\"grp\",\"r\":%d,\"s\":\"%s\",\"b\":%d}", rr.me->r, SNAME[s], begin);
    rr.tr->line(b);
}
void tr_recv(RankRun& rr, int s, int peer, long count, Fp f) {
    if (!rr.tr) return;
    char b[128];
    snprintf(b, sizeof b, "{\"t\":\"recv\",\"r\":%d,\"s\":\"%s\",\"from\":%d,\"n\":%ld,\"W\":", rr.me->r, SNAME[s], peer, count);
    rr.tr->line(std::string(b) + Trace::fps({f}) + "}");
}

// The whole pair on one rank.  Y-columns ride as RHS rows; alpha (column 0) only when want_alpha.
// dry: only the schedule (operations, events, footprints) is produced — the same control flow, no device.
int32_t fit_rank(gp_multi* M, MRank* me, const Dims& dm, bool dry, int kind, double variance, const double* xs_h, const double* noise_h,
                 const double* rhs_h /* ncols × npad */, int ncols, bool want_alpha, bool keep, double* alpha_host /* npad, pinned */,
                 double* scal_out /* [0] Σlog L_ii, [1] unused, [8+s] ‖z_s‖² partial */, int* info_out, DevBufs* bufs, long seq) {
    const int P = M->P, Q = M->Q, p = me->p, q = me->q, depth = M->depth;
    const long NB = dm.NB, nblk = dm.nblk, npad = dm.npad, n = dm.n, LDP = dm.LDP;
    const long nlb_r = dm.nlb_r, nlb_c = dm.nlb_c;
    const bool rhs_row = (p == 0);
    const int RHSF = rhs_row ? 2 : 0;  // footprint flag: the RHS block row rides along
    const long m_loc = nlb_r * NB + (rhs_row ? RHS_ROWS : 0), n_loc = nlb_c * NB;
    const long ld = n_loc + 32;
    const int NBUF = depth + 1;
    const int R_ = me->r;
    gp_ctx* c = me->c;
    RankRun rr{M, me, dm, seq, dry, dry ? 0 : M->check, M->tr};
    // "multi_chain_cus": the diagonal block's Cholesky (and its inverse) on a stream of their own that OWNS a few CUs, everything else of the panel and
    // main streams on the complement — beside an unmasked bulk update a 1 024-column block's chain takes 3.6 ms instead of 0.36 (its workgroups wait
    // for slots the update refills), on 16 masked CUs 0.46 ms while the update on the other 240 runs 4 % slower (profiles/r5/cumask_chain_probe.jsonl)
    const bool chain = P > 1 && (dry ? M->dry_chain : (M->chain_cus > 0 && me->sd && me->sm_m && me->sp_m));
    if (!dry) {
        rr.st[SM] = chain ? me->sm_m : c->sm;
        rr.st[SP] = chain ? me->sp_m : c->sp;
        rr.st[SC] = me->sc;
        rr.st[SD] = chain ? me->sd : rr.st[SP];
        MCHK(hipSetDevice(me->device));
        c->ev_used = 0;
        c->gemm_recs.clear();
        if (!c->info_dev) MCHK(hipMalloc((void**)&c->info_dev, sizeof(int)));
        RC(ctx_scal(c, 8 + RHS_ROWS));
    }
    hipStream_t sm = rr.st[SM], sp = rr.st[SP], sc = rr.st[SC];
    const bool rccl = (M->comm == 1);
    const int check = rr.check;

    // ---- buffers
    void *A_v = 0, *xs_v = 0, *nz_v = 0, *Lkk_v = 0, *acc_v = 0, *ab_v = 0, *tmp_v = 0, *chk_v = 0, *ver_v = 0, *Wi_v = 0, *Iw_v = 0, *Sx_v = 0, *Vs_v = 0;
    // rows-below solve by the explicit inverse of the diagonal block (M->use_inv): the diagonal owner of block column k forms W_k = −inv(L_kk) into ITS
    // slot k / lcm(P, Q) of Winv (a slot per owned diagonal block: nothing is rewritten while a peer may still read it) and W_k travels instead of L_kk;
    // every owner of the column then solves its rows with ONE triangular-k MFMA GEMM, S = −X W_kᵀ = X L_kk⁻ᵀ, copied back over X — instead of the
    // recursion's NB/64 latency-bound leaf launches and as many few-tile GEMMs per rank and step (NB = 1 024: 31 launches -> 2)
    const bool use_inv = M->use_inv && P > 1;
    const long LCM = [&] { long a = P, b = Q; while (b) { const long t = a % b; a = b; b = t; } return (long)P / a * Q; }();
    const long n_own = use_inv ? (nblk + LCM - 1) / LCM : 0;
    void* Ab_v[4] = {0, 0, 0, 0};
    void* Bb_v[4] = {0, 0, 0, 0};
    void* St_v[4] = {0, 0, 0, 0};
    const size_t A_b = sizeof(double) * (size_t)(m_loc + 128) * ld;
    const long own_cap = 12 * nblk + 64;
    rr.nflags = NXKIND * nblk + own_cap;
    if (!dry) {
        RC(bufs->get(A_b, &A_v));
        RC(bufs->get(sizeof(double) * (size_t)dm.d * npad, &xs_v));
        RC(bufs->get(sizeof(double) * (size_t)npad, &nz_v));
        RC(bufs->get(sizeof(double) * (size_t)(NB + 128) * LDP, &Lkk_v));
        RC(bufs->get(sizeof(double) * (size_t)(n_loc + 128), &acc_v));
        RC(bufs->get(sizeof(double) * (size_t)npad, &ab_v));
        RC(bufs->get(sizeof(double) * (size_t)NB * (P + 1), &tmp_v));
        RC(bufs->get(sizeof(double) * (size_t)npad, &ver_v));
        for (int s = 0; s < NBUF; ++s) {
            if (Q > 1) RC(bufs->get(sizeof(double) * (size_t)(m_loc + 128) * LDP, &Ab_v[s]));
            RC(bufs->get(sizeof(double) * (size_t)(n_loc + 128) * LDP, &Bb_v[s]));
            if (rccl) RC(bufs->get(sizeof(double) * (size_t)(m_loc + 128) * LDP, &St_v[s]));
        }
        if (check) RC(bufs->get(sizeof(int) * (size_t)(rr.nflags + 256), &chk_v));
        if (use_inv) {
            RC(bufs->get(sizeof(double) * (size_t)(n_own * NB + 128) * LDP, &Wi_v));
            RC(bufs->get(sizeof(double) * (size_t)(NB + 128) * LDP, &Iw_v));
            if (M->trsm_inv == 1) RC(bufs->get(sizeof(double) * (size_t)(NB + 128) * LDP, &Vs_v));  // second scratch: the level-wise inverse (eng_inv_lower)
            RC(bufs->get(sizeof(double) * (size_t)(m_loc + 128) * LDP, &Sx_v));
        }
    }
    double* A = (double*)A_v;
    me->Winv = (double*)Wi_v;
    me->A = A; me->ld = ld; me->m_loc = m_loc; me->n_loc = n_loc;
    me->Lkk = (double*)Lkk_v;
    me->acc = (double*)acc_v;
    me->alpha_blk = (double*)ab_v;
    me->xs = (double*)xs_v;
    me->ver = (double*)ver_v;
    me->flags = (int*)chk_v;
    me->log = chk_v ? (int*)chk_v + rr.nflags : nullptr;
    for (int s = 0; s < 4; ++s) me->stage[s] = (double*)St_v[s];

    GridMap g = plain_map(1, 0, 0);
    g.P = P; g.p = p; g.Q = Q; g.q = q; g.nb = NB;

    auto lrow_from = [&](long gblk, int pp) { return nlb_before(gblk - 1, pp, P); };          // first local block row (process row pp) with global block >= gblk
    auto rows_from = [&](long gblk, int pp) { return nlb_before(gblk - 1, pp, P) * NB; };     // ... as a row index
    auto mloc_of = [&](int pp) { return nlb_r * NB + (pp == 0 ? RHS_ROWS : 0); };
    auto lcol_from = [&](long gblk) { return nlb_before(gblk - 1, q, Q); };                   // first local block column with global block >= gblk
    auto rank_of = [&](int pp, int qq) -> MRank& { return M->ranks[(size_t)pp * Q + qq]; };
    auto rrank = [&](int pp, int qq) { return pp * Q + qq; };

    // ---- upload + assemble (one traced operation: everything below is issued back to back on the main stream)
    RC(rr.op(SM, "init", 0, 0, {},
             {Fp{"A", R_, 0, nlb_r, 0, nlb_c, RHSF}, Fp{"Ab", R_, 0, NBUF, 0, nlb_r, 2}, Fp{"Bb", R_, 0, NBUF, 0, nlb_c, 0},
              Fp{"St", R_, 0, NBUF, 0, nlb_r, 2}, Fp{"Lkk", R_, 0, 0, 0, 0, 0}, Fp{"acc", R_, 0, 0, 0, nlb_c, 0}, Fp{"Wi", R_, 0, 0, 0, n_own, 0}},
             [&]() -> int32_t {
                 MCHK(hipMemcpyAsync(xs_v, xs_h, sizeof(double) * (size_t)dm.d * npad, hipMemcpyHostToDevice, sm));
                 MCHK(hipMemcpyAsync(nz_v, noise_h, sizeof(double) * (size_t)npad, hipMemcpyHostToDevice, sm));
                 MCHK(hipMemsetAsync(c->info_dev, 0, sizeof(int), sm));
                 MCHK(hipMemsetAsync(c->scal_dev, 0, sizeof(double) * (8 + RHS_ROWS), sm));
                 MCHK(hipMemsetAsync(acc_v, 0, sizeof(double) * (size_t)(n_loc + 128), sm));
                 MCHK(hipMemsetAsync(A + nlb_r * NB * ld, 0, sizeof(double) * (size_t)(m_loc - nlb_r * NB + 128) * ld, sm));  // RHS + slack rows
                 if (chk_v) MCHK(hipMemsetAsync(chk_v, 0, sizeof(int) * (size_t)(rr.nflags + 256), sm));
                 // operand buffers: the 128 slack rows are over-read by the GEMM and must stay finite; the payload rows are zero, or NaN
                 // under "multi_check" & 4 so that a read before arrival cannot pass as a plausible number
                 const int fill = (check & 4) ? 0xFF : 0;
                 for (int s = 0; s < NBUF; ++s) {
                     if (Ab_v[s]) {
                         MCHK(hipMemsetAsync(Ab_v[s], fill, sizeof(double) * (size_t)m_loc * LDP, sm));
                         MCHK(hipMemsetAsync((double*)Ab_v[s] + m_loc * LDP, 0, sizeof(double) * (size_t)128 * LDP, sm));
                     }
                     MCHK(hipMemsetAsync(Bb_v[s], fill, sizeof(double) * (size_t)n_loc * LDP, sm));
                     MCHK(hipMemsetAsync((double*)Bb_v[s] + n_loc * LDP, 0, sizeof(double) * (size_t)128 * LDP, sm));
                     if (St_v[s]) {
                         MCHK(hipMemsetAsync(St_v[s], fill, sizeof(double) * (size_t)m_loc * LDP, sm));
                         MCHK(hipMemsetAsync((double*)St_v[s] + m_loc * LDP, 0, sizeof(double) * (size_t)128 * LDP, sm));
                     }
                 }
                 if (Wi_v) MCHK(hipMemsetAsync(Wi_v, 0, sizeof(double) * (size_t)(n_own * NB + 128) * LDP, sm));  // (slack rows below the last slot stay zero)
                 if (Iw_v) MCHK(hipMemsetAsync(Iw_v, 0, sizeof(double) * (size_t)(NB + 128) * LDP, sm));  // (the level-wise inverse never writes its lower-left triangle)
                 if (Vs_v) MCHK(hipMemsetAsync(Vs_v, 0, sizeof(double) * (size_t)(NB + 128) * LDP, sm));
                 MCHK(hipMemsetAsync(Lkk_v, fill, sizeof(double) * (size_t)NB * LDP, sm));
                 MCHK(hipMemsetAsync((double*)Lkk_v + NB * LDP, 0, sizeof(double) * (size_t)128 * LDP, sm));
                 RC(eng_assemble(c, sm, kind, variance, (const double*)xs_v, n, npad, dm.d, (const double*)nz_v, g, A, ld, nlb_r * NB, n_loc));
                 if (rhs_row) {  // RHS rows: δ_sᵀ restricted to my local columns (global block lj·Q + q)
                     for (int s = 0; s < ncols; ++s)
                         for (long lj = 0; lj < nlb_c; ++lj)
                             MCHK(hipMemcpyAsync(A + (nlb_r * NB + s) * ld + lj * NB, rhs_h + (size_t)s * npad + (lj * Q + q) * NB, sizeof(double) * NB,
                                                 hipMemcpyHostToDevice, sm));
                 }
                 return 0;
             }));
    Ev ev_asm;
    RC(rr.own_event(&ev_asm, "assembled"));
    RC(rr.rec(SM, ev_asm));
    RC(rr.wait(SP, ev_asm));
    RC(rr.wait(SC, ev_asm));

    std::vector<Ev> arrived((size_t)nblk), bulk_done((size_t)nblk), la_done((size_t)nblk);

    // A operand of panel i for my rows: my own matrix columns when I sit in the owner column, else the fetched copy
    auto a_operand = [&](long i, const double** ptr, long* lda) {
        if (q == (int)(i % Q)) {
            *ptr = A + (i / Q) * NB;
            *lda = ld;
        } else {
            *ptr = (const double*)Ab_v[i % NBUF];
            *lda = LDP;
        }
    };
    // C[rows >= global block gr0, local columns [c_lo, c_hi)] -= panel_i(rows) · panel_i(cols)ᵀ   (lower part only)
    auto update = [&](int s, const char* name, long j, long i, long gr0, long c_lo, long c_hi) -> int32_t {
        const long r0 = rows_from(gr0, p);
        const long mrows = m_loc - r0, ncolsu = c_hi - c_lo;
        if (mrows <= 0 || ncolsu <= 0) return 0;
        const long lr0 = r0 / NB, lc0 = c_lo / NB, lc1 = c_hi / NB;
        const int slot = (int)(i % NBUF);
        const bool own_col = (q == (int)(i % Q));
        const Fp fa = own_col ? Fp{"A", R_, lr0, nlb_r, i / Q, i / Q + 1, RHSF} : Fp{"Ab", R_, slot, slot + 1, lr0, nlb_r, RHSF};
        const Fp fc = Fp{"A", R_, lr0, nlb_r, lc0, lc1, 1 | RHSF};
        return rr.op(s, name, j, i, {fa, Fp{"Bb", R_, slot, slot + 1, lc0, lc1, 0}, fc}, {fc}, [&]() -> int32_t {
            const double* ap;
            long lda;
            a_operand(i, &ap, &lda);
            const double* Bb = (const double*)Bb_v[slot];
            if (check & 2) {  // the operands this GEMM is about to read vs the owners' final panel blocks
                const int qi = (int)(i % Q);
                if (!own_col) {
                    MRank& src = rank_of(p, qi);
                    RC(rr.verify(s, ap + r0 * lda, lda, src.A + r0 * src.ld + (i / Q) * NB, src.ld, mrows, NB, 2, i, -1));
                }
                for (long lj = lc0; lj < lc1; ++lj) {
                    const long gj = lj * Q + q;
                    MRank& src = rank_of((int)(gj % P), qi);
                    RC(rr.verify(s, Bb + lj * NB * LDP, LDP, src.A + (gj / P) * NB * src.ld + (i / Q) * NB, src.ld, NB, NB, 3, i, lj));
                }
            }
            GridMap gm = g;
            gm.row0 = r0;
            gm.col0 = c_lo;
            const size_t before = c->gemm_recs.size();
            RC(eng_gemm_nt(c, rr.st[s], A + r0 * ld + c_lo, ld, ap + r0 * lda, lda, Bb + c_lo * LDP, LDP, mrows, ncolsu, NB, gm));
            if (c->gemm_recs.size() > before) {  // exact algorithmic flops of this launch: local elements on/below the global diagonal × 2·NB
                double cnt = 0;
                for (long lj = lc0; lj < lc1; ++lj) {
                    const long gj = lj * Q + q;
                    for (long li = lr0; li < nlb_r; ++li) {
                        const long gi = li * P + p;
                        cnt += gi > gj ? (double)NB * NB : (gi == gj ? (double)NB * (NB + 1) / 2 : 0.0);
                    }
                    if (rhs_row) cnt += (double)RHS_ROWS * NB;
                }
                c->gemm_recs.back().flops = 2.0 * NB * cnt;
            }
            return 0;
        });
    };

    Ev lkk_free;  // RCCL: my L_kk image may be overwritten again after this event
    // factor block column k on its owners (panel stream) and publish it
    auto panel = [&](long k) -> int32_t {
        const int pk = (int)(k % P), qk = (int)(k % Q);
        if (q != qk) return 0;
        const long c0 = (k / Q) * NB, lc = k / Q;
        if (P == 1) {
            const long r0 = k * NB;  // (k / P) * NB
            const Fp f = Fp{"A", R_, k, nlb_r, lc, lc + 1, RHSF};
            RC(rr.op(SP, "potrf", k, 0, {f}, {f}, [&]() { return eng_potrf(c, sp, A + r0 * ld + c0, ld, m_loc - r0, NB, c->info_dev, k * NB, n, c->scal_dev); }));
        } else {
            const double* lkk_ptr = nullptr;
            long lkk_ld = 0;
            Fp f_l;
            // (RCCL: every collective of a rank is issued on ITS comm stream, in the same global order on all ranks — L_kk(k),
            //  then exchange(k) — and chained to the panel stream by events)
            if (p == pk) {
                const long r0 = (k / P) * NB;
                const Fp fd = Fp{"A", R_, k / P, k / P + 1, lc, lc + 1, 0};
                const int SDg = chain ? SD : SP;  // the stream of the diagonal block's chain
                hipStream_t sdg = rr.st[SDg];
                if (chain) {  // the block's last update (look-ahead, panel stream) precedes the chain
                    Ev e;
                    RC(rr.own_event(&e, "to_chain", k));
                    RC(rr.rec(SP, e));
                    RC(rr.wait(SD, e));
                }
                RC(rr.op(SDg, "potrf_diag", k, 0, {fd}, {fd}, [&]() { return eng_potrf(c, sdg, A + r0 * ld + c0, ld, NB, NB, c->info_dev, k * NB, n, c->scal_dev); }));
                lkk_ptr = A + r0 * ld + c0;
                lkk_ld = ld;
                f_l = fd;
                if (use_inv) {  // W_k = −inv(L_kk) into my slot of this block; W_k is what the column's other owners get
                    const long slot = k / LCM;
                    double* Wk = me->Winv + slot * NB * LDP;
                    const Fp fw = Fp{"Wi", R_, 0, 0, slot, slot + 1, 0};
                    const double* Ld = A + r0 * ld + c0;
                    RC(rr.op(SDg, "inv_lkk", k, 0, {fd}, {fw}, [&]() { return eng_inv_lower(c, sdg, Ld, ld, NB, Wk, LDP, (double*)Iw_v, (double*)Vs_v); }));
                    lkk_ptr = Wk;
                    lkk_ld = LDP;
                    f_l = fw;
                }
                if (chain) {  // ... and the panel stream continues after it (image / sends / publication / my rows-below solve)
                    Ev e;
                    RC(rr.own_event(&e, "from_chain", k));
                    RC(rr.rec(SD, e));
                    RC(rr.wait(SP, e));
                }
                if (use_inv) {
                    const long slot = k / LCM;
                    double* Wk = me->Winv + slot * NB * LDP;
                    const Fp fw = Fp{"Wi", R_, 0, 0, slot, slot + 1, 0};
                    if (rccl) {  // the slot is contiguous and never rewritten: sent as it lies
                        Ev e;
                        RC(rr.own_event(&e, "lkk_image", k));
                        RC(rr.rec(SP, e));
                        RC(rr.wait(SC, e));
                        tr_group(rr, SC, 1);
                        if (!dry) NCHK(g_rccl.GroupStart());
                        for (int pp = 0; pp < P; ++pp)
                            if (pp != pk) {
                                tr_send(rr, SC, rrank(pp, qk), (long)NB * LDP, fw);
                                if (!dry) NCHK(g_rccl.Send(Wk, (size_t)NB * LDP, NCCL_FLOAT64, rank_of(pp, qk).r, me->comm, sc));
                            }
                        if (!dry) NCHK(g_rccl.GroupEnd());
                        tr_group(rr, SC, 0);
                    }
                } else if (rccl) {  // contiguous image for the sends
                    RC(rr.wait(SP, lkk_free));
                    RC(rr.op(SP, "lkk_image", k, 0, {fd}, {Fp{"Lkk", R_, 0, 0, 0, 0, 0}}, [&]() { return rr.pull(SP, me->Lkk, LDP, lkk_ptr, ld, NB, NB); }));
                    Ev e;
                    RC(rr.own_event(&e, "lkk_image", k));
                    RC(rr.rec(SP, e));
                    RC(rr.wait(SC, e));
                    tr_group(rr, SC, 1);
                    if (!dry) NCHK(g_rccl.GroupStart());
                    for (int pp = 0; pp < P; ++pp)
                        if (pp != pk) {
                            tr_send(rr, SC, rrank(pp, qk), (long)NB * LDP, Fp{"Lkk", R_, 0, 0, 0, 0, 0});
                            if (!dry) NCHK(g_rccl.Send(me->Lkk, (size_t)NB * LDP, NCCL_FLOAT64, rank_of(pp, qk).r, me->comm, sc));
                        }
                    if (!dry) NCHK(g_rccl.GroupEnd());
                    tr_group(rr, SC, 0);
                    RC(rr.own_event(&lkk_free, "lkk_free", k));
                    RC(rr.rec(SC, lkk_free));
                }
                RC(rr.publish(me->lkk[k], SP));
            } else {
                MRank& own = rank_of(pk, qk);
                if (rccl) {
                    RC(rr.wait(SC, lkk_free));  // the previous L_kk image has been consumed by my trsm
                    tr_recv(rr, SC, own.r, (long)NB * LDP, Fp{"Lkk", R_, 0, 0, 0, 0, 0});
                    if (!dry) NCHK(g_rccl.Recv(me->Lkk, (size_t)NB * LDP, NCCL_FLOAT64, own.r, me->comm, sc));
                    Ev e;
                    RC(rr.own_event(&e, "lkk_recv", k));
                    RC(rr.rec(SC, e));
                    RC(rr.wait(SP, e));
                } else {
                    RC(rr.await(own.lkk[k], SP));
                    if (use_inv) {
                        const long slot = k / LCM;
                        RC(rr.op(SP, "pull_lkk", k, 0, {Fp{"Wi", own.r, 0, 0, slot, slot + 1, 0}}, {Fp{"Lkk", R_, 0, 0, 0, 0, 0}},
                                 [&]() { return rr.pull(SP, me->Lkk, LDP, own.Winv + slot * NB * LDP, LDP, NB, NB); }));
                    } else {
                        RC(rr.op(SP, "pull_lkk", k, 0, {Fp{"A", own.r, k / P, k / P + 1, lc, lc + 1, 0}}, {Fp{"Lkk", R_, 0, 0, 0, 0, 0}},
                                 [&]() { return rr.pull(SP, me->Lkk, LDP, own.A + (k / P) * NB * own.ld + c0, own.ld, NB, NB); }));
                    }
                }
                lkk_ptr = me->Lkk;
                lkk_ld = LDP;
                f_l = Fp{"Lkk", R_, 0, 0, 0, 0, 0};
            }
            const long r0b = rows_from(k + 1, p);
            if (m_loc - r0b > 0) {
                const Fp fx = Fp{"A", R_, r0b / NB, nlb_r, lc, lc + 1, RHSF};
                RC(rr.op(SP, "trsm", k, 0, {fx, f_l}, {fx}, [&]() {
                    return use_inv ? eng_trsm_inv(c, sp, A + r0b * ld + c0, ld, m_loc - r0b, lkk_ptr, lkk_ld, NB, (double*)Sx_v, LDP)
                                   : eng_trsm(c, sp, A + r0b * ld + c0, ld, m_loc - r0b, lkk_ptr, lkk_ld, NB);
                }));
            }
            if (rccl && p != pk) {
                RC(rr.own_event(&lkk_free, "lkk_free", k));
                RC(rr.rec(SP, lkk_free));
            }
        }
        if (rccl) {  // contiguous image of my piece (rows below k, all of them incl. RHS rows) for the sends of exchange(k)
            const long r0b = rows_from(k + 1, p);
            if (m_loc - r0b > 0) {
                const int slot = (int)(k % NBUF);
                RC(rr.op(SP, "stage", k, 0, {Fp{"A", R_, r0b / NB, nlb_r, lc, lc + 1, RHSF}}, {Fp{"St", R_, slot, slot + 1, r0b / NB, nlb_r, RHSF}},
                         [&]() { return rr.pull(SP, me->stage[slot] + r0b * LDP, LDP, A + r0b * ld + c0, ld, m_loc - r0b, NB); }));
            }
        }
        RC(rr.publish(me->ready[k], SP));
        if (!dry && (M->debug_sync & 8)) MCHK(hipStreamSynchronize(sp));
        return 0;
    };

    // fetch what I consume of panel k into buffer set k % NBUF (comm stream)
    auto exchange = [&](long k) -> int32_t {
        const int qk = (int)(k % Q);
        const int s = (int)(k % NBUF);
        const long lc = k / Q;
        if (k - NBUF >= 0) {  // the set's previous panel must have been consumed
            RC(rr.wait(SC, bulk_done[k - NBUF]));
            RC(rr.wait(SC, la_done[k - NBUF]));
        }
        double* Ab = (double*)Ab_v[s];
        double* Bb = (double*)Bb_v[s];
        // my own piece of panel k (when I sit in the owner column) is read straight from my matrix by my updates: arrived[k]
        // must therefore also cover MY panel-stream work — with gcd(P, Q) > 1 a rank may fetch nothing from itself below
        if (q == qk) RC(rr.wait(SC, me->ready[k]));
        if (!rccl) {
            // A part: my process row's piece, rows in local row order (nothing to fetch when I own the column)
            if (q != qk) {
                MRank& src = rank_of(p, qk);
                const long r0 = rows_from(k + 1, p);
                if (m_loc - r0 > 0) {
                    RC(rr.await(src.ready[k], SC));
                    RC(rr.op(SC, "pullA", k, 0, {Fp{"A", src.r, r0 / NB, nlb_r, lc, lc + 1, RHSF}}, {Fp{"Ab", R_, s, s + 1, r0 / NB, nlb_r, RHSF}},
                             [&]() { return rr.pull(SC, Ab + r0 * LDP, LDP, src.A + r0 * src.ld + lc * NB, src.ld, m_loc - r0, NB); }));
                }
            }
            // B part: the global blocks of my process column, rows in local column order
            for (int pp = 0; pp < P; ++pp) {
                MRank& src = rank_of(pp, qk);
                bool waited = false;
                for (long lj = nlb_before(k, q, Q); lj < nlb_c; ++lj) {
                    const long gj = lj * Q + q;
                    if ((int)(gj % P) != pp) continue;
                    if (!waited) {
                        RC(rr.await(src.ready[k], SC));
                        waited = true;
                    }
                    RC(rr.op(SC, "pullB", k, lj, {Fp{"A", src.r, gj / P, gj / P + 1, lc, lc + 1, 0}}, {Fp{"Bb", R_, s, s + 1, lj, lj + 1, 0}},
                             [&]() { return rr.pull(SC, Bb + lj * NB * LDP, LDP, src.A + (gj / P) * NB * src.ld + lc * NB, src.ld, NB, NB); }));
                }
            }
        } else {
            // the same transfers as matched ncclSend / ncclRecv pairs of ONE group per rank (all peers progress concurrently);
            // both sides enumerate (source process row, destination rank, block) in the same order
            tr_group(rr, SC, 1);
            if (!dry) NCHK(g_rccl.GroupStart());
            for (int pp = 0; pp < P; ++pp) {
                MRank& src = rank_of(pp, qk);
                const long r0s = rows_from(k + 1, pp), ms = mloc_of(pp);
                const int srhs = pp == 0 ? 2 : 0;
                for (int dp = 0; dp < P; ++dp)
                    for (int dq = 0; dq < Q; ++dq) {
                        MRank& dst = rank_of(dp, dq);
                        const bool i_send = (&src == me), i_recv = (&dst == me);
                        if (!i_send && !i_recv) continue;
                        if (dp == pp && dq != qk && ms - r0s > 0) {  // A part
                            if (i_send) {
                                tr_send(rr, SC, dst.r, (ms - r0s) * LDP, Fp{"St", R_, s, s + 1, r0s / NB, nlb_r, srhs});
                                if (!dry) NCHK(g_rccl.Send(me->stage[s] + r0s * LDP, (size_t)(ms - r0s) * LDP, NCCL_FLOAT64, dst.r, me->comm, sc));
                            }
                            if (i_recv) {
                                tr_recv(rr, SC, src.r, (ms - r0s) * LDP, Fp{"Ab", R_, s, s + 1, r0s / NB, nlb_r, srhs});
                                if (!dry) NCHK(g_rccl.Recv(Ab + r0s * LDP, (size_t)(ms - r0s) * LDP, NCCL_FLOAT64, src.r, me->comm, sc));
                            }
                        }
                        for (long lj = nlb_before(k, dq, Q); lj < nlb_c; ++lj) {  // B part
                            const long gj = lj * Q + dq;
                            if ((int)(gj % P) != pp) continue;
                            const long srow = (gj / P) * NB;
                            if (i_send && i_recv) {
                                RC(rr.op(SC, "selfB", k, lj, {Fp{"St", R_, s, s + 1, gj / P, gj / P + 1, 0}}, {Fp{"Bb", R_, s, s + 1, lj, lj + 1, 0}},
                                         [&]() { return rr.pull(SC, Bb + lj * NB * LDP, LDP, me->stage[s] + srow * LDP, LDP, NB, NB); }));
                            } else if (i_send) {
                                tr_send(rr, SC, dst.r, NB * LDP, Fp{"St", R_, s, s + 1, gj / P, gj / P + 1, 0});
                                if (!dry) NCHK(g_rccl.Send(me->stage[s] + srow * LDP, (size_t)NB * LDP, NCCL_FLOAT64, dst.r, me->comm, sc));
                            } else {
                                tr_recv(rr, SC, src.r, NB * LDP, Fp{"Bb", R_, s, s + 1, lj, lj + 1, 0});
                                if (!dry) NCHK(g_rccl.Recv(Bb + lj * NB * LDP, (size_t)NB * LDP, NCCL_FLOAT64, src.r, me->comm, sc));
                            }
                        }
                    }
            }
            if (!dry) NCHK(g_rccl.GroupEnd());
            tr_group(rr, SC, 0);
        }
        RC(rr.own_event(&arrived[k], "arrived", k));
        RC(rr.rec(SC, arrived[k]));
        if (!dry) {
            if (M->debug_sync & 1) MCHK(hipStreamSynchronize(sc));
            if (M->debug_sync & 2) {
                MCHK(hipStreamSynchronize(sp));
                MCHK(hipStreamSynchronize(sm));
            }
        }
        return 0;
    };

    // look-ahead update of block column j with panel i on the panel stream
    auto la_update = [&](long j, long i) -> int32_t {
        if (q != (int)(j % Q)) return 0;
        RC(rr.wait(SP, arrived[i]));
        const long first = std::max(0L, j - depth);
        if (i == first && first - 1 >= 0) RC(rr.wait(SP, bulk_done[first - 1]));
        const long c0 = (j / Q) * NB;
        return update(SP, "la", j, i, j, c0, c0 + NB);
    };

    RC(panel(0));
    RC(exchange(0));
    for (long k = 0; k < nblk; ++k) {
        if (k + 1 < nblk) {
            RC(la_update(k + 1, k));  // column k+1 first: it is on the critical path
            RC(panel(k + 1));
            RC(exchange(k + 1));
            for (long j = k + 2; j <= std::min(k + depth, nblk - 1); ++j) RC(la_update(j, k));
        }
        RC(rr.own_event(&la_done[k], "la_done", k));
        RC(rr.rec(SP, la_done[k]));
        // bulk of the trailing update: every local column right of the look-ahead window
        RC(rr.wait(SM, arrived[k]));
        const long gfirst = k + depth + 1;
        if (gfirst < nblk) RC(update(SM, "bulk", gfirst, k, gfirst, lcol_from(gfirst) * NB, n_loc));
        RC(rr.own_event(&bulk_done[k], "bulk_done", k));
        RC(rr.rec(SM, bulk_done[k]));
        // bounded run-ahead of the host: at most `window` block steps are queued beyond what the device has finished (hundreds of
        // queued steps on 16+ streams of mixed priority stopped making progress in tools/hip_event_repro.hip — profiles/r3)
        if (!dry && k >= M->window && bulk_done[k - M->window].ev) MCHK(hipEventSynchronize(bulk_done[k - M->window].ev));
        if (!dry && (M->debug_sync & 16)) MCHK(hipStreamSynchronize(sm));
    }
    // join the panel and comm streams into the main stream
    {
        Ev e;
        RC(rr.own_event(&e, "join_sp"));
        RC(rr.rec(SP, e));
        RC(rr.wait(SM, e));
        RC(rr.own_event(&e, "join_sc"));
        RC(rr.rec(SC, e));
        RC(rr.wait(SM, e));
    }
    // ---- ‖z_s‖² over my local columns of the RHS rows
    if (rhs_row)
        RC(rr.op(SM, "rowsumsq", 0, 0, {Fp{"A", R_, 0, 0, 0, nlb_c, 4}}, {}, [&]() { return eng_rowsumsq(c, sm, A + nlb_r * NB * ld, ld, ncols, n_loc, c->scal_dev + 8); }));

    // ---- backward substitution α = L⁻ᵀ z (column 0), block sweep from the last block column
    if (want_alpha) {
        double* acc = me->acc;
        double* tmp = (double*)tmp_v;
        for (long k = nblk - 1; k >= 0; --k) {
            const int pk = (int)(k % P), qk = (int)(k % Q);
            const long c0 = (k / Q) * NB, lc = k / Q;
            const Fp f_acc = Fp{"acc", R_, 0, 0, lc, lc + 1, 0};
            const Fp f_alb = Fp{"alb", R_, 0, 0, k, k + 1, 0};
            if (q == qk) {
                if (rhs_row)  // + z_k from the RHS row
                    RC(rr.op(SM, "addz", k, 0, {Fp{"A", R_, 0, 0, lc, lc + 1, 4}, f_acc}, {f_acc}, [&]() { return eng_add_vec(c, sm, acc + c0, A + nlb_r * NB * ld + c0, NB); }));
                if (p == pk) {
                    double* ak = me->alpha_blk + k * NB;
                    RC(rr.op(SM, "ak", k, 0, {f_acc}, {f_alb}, [&]() -> int32_t {
                        MCHK(hipMemcpyAsync(ak, acc + c0, sizeof(double) * NB, hipMemcpyDeviceToDevice, sm));
                        return 0;
                    }));
                    for (int pp = 0; pp < P; ++pp) {  // + the partial sums of the other process rows
                        if (pp == pk) continue;
                        MRank& src = rank_of(pp, qk);
                        const Fp f_tmp = Fp{"tmp", R_, 0, 0, pp, pp + 1, 0};
                        if (rccl) {
                            tr_recv(rr, SM, src.r, NB, f_tmp);
                            if (!dry) NCHK(g_rccl.Recv(tmp + (size_t)pp * NB, (size_t)NB, NCCL_FLOAT64, src.r, me->comm, sm));
                        } else {
                            RC(rr.await(src.accr[k], SM));
                            RC(rr.op(SM, "pull_acc", k, pp, {Fp{"acc", src.r, 0, 0, lc, lc + 1, 0}}, {f_tmp}, [&]() { return rr.pull(SM, tmp + (size_t)pp * NB, NB, src.acc + c0, NB, 1, NB); }));
                        }
                        RC(rr.op(SM, "add_acc", k, pp, {f_tmp, f_alb}, {f_alb}, [&]() { return eng_add_vec(c, sm, ak, tmp + (size_t)pp * NB, NB); }));
                    }
                    RC(rr.op(SM, "trsv", k, 0, {Fp{"A", R_, k / P, k / P + 1, lc, lc + 1, 0}, f_alb}, {f_alb}, [&]() -> int32_t {
                        RC(eng_trsv(c, sm, A + (k / P) * NB * ld + c0, ld, NB, ak, NB, 1, false));
                        MCHK(hipMemcpyAsync(alpha_host + k * NB, ak, sizeof(double) * NB, hipMemcpyDeviceToHost, sm));
                        return 0;
                    }));
                    if (rccl) {  // α_k to the ranks of my process row that hold columns left of k (exactly those post the receive)
                        bool any = false;
                        for (int qq = 0; qq < Q; ++qq) any = any || (qq != qk && k > 0 && nlb_before(k - 1, qq, Q) > 0);
                        if (any) {
                            tr_group(rr, SM, 1);
                            if (!dry) NCHK(g_rccl.GroupStart());
                            for (int qq = 0; qq < Q; ++qq)
                                if (qq != qk && k > 0 && nlb_before(k - 1, qq, Q) > 0) {
                                    tr_send(rr, SM, rrank(pk, qq), NB, f_alb);
                                    if (!dry) NCHK(g_rccl.Send(ak, (size_t)NB, NCCL_FLOAT64, rank_of(pk, qq).r, me->comm, sm));
                                }
                            if (!dry) NCHK(g_rccl.GroupEnd());
                            tr_group(rr, SM, 0);
                        }
                    }
                    RC(rr.publish(me->alr[k], SM));
                } else {
                    if (rccl) {
                        tr_send(rr, SM, rrank(pk, qk), NB, f_acc);
                        if (!dry) NCHK(g_rccl.Send(acc + c0, (size_t)NB, NCCL_FLOAT64, rank_of(pk, qk).r, me->comm, sm));
                    }
                    RC(rr.publish(me->accr[k], SM));
                }
            }
            if (p == pk && k > 0) {  // my block row k: acc_j −= L[k][j]ᵀ α_k for my local columns j < k
                const long ncb = nlb_before(k - 1, q, Q);
                if (ncb > 0) {
                    const double* ak;
                    Fp f_ak = f_alb;
                    if (q == qk) {
                        ak = me->alpha_blk + k * NB;
                    } else {
                        MRank& own = rank_of(pk, qk);
                        double* dst = tmp + (size_t)P * NB;
                        f_ak = Fp{"tmp", R_, 0, 0, P, P + 1, 0};
                        if (rccl) {
                            tr_recv(rr, SM, own.r, NB, f_ak);
                            if (!dry) NCHK(g_rccl.Recv(dst, (size_t)NB, NCCL_FLOAT64, own.r, me->comm, sm));
                        } else {
                            RC(rr.await(own.alr[k], SM));
                            RC(rr.op(SM, "pull_alpha", k, 0, {Fp{"alb", own.r, 0, 0, k, k + 1, 0}}, {f_ak}, [&]() { return rr.pull(SM, dst, NB, own.alpha_blk + k * NB, NB, 1, NB); }));
                        }
                        ak = dst;
                    }
                    const Fp f_accs = Fp{"acc", R_, 0, 0, 0, ncb, 0};
                    RC(rr.op(SM, "gemv", k, 0, {Fp{"A", R_, k / P, k / P + 1, 0, ncb, 0}, f_ak, f_accs}, {f_accs},
                             [&]() { return eng_gemv_t(c, sm, A + (k / P) * NB * ld, ld, NB, ncb * NB, ak, acc); }));
                }
            }
        }
    }
    if (dry) return 0;
    // ---- results of this rank
    std::vector<int> log_h;
    if (chk_v) {
        log_h.assign(256, 0);
        MCHK(hipMemcpyAsync(log_h.data(), me->log, sizeof(int) * 256, hipMemcpyDeviceToHost, sm));
    }
    MCHK(hipMemcpyAsync(info_out, c->info_dev, sizeof(int), hipMemcpyDeviceToHost, sm));
    MCHK(hipMemcpyAsync(scal_out, c->scal_dev, sizeof(double) * (8 + RHS_ROWS), hipMemcpyDeviceToHost, sm));
    RC(rr.drain(SM));
    RC(rr.drain(SP));
    RC(rr.drain(SC));
    if (chain) RC(rr.drain(SD));
    me->gemm_ms = me->gemm_flops = 0;
    me->gemm_launches = (long)c->gemm_recs.size();
    for (auto& r : c->gemm_recs) {
        float ms = 0;
        MCHK(hipEventElapsedTime(&ms, r.a, r.b));
        me->gemm_ms += ms;
        me->gemm_flops += r.flops;
    }
    if (chk_v && log_h[0] > 0) {
        std::string msg = "multi_check: rank " + std::to_string(me->r) + " logged " + std::to_string(log_h[0]) + " finding(s):";
        for (int i = 0; i < std::min(log_h[0], 12); ++i) {
            const int* e = &log_h[4 + 4 * i];
            char b[200];
            if (e[0] == 1)
                snprintf(b, sizeof b, " [stream %s ran past its wait for event r%de%d: marker holds generation %d, expected %ld]", SNAME[(e[3] >> 16) & 3], e[1], e[2],
                         e[3] & 0xffff, seq & 0xffff);
            else
                snprintf(b, sizeof b, " [%s operand of panel %d, local block %d differs from the owner's final block]", e[0] == 2 ? "A" : "B", e[1], e[2]);
            msg += b;
        }
        return set_err_text(-1990, msg);
    }
    if (keep) bufs->keep(A_v);
    return 0;
}


// cross-thread events of every rank sized for nblk block columns (ids: kind · nblk + k — the layout of the marker flags and of the
// trace names); dry: no HIP events behind them
int32_t size_xevents(gp_multi* M, long nblk, bool dry) {
    for (auto& rk : M->ranks) {
        if (!dry) (void)hipSetDevice(rk.device);
        int kind = 0;
        for (auto* v : {&rk.ready, &rk.lkk, &rk.accr, &rk.alr, &rk.sx, &rk.sa, &rk.bar}) {
            if ((long)v->size() < nblk) {
                std::vector<XEvent> nv((size_t)nblk);
                for (size_t i = 0; i < v->size(); ++i) {
                    nv[i].ev = (*v)[i].ev;
                    nv[i].gen.store((*v)[i].gen.load());
                }
                if (!dry)
                    for (size_t i = v->size(); i < (size_t)nblk; ++i)
                        if (hipEventCreateWithFlags(&nv[i].ev, hipEventDisableTiming) != hipSuccess) return set_err_text(-1995, "hipEventCreate failed");
                v->swap(nv);
            }
            for (size_t i = 0; i < v->size(); ++i) {
                (*v)[i].owner = rk.r;
                (*v)[i].id = (long)i < nblk ? (int)(kind * nblk + (long)i) : -1;
                (*v)[i].tag = XTAG[kind];
                (*v)[i].k = (long)i;
            }
            ++kind;
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Forward solve on the distributed factor: X = K_*x L⁻ᵀ for a chunk of test points, and Σ_c X[s][c]² per test point — the predictive
// variance var* = k** − colsumsq(U⁻ᵀ K_x*) (src/exact_gpr_posterior.jl:68-70, src/util/common_covmat_ops.jl:90) WITHOUT gathering
// the factor onto one device.  L stays where the factorisation left it (block (i, j) on rank (i mod P, j mod Q)); what travels are
// N*×NB blocks of the solution:
//   step k   diagonal owner (k mod P, k mod Q):  X_k = (K(x*, x_k) + Σ_q ACC_q[k]) L_kk⁻ᵀ     (its own accumulator + the Q−1 partial
//                                                sums of its process row), colsumsq(X_k) into its variance partial, X_k published
//            every rank of process column k mod Q: fetch X_k, ACC[rows i > k] −= X_k L_ikᵀ     (ONE MFMA GEMM over its local rows)
// One stream per rank (the main stream); cross-rank dependencies are the generation-numbered events sx[k] (X_k final) and sa[k]
// (accumulators after step k).  First version: no look-ahead — the chain X_k → update of row k+1 → X_k+1 is serial.
// ------------------------------------------------------------------------------------------------
struct SolveDims {
    long n, npad, nblk, NB, ns, nsp;
    int d;
    const long* valid = nullptr;  // real points per block (nullptr: the first n points are real)
    bool rhs = false;             // the right-hand sides are GIVEN (rhs_h: nsp × npad, row-major, padded column layout) instead of K(x*, x)
    long sink_nb2 = 0;            // > 0: rows [t·NB, (t+1)·NB) of every X_k also go to block row nblk + t of the rank's NEW piece (MRank::A2)
    int nbwd = 0;                 // > 0: backward sweeps Z_s = X_s L⁻¹ for the first nbwd rows (one vector sweep each) after the forward pass
    long seq_bwd0 = 0;            // generations of the backward sweeps: seq_bwd0 + 2 s for the barrier before sweep s, + 1 for its events
};

// Rows of X = B L⁻ᵀ on the block-cyclic pieces (B = K(x*, x) or given rows), optionally followed by Z = X L⁻¹ row by row.
int32_t solve_rank(gp_multi* M, MRank* me, const SolveDims& sd, bool dry, const gp_multi_post::Piece* piece, int kind, double variance,
                   const double* x_h /* [d][npad] scaled training inputs */, const double* xs_h /* [d][nsp] scaled test inputs */,
                   const double* rhs_h /* sd.rhs: nsp × npad */, double* vsum_host /* nsp: this rank's partial Σ_c X² */,
                   double* cov_host /* nullable: nsp×nsp partial X Xᵀ (lower) */, double* z_host /* sd.nbwd × npad: Z rows (diagonal owners write their blocks) */,
                   DevBufs* bufs, long seq) {
    const int P = M->P, Q = M->Q, p = me->p, q = me->q, R_ = me->r;
    const long NB = sd.NB, nblk = sd.nblk, npad = sd.npad, nsp = sd.nsp, n = sd.n;
    const long nlb_r = nblk / P, nlb_c = nblk / Q;
    long g_ = P, h_ = Q;
    while (h_) {
        const long t_ = g_ % h_;
        g_ = h_;
        h_ = t_;
    }
    const long lcm = (long)P * Q / g_;
    const long n_own = nblk / lcm;  // diagonal blocks of this rank (one per lcm block columns; 0 when (p, q) never meets the diagonal)
    const bool on_diag = [&]() {
        for (long k = 0; k < lcm; ++k)
            if (k % P == p && k % Q == q) return true;
        return false;
    }();
    Dims dm{n, npad, nblk, NB, nlb_r, nlb_c, NB + 32, sd.d};
    RankRun rr{M, me, dm, seq, dry, 0, M->tr};
    gp_ctx* c = me->c;
    hipStream_t sm = nullptr;
    if (!dry) {
        rr.st[SM] = sm = c->sm;
        MCHK(hipSetDevice(me->device));
        c->ev_used = 0;
        c->gemm_recs.clear();
    }
    const long ldacc = nlb_r * NB + 32, ldx = std::max(1L, n_own) * NB + 32, ldb = NB + 32;
    void *x_v = 0, *xs_v = 0, *acc_v = 0, *xown_v = 0, *xb_v = 0, *t_v = 0, *vs_v = 0, *vt_v = 0, *cv_v = 0, *rhs_v = 0, *bacc_v = 0, *alb_v = 0, *tmp_v = 0;
    const bool want_cov = dry ? !sd.rhs : cov_host != nullptr;
    const long ldcv = nsp + 32;
    if (!dry) {
        if (!sd.rhs) {
            RC(bufs->get(sizeof(double) * (size_t)sd.d * npad, &x_v));
            RC(bufs->get(sizeof(double) * (size_t)sd.d * nsp, &xs_v));
        } else if (on_diag) {
            RC(bufs->get(sizeof(double) * (size_t)nsp * npad, &rhs_v));
        }
        RC(bufs->get(sizeof(double) * (size_t)(nsp + 128) * ldacc, &acc_v));
        RC(bufs->get(sizeof(double) * (size_t)(nsp + 128) * ldx, &xown_v));
        RC(bufs->get(sizeof(double) * (size_t)(nsp + 128) * ldb, &xb_v));
        RC(bufs->get(sizeof(double) * (size_t)(nsp + 128) * ldb, &t_v));
        RC(bufs->get(sizeof(double) * (size_t)nsp, &vs_v));
        RC(bufs->get(sizeof(double) * (size_t)nsp, &vt_v));
        if (want_cov) RC(bufs->get(sizeof(double) * (size_t)(nsp + 128) * ldcv, &cv_v));
        if (sd.nbwd > 0) {
            RC(bufs->get(sizeof(double) * (size_t)(nlb_c * NB + 128), &bacc_v));
            RC(bufs->get(sizeof(double) * (size_t)npad, &alb_v));
            RC(bufs->get(sizeof(double) * (size_t)NB * (P + 1), &tmp_v));
        }
    }
    me->sacc = (double*)acc_v; me->sacc_ld = ldacc;
    me->sxown = (double*)xown_v; me->sxown_ld = ldx;
    me->acc = (double*)bacc_v;
    me->alpha_blk = (double*)alb_v;
    const double* A = piece ? (const double*)piece->A : nullptr;
    const long ld = piece ? piece->ld : 0;
    auto rank_of = [&](int pp, int qq) -> MRank& { return M->ranks[(size_t)pp * Q + qq]; };
    auto addmat = [&](double* dst, long ldd, const double* src, long lds, long rows, long cols) -> int32_t {
        hipLaunchKernelGGL(mk_addmat_kernel, dim3((unsigned)std::min<long>(1024, (rows * cols + 255) / 256)), dim3(256), 0, sm, dst, ldd, src, lds, rows, cols);
        MCHK(hipGetLastError());
        return 0;
    };
    auto valid_of = [&](long k) -> long { return sd.valid ? sd.valid[k] : std::max(0L, std::min(NB, n - k * NB)); };

    RC(rr.op(SM, "solve_init", 0, 0, {}, {Fp{"ACC", R_, 0, 0, 0, nlb_r, 0}, Fp{"Vs", R_, 0, 0, 0, 1, 0}}, [&]() -> int32_t {
        if (x_v) MCHK(hipMemcpyAsync(x_v, x_h, sizeof(double) * (size_t)sd.d * npad, hipMemcpyHostToDevice, sm));
        if (xs_v) MCHK(hipMemcpyAsync(xs_v, xs_h, sizeof(double) * (size_t)sd.d * nsp, hipMemcpyHostToDevice, sm));
        if (rhs_v) MCHK(hipMemcpyAsync(rhs_v, rhs_h, sizeof(double) * (size_t)nsp * npad, hipMemcpyHostToDevice, sm));
        MCHK(hipMemsetAsync(acc_v, 0, sizeof(double) * (size_t)(nsp + 128) * ldacc, sm));
        MCHK(hipMemsetAsync(xown_v, 0, sizeof(double) * (size_t)(nsp + 128) * ldx, sm));
        MCHK(hipMemsetAsync(xb_v, 0, sizeof(double) * (size_t)(nsp + 128) * ldb, sm));
        MCHK(hipMemsetAsync(t_v, 0, sizeof(double) * (size_t)(nsp + 128) * ldb, sm));
        MCHK(hipMemsetAsync(vs_v, 0, sizeof(double) * (size_t)nsp, sm));
        return 0;
    }));
    for (long k = 0; k < nblk; ++k) {
        const int pk = (int)(k % P), qk = (int)(k % Q);
        if (q != qk) continue;  // only the process column of block column k takes part in step k
        const long li = k / P, lc = k / Q, ko = k / lcm;
        const double* xk = nullptr;
        long xk_ld = 0;
        Fp f_xk;
        if (p == pk) {  // diagonal owner
            double* Xk = me->sxown ? me->sxown + ko * NB : nullptr;
            const Fp fX = Fp{"X", R_, 0, 0, k, k + 1, 0};
            const Fp fAcc = Fp{"ACC", R_, 0, 0, li, li + 1, 0};
            if (sd.rhs)
                RC(rr.op(SM, "rhs", k, 0, {}, {fX}, [&]() -> int32_t {
                    MCHK(hipMemcpy2DAsync(Xk, sizeof(double) * ldx, (const double*)rhs_v + k * NB, sizeof(double) * npad, sizeof(double) * NB, nsp,
                                          hipMemcpyDeviceToDevice, sm));
                    return 0;
                }));
            else
                RC(rr.op(SM, "kstar", k, 0, {}, {fX}, [&]() {
                    return eng_kcross(c, sm, kind, variance, (const double*)xs_v, nsp, sd.ns, nsp, (const double*)x_v + k * NB, npad, valid_of(k), NB, sd.d, Xk,
                                      ldx);
                }));
            RC(rr.op(SM, "add_own", k, 0, {fAcc, fX}, {fX}, [&]() { return addmat(Xk, ldx, me->sacc + li * NB, ldacc, nsp, NB); }));
            for (int qq = 0; qq < Q; ++qq) {  // + the partial sums of the other ranks of my process row (their columns j < k, j ≡ qq mod Q)
                if (qq == qk) continue;
                long jl = -1;
                for (long j = k - 1; j >= 0; --j)
                    if ((int)(j % Q) == qq) {
                        jl = j;
                        break;
                    }
                if (jl < 0) continue;  // that rank holds no column left of k: its accumulator is still zero
                MRank& src = rank_of(pk, qq);
                const Fp fT = Fp{"T", R_, 0, 0, 0, 1, 0};
                RC(rr.await(src.sa[jl], SM));
                RC(rr.op(SM, "pull_acc", k, qq, {Fp{"ACC", src.r, 0, 0, li, li + 1, 0}}, {fT},
                         [&]() { return rr.pull(SM, (double*)t_v, ldb, src.sacc + li * NB, src.sacc_ld, nsp, NB); }));
                RC(rr.op(SM, "add_t", k, qq, {fT, fX}, {fX}, [&]() { return addmat(Xk, ldx, (const double*)t_v, ldb, nsp, NB); }));
            }
            RC(rr.op(SM, "trsm", k, 0, {Fp{"A", R_, li, li + 1, lc, lc + 1, 0}, fX}, {fX},
                     [&]() { return eng_trsm(c, sm, Xk, ldx, nsp, A + li * NB * ld + lc * NB, ld, NB); }));
            RC(rr.op(SM, "colsq", k, 0, {fX, Fp{"Vs", R_, 0, 0, 0, 1, 0}}, {Fp{"Vs", R_, 0, 0, 0, 1, 0}}, [&]() -> int32_t {
                RC(eng_rowsumsq(c, sm, Xk, ldx, nsp, NB, (double*)vt_v));
                return eng_add_vec(c, sm, (double*)vs_v, (const double*)vt_v, nsp);
            }));
            RC(rr.publish(me->sx[k], SM));
            xk = Xk;
            xk_ld = ldx;
            f_xk = fX;
        } else {
            MRank& own = rank_of(pk, qk);
            const Fp fB = Fp{"Xb", R_, 0, 0, 0, 1, 0};
            RC(rr.await(own.sx[k], SM));
            RC(rr.op(SM, "pull_x", k, 0, {Fp{"X", own.r, 0, 0, k, k + 1, 0}}, {fB},
                     [&]() { return rr.pull(SM, (double*)xb_v, ldb, own.sxown + ko * NB, own.sxown_ld, nsp, NB); }));
            xk = (const double*)xb_v;
            xk_ld = ldb;
            f_xk = fB;
        }
        // sequential update: the rows of X_k that form block (nblk + t, k) of the extended factor stay on this rank's new piece
        for (long t = 0; t < sd.sink_nb2; ++t) {
            const long I = nblk + t;
            if ((int)(I % P) != p) continue;
            const long rows = std::min(NB, nsp - t * NB);
            if (rows <= 0) continue;
            RC(rr.op(SM, "sink", k, t, {f_xk}, {Fp{"A2", R_, I / P, I / P + 1, lc, lc + 1, 0}}, [&]() -> int32_t {
                MCHK(hipMemcpy2DAsync(me->A2 + (I / P) * NB * me->ld2 + lc * NB, sizeof(double) * me->ld2, xk + t * NB * xk_ld, sizeof(double) * xk_ld,
                                      sizeof(double) * NB, rows, hipMemcpyDeviceToDevice, sm));
                return 0;
            }));
        }
        // accumulators of my block rows below k:  ACC[:, rows i > k] −= X_k · L[rows i > k, block column k]ᵀ
        const long lr0 = nlb_before(k, p, P);  // first local block row with global block > k
        if (lr0 < nlb_r) {
            const Fp fAcc = Fp{"ACC", R_, 0, 0, lr0, nlb_r, 0};
            RC(rr.op(SM, "gemm", k, 0, {f_xk, Fp{"A", R_, lr0, nlb_r, lc, lc + 1, 0}, fAcc}, {fAcc}, [&]() {
                return eng_gemm_nt(c, sm, me->sacc + lr0 * NB, ldacc, xk, xk_ld, A + lr0 * NB * ld + lc * NB, ld, nsp, (nlb_r - lr0) * NB, NB,
                                   plain_map(0, 0, 0));
            }));
        }
        RC(rr.publish(me->sa[k], SM));
    }
    // full covariance: cov* = K** − X Xᵀ, and X Xᵀ = Σ_k X_k X_kᵀ is a sum over the diagonal owners' blocks — ONE MFMA SYRK per rank
    // over the solution blocks it owns (lower triangle), summed on the host
    if (want_cov && on_diag)
        RC(rr.op(SM, "syrk", 0, 0, {Fp{"X", R_, 0, 0, 0, nblk, 0}}, {Fp{"Cv", R_, 0, 0, 0, 1, 0}}, [&]() -> int32_t {
            MCHK(hipMemsetAsync(cv_v, 0, sizeof(double) * (size_t)(nsp + 128) * ldcv, sm));
            return eng_gemm_nt(c, sm, (double*)cv_v, ldcv, me->sxown, ldx, me->sxown, ldx, nsp, nsp, n_own * NB, plain_map(1, 0, 0));
        }));

    // ---- backward sweeps  Z_s = X_s L⁻¹  (row s of the forward result as the right-hand side; one vector sweep per row, the block
    // sweep of fit_rank: partial sums per local column, reduced over the process rows by the diagonal owner).  Between two sweeps
    // every rank waits for every other rank's stream ("bar"): the partial sums and solution blocks of sweep s are pulled by peers.
    for (int s = 0; s < sd.nbwd; ++s) {
        double* acc = me->acc;
        double* tmp = (double*)tmp_v;
        if (s > 0 && M->R > 1) {  // (two events in turn: a rank may reach the next barrier before a slow peer has issued its wait for this one)
            rr.seq = sd.seq_bwd0 + 2 * s;
            RC(rr.publish(me->bar[s & 1], SM));
            for (int r2 = 0; r2 < M->R; ++r2)
                if (r2 != R_) RC(rr.await(M->ranks[(size_t)r2].bar[s & 1], SM));
        }
        rr.seq = sd.seq_bwd0 + 2 * s + 1;
        RC(rr.op(SM, "bwd_init", s, 0, {}, {Fp{"acc", R_, 0, 0, 0, nlb_c, 0}}, [&]() -> int32_t {
            MCHK(hipMemsetAsync(acc, 0, sizeof(double) * (size_t)(nlb_c * NB + 128), sm));
            return 0;
        }));
        for (long k = nblk - 1; k >= 0; --k) {
            const int pk = (int)(k % P), qk = (int)(k % Q);
            const long c0 = (k / Q) * NB, lc = k / Q, ko = k / lcm;
            const Fp f_acc = Fp{"acc", R_, 0, 0, lc, lc + 1, 0};
            const Fp f_alb = Fp{"alb", R_, 0, 0, k, k + 1, 0};
            if (q == qk) {
                if (p == pk) {
                    double* ak = me->alpha_blk ? me->alpha_blk + k * NB : nullptr;
                    RC(rr.op(SM, "ak", k, s, {f_acc, Fp{"X", R_, 0, 0, k, k + 1, 0}}, {f_alb}, [&]() -> int32_t {
                        MCHK(hipMemcpyAsync(ak, acc + c0, sizeof(double) * NB, hipMemcpyDeviceToDevice, sm));
                        return eng_add_vec(c, sm, ak, me->sxown + ko * NB + (long)s * ldx, NB);  // + row s of X_k
                    }));
                    for (int pp = 0; pp < P; ++pp) {  // + the partial sums of the other process rows
                        if (pp == pk) continue;
                        MRank& src = rank_of(pp, qk);
                        const Fp f_tmp = Fp{"tmp", R_, 0, 0, pp, pp + 1, 0};
                        RC(rr.await(src.accr[k], SM));
                        RC(rr.op(SM, "pull_acc", k, pp, {Fp{"acc", src.r, 0, 0, lc, lc + 1, 0}}, {f_tmp}, [&]() { return rr.pull(SM, tmp + (size_t)pp * NB, NB, src.acc + c0, NB, 1, NB); }));
                        RC(rr.op(SM, "add_acc", k, pp, {f_tmp, f_alb}, {f_alb}, [&]() { return eng_add_vec(c, sm, ak, tmp + (size_t)pp * NB, NB); }));
                    }
                    RC(rr.op(SM, "trsv", k, s, {Fp{"A", R_, k / P, k / P + 1, lc, lc + 1, 0}, f_alb}, {f_alb}, [&]() -> int32_t {
                        RC(eng_trsv(c, sm, A + (k / P) * NB * ld + c0, ld, NB, ak, NB, 1, false));
                        MCHK(hipMemcpyAsync(z_host + (size_t)s * npad + k * NB, ak, sizeof(double) * NB, hipMemcpyDeviceToHost, sm));
                        return 0;
                    }));
                    RC(rr.publish(me->alr[k], SM));
                } else {
                    RC(rr.publish(me->accr[k], SM));
                }
            }
            if (p == pk && k > 0) {  // my block row k: acc_j −= L[k][j]ᵀ z_k for my local columns j < k
                const long ncb = nlb_before(k - 1, q, Q);
                if (ncb > 0) {
                    const double* ak;
                    Fp f_ak = f_alb;
                    if (q == qk) {
                        ak = me->alpha_blk ? me->alpha_blk + k * NB : nullptr;
                    } else {
                        MRank& own = rank_of(pk, qk);
                        double* dst = tmp + (size_t)P * NB;
                        f_ak = Fp{"tmp", R_, 0, 0, P, P + 1, 0};
                        RC(rr.await(own.alr[k], SM));
                        RC(rr.op(SM, "pull_alpha", k, 0, {Fp{"alb", own.r, 0, 0, k, k + 1, 0}}, {f_ak}, [&]() { return rr.pull(SM, dst, NB, own.alpha_blk + k * NB, NB, 1, NB); }));
                        ak = dst;
                    }
                    const Fp f_accs = Fp{"acc", R_, 0, 0, 0, ncb, 0};
                    RC(rr.op(SM, "gemv", k, s, {Fp{"A", R_, k / P, k / P + 1, 0, ncb, 0}, f_ak, f_accs}, {f_accs},
                             [&]() { return eng_gemv_t(c, sm, A + (k / P) * NB * ld, ld, NB, ncb * NB, ak, acc); }));
                }
            }
        }
    }
    if (dry) return 0;
    MCHK(hipMemcpyAsync(vsum_host, vs_v, sizeof(double) * (size_t)nsp, hipMemcpyDeviceToHost, sm));
    if (cov_host) {
        if (on_diag) MCHK(hipMemcpy2DAsync(cov_host, sizeof(double) * nsp, cv_v, sizeof(double) * ldcv, sizeof(double) * nsp, nsp, hipMemcpyDeviceToHost, sm));
        else memset(cov_host, 0, sizeof(double) * (size_t)nsp * nsp);
    }
    RC(rr.drain(SM));  // (the caller releases the buffers after EVERY rank thread has drained its stream: peers' pulls included)
    return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// entry points (engine.hpp)
// ------------------------------------------------------------------------------------------------
void multi_destroy(gp_multi* m) {
    if (!m) return;
    for (auto& rk : m->ranks) {
        (void)hipSetDevice(rk.device);
        if (rk.comm) {
            std::lock_guard<std::mutex> l(g_rccl_mu);
            if (g_rccl.ok && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(rk.comm);
            if (g_rccl.live > 0) --g_rccl.live;
            rk.comm = nullptr;
        }
        for (auto* v : {&rk.ready, &rk.lkk, &rk.accr, &rk.alr, &rk.sx, &rk.sa, &rk.bar})
            for (auto& x : *v)
                if (x.ev) (void)hipEventDestroy(x.ev);
        for (auto e : rk.own) (void)hipEventDestroy(e);
        if (rk.sc) (void)hipStreamDestroy(rk.sc);
        for (hipStream_t st : {rk.sd, rk.sm_m, rk.sp_m})
            if (st) (void)hipStreamDestroy(st);
        if (rk.c) (void)gp_ctx_destroy(rk.c);
    }
    delete m;
}

// "multi_chain_cus" = r: (re)create the masked stream triple of every rank — sd on CUs [0, r), sm_m / sp_m on [r, num_cus) — or drop it (r = 0).
// Mask bit i is CU i / 8 of XCC i % 8 (profiles/r2/traces/cumask_probe.txt), so [0, r) takes r / 8 CUs of every XCD; r is rounded to a multiple of 8
// and kept within [8, num_cus / 2].
static int32_t multi_set_chain_cus(gp_multi* m, int r) {
    for (auto& rk : m->ranks) {
        if (!rk.c) continue;
        MCHK(hipSetDevice(rk.device));
        for (hipStream_t* ps : {&rk.sd, &rk.sm_m, &rk.sp_m})
            if (*ps) {
                (void)hipStreamSynchronize(*ps);
                (void)hipStreamDestroy(*ps);
                *ps = nullptr;
            }
    }
    m->chain_cus = 0;
    if (r <= 0) return 0;
    for (auto& rk : m->ranks) {
        if (!rk.c) continue;
        const int ncu = rk.c->num_cus > 0 ? rk.c->num_cus : 256;
        const int rr = std::min(std::max(8, (r + 7) / 8 * 8), ncu / 2);
        const uint32_t words = (uint32_t)((ncu + 31) / 32);
        std::vector<uint32_t> lo(words, 0u), hi(words, 0u);
        for (int i = 0; i < ncu; ++i) (i < rr ? lo : hi)[(size_t)i / 32] |= 1u << (i % 32);
        MCHK(hipSetDevice(rk.device));
        MCHK(hipExtStreamCreateWithCUMask(&rk.sd, words, lo.data()));
        MCHK(hipExtStreamCreateWithCUMask(&rk.sm_m, words, hi.data()));
        MCHK(hipExtStreamCreateWithCUMask(&rk.sp_m, words, hi.data()));
        for (hipStream_t st : {rk.sd, rk.sm_m, rk.sp_m}) RC(ctx_prime_stream(rk.c, st));  // the hardware queues exist before the first fit (see gp_ctx_create_multi)
        m->chain_cus = rr;
    }
    return 0;
}

int32_t multi_set_param(gp_ctx* c, const char* name, int64_t v) {
    gp_multi* m = c->multi;
    if (!m) return 1;
    if (!strcmp(name, "lookahead_depth")) {
        m->depth = (int)std::min<int64_t>(3, std::max<int64_t>(1, v));
        return 0;
    }
    if (!strcmp(name, "copy_kernel")) {
        m->copy_kernel = v != 0;
        return 0;
    }
    if (!strcmp(name, "multi_debug_sync")) {
        m->debug_sync = (int)v;
        return 0;
    }
    if (!strcmp(name, "multi_check")) {
        m->check = (int)v;
        return 0;
    }
    if (!strcmp(name, "multi_dist_predict")) {
        m->dist_predict = v != 0;
        return 0;
    }
    if (!strcmp(name, "multi_inject_fault")) {
        m->inject_fault = v != 0;
        return 0;
    }
    if (!strcmp(name, "multi_verify")) {
        m->verify = v != 0;
        return 0;
    }
    if (!strcmp(name, "multi_window")) {
        m->window = std::max<int64_t>(4, v);
        return 0;
    }
    if (!strcmp(name, "multi_trsm_inv")) {
        m->trsm_inv = v <= 0 ? 0 : (v >= 2 ? 2 : 1);
        return 0;
    }
    if (!strcmp(name, "multi_chain_cus")) return multi_set_chain_cus(m, (int)v);
    if (!strcmp(name, "multi_timeout_s")) {
        m->timeout_s = (double)std::max<int64_t>(1, v);
        return 0;
    }
    if (!strcmp(name, "dist_nb")) {
        m->nb = std::max<int64_t>(128, (v + 127) / 128 * 128);
        return 0;
    }
    if (!strcmp(name, "multi_gemm_streamk")) {  // stream-K cuts of the few-tile GEMMs inside the rank contexts (see gp_ctx_create_multi)
        for (auto& rk : m->ranks) (void)gp_ctx_set_param(rk.c, "gemm_streamk", v);
        return 0;
    }
    if (!strcmp(name, "multi_leaf_cols")) {  // columns per register-resident leaf inside the rank contexts (64: co-resident with the bulk update; 128)
        for (auto& rk : m->ranks) (void)gp_ctx_set_param(rk.c, "leaf_cols", v);
        return 0;
    }
    if (!strcmp(name, "gemm_streamk") || !strcmp(name, "leaf_cols")) return 1;  // the main ctx only (everything that runs on devices[0] alone)
    // every other parameter also goes to the rank contexts (kernel variants, timing switches)
    for (auto& rk : m->ranks) (void)gp_ctx_set_param(rk.c, name, v);
    return 1;
}

// read-back of the multi-device parameters (gp_ctx_get_param on a multi-device ctx); 1 = not a multi parameter
int32_t multi_get_param(gp_ctx* c, const char* name, int64_t* out) {
    gp_multi* m = c->multi;
    if (!m) return 1;
    int64_t rk_sk = 0, rk_lc = 0;
    if (!m->ranks.empty() && m->ranks[0].c) {
        rk_sk = m->ranks[0].c->gemm_streamk;
        rk_lc = m->ranks[0].c->leaf_cols;
    }
    const struct { const char* n; int64_t v; } tab[] = {
        {"lookahead_depth", m->depth}, {"copy_kernel", m->copy_kernel}, {"multi_debug_sync", m->debug_sync}, {"multi_check", m->check},
        {"multi_dist_predict", m->dist_predict}, {"multi_inject_fault", m->inject_fault}, {"multi_verify", m->verify},
        {"multi_window", (int64_t)m->window}, {"multi_trsm_inv", m->trsm_inv}, {"multi_chain_cus", m->chain_cus}, {"multi_timeout_s", (int64_t)m->timeout_s}, {"dist_nb", (int64_t)m->nb},
        {"multi_gemm_streamk", rk_sk}, {"multi_leaf_cols", rk_lc}};
    for (const auto& e : tab)
        if (!strcmp(name, e.n)) {
            *out = e.v;
            return 0;
        }
    return 1;
}

// drop the cached device blocks of the rank contexts (gp_ctx_trim on a multi-device ctx)
void multi_trim(gp_ctx* c) {
    gp_multi* m = c->multi;
    if (!m) return;
    for (auto& rk : m->ranks) (void)gp_ctx_trim(rk.c);
    (void)hipSetDevice(c->device);
}

extern "C" int32_t gp_ctx_create_multi(gp_ctx** out, const int32_t* devices, int32_t ndev, int32_t P, int32_t Q, int32_t nb) {
    if (!out) return set_arg_err(1, "out is NULL");
    *out = nullptr;
    if (!devices || ndev < 1) return set_arg_err(2, "devices / ndev");
    if (P <= 0 && Q <= 0) {
        P = ndev;  // full-mesh point-to-point fabric: as many distinct peers per exchange as possible (see the header comment)
        Q = 1;
    } else if (P <= 0) {
        P = ndev / Q;
    } else if (Q <= 0) {
        Q = ndev / P;
    }
    if (P * Q != ndev) return set_arg_err(4, "P * Q must equal ndev");
    if (nb <= 0) nb = 1024;
    if (nb % 128) return set_arg_err(6, "nb must be a multiple of 128");
    int devcount = 0;
    MCHK(hipGetDeviceCount(&devcount));
    bool dup = false;
    for (int i = 0; i < ndev; ++i) {
        if (devices[i] < 0 || devices[i] >= devcount) return set_arg_err(2, "no such device (fewer GPUs visible than requested)");
        for (int j = 0; j < i; ++j) dup = dup || devices[i] == devices[j];
    }
    gp_ctx* main_ctx = nullptr;
    RC(gp_ctx_create(&main_ctx, devices[0], nullptr));
    gp_multi* m = new gp_multi();
    m->P = P; m->Q = Q; m->R = ndev; m->nb = nb; m->virt = dup;
    if (const char* ts = getenv("GPMI_MULTI_TIMEOUT_S")) m->timeout_s = std::max(1.0, atof(ts));
    if (const char* cs = getenv("GPMI_MULTI_CHECK")) m->check = atoi(cs);
    m->ranks.resize((size_t)ndev);
    int32_t rc = 0;
    for (int r = 0; r < ndev && rc == 0; ++r) {
        MRank& rk = m->ranks[r];
        rk.r = r; rk.p = r / Q; rk.q = r % Q; rk.device = devices[r];
        rc = gp_ctx_create(&rk.c, devices[r], nullptr);
        if (rc != 0) break;
        // Rank contexts: hardware-dispatched GEMMs unless "multi_gemm_streamk" / GPMI_MULTI_SK=1 asks for the stream-K cuts, comm
        // stream of the LOWEST priority unless GPMI_COMM_PRIO=1 (the first-fit item of round 2).
        const char* ske = getenv("GPMI_MULTI_SK");
        (void)gp_ctx_set_param(rk.c, "gemm_streamk", (ske && ske[0] == '1') ? 1 : 0);
        // 64-column leaves in the rank contexts ("multi_leaf_cols"): the diagonal block of the look-ahead panel is factored on the panel stream
        // WHILE the bulk update runs on the main stream.  The 128-column leaf needs 152 KB of LDS, i.e. an empty CU, and a running tile GEMM
        // refills every workgroup slot it frees: each of a block's leaves would wait for the end of a GEMM launch (measured on one GPU,
        // profiles/r4/traces/c3_la1_summary.txt: the first leaf of a panel "runs" for the whole 25 ms of the update beside it) — more launch
        // boundaries than the look-ahead depth covers.  The 64-column leaf (46 KB, 226 VGPRs) fits beside one GEMM workgroup and starts at once.
        (void)gp_ctx_set_param(rk.c, "leaf_cols", 64);
        int plo = 0, phi = 0;
        const char* pe = getenv("GPMI_COMM_PRIO");
        if (hipSetDevice(devices[r]) != hipSuccess || hipDeviceGetStreamPriorityRange(&plo, &phi) != hipSuccess ||
            hipStreamCreateWithPriority(&rk.sc, hipStreamNonBlocking, (pe && pe[0] == '1') ? phi : plo) != hipSuccess) {
            rc = set_err_text(-1997, "could not create the comm stream of a rank");
            break;
        }
        for (int j = 0; j < ndev; ++j) {  // peer access for the direct copies (errors: already enabled / same device — ignored)
            if (devices[j] != devices[r]) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devices[r], devices[j]) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(devices[j], 0);
                (void)hipGetLastError();
            }
        }
    }
    // every lazily created resource of the rank contexts — workspaces and, above all, the hardware queues behind the 3·R streams —
    // exists before the first fit (GPMI_MULTI_PRIME=0 leaves them lazy: the configuration in which first fits went wrong)
    {
        const char* pe = getenv("GPMI_MULTI_PRIME");
        if (rc == 0 && !(pe && pe[0] == '0')) {
            for (int r = 0; r < ndev && rc == 0; ++r) {
                rc = ctx_prime(m->ranks[r].c, nb);
                if (rc == 0) rc = ctx_prime_stream(m->ranks[r].c, m->ranks[r].sc);
            }
            if (rc == 0) rc = ctx_prime(main_ctx, nb);
            for (int r = 0; r < ndev && rc == 0; ++r)
                if (hipSetDevice(devices[r]) != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = set_err_text(-1997, "device synchronisation after priming failed");
        }
    }
    // transport: RCCL between distinct devices unless told otherwise (GPMI_COMM=rccl|p2p); virtual ranks can only copy — unless a
    // stand-in library that accepts duplicate devices is named (GPMI_RCCL_LIB, tests/rccl_mock)
    const char* want = getenv("GPMI_COMM");
    m->comm = 2;
    const bool force_rccl = want && !strcmp(want, "rccl");  // also with ONE device: exercises dlopen + ncclCommInitAll (API check)
    const bool standin = getenv("GPMI_RCCL_LIB") != nullptr;
    if (rc == 0 && (!dup || (force_rccl && standin)) && (ndev > 1 || force_rccl) && !(want && !strcmp(want, "p2p"))) {
        std::lock_guard<std::mutex> l(g_rccl_mu);
        if (g_rccl.load()) {
            std::vector<ncclComm_t_> comms((size_t)ndev, nullptr);
            std::vector<int> devs(devices, devices + ndev);
            const int nrc = g_rccl.CommInitAll(comms.data(), ndev, devs.data());
            if (nrc == 0) {
                for (int r = 0; r < ndev; ++r) m->ranks[r].comm = comms[r];
                g_rccl.live += ndev;
                m->comm = 1;
                m->comm_note = standin ? "grouped send/recv through the stand-in library of GPMI_RCCL_LIB" : "RCCL grouped send/recv over xGMI (ncclCommInitAll)";
            } else {
                m->comm_note = std::string("peer copies (ncclCommInitAll failed: ") + g_rccl.GetErrorString(nrc) + ")";
            }
        } else {
            m->comm_note = "peer copies (" + g_rccl.err + ")";
        }
        if (force_rccl && m->comm != 1) rc = set_err_text(-1996, "GPMI_COMM=rccl but RCCL is unavailable: " + m->comm_note);
    } else if (rc == 0) {
        m->comm_note = dup ? "same-device copies (virtual ranks)" : (ndev > 1 ? "peer copies (GPMI_COMM=p2p)" : "single rank");
    }
    if (rc != 0) {
        multi_destroy(m);
        (void)gp_ctx_destroy(main_ctx);
        return rc;
    }
    (void)hipSetDevice(devices[0]);
    m->pool.reset(new RankPool(m->R));  // the rank threads exist with the context (run_ranks' lazy creation is for the dry-run traces on a stack gp_multi)
    main_ctx->multi = m;
    *out = main_ctx;
    return 0;
}

extern "C" int32_t gp_ctx_multi_info(gp_ctx* c, int32_t* P, int32_t* Q, int32_t* nb, int32_t* comm, int32_t* depth) {
    Guard gd(c);
    if (!gd.ok) return set_arg_err(1, "not a live gp_ctx");
    gp_multi* m = c->multi;
    if (P) *P = m ? m->P : 1;
    if (Q) *Q = m ? m->Q : 1;
    if (nb) *nb = m ? (int32_t)m->nb : 0;
    if (comm) *comm = m ? m->comm : 0;
    if (depth) *depth = m ? m->depth : 0;
    return 0;
}

extern "C" int32_t gp_ctx_multi_stats(gp_ctx* c, int64_t* fits, int64_t* retries, int64_t* solves) {
    Guard gd(c);
    if (!gd.ok) return set_arg_err(1, "not a live gp_ctx");
    gp_multi* m = c->multi;
    if (fits) *fits = m ? m->fits : 0;
    if (retries) *retries = m ? m->retries : 0;
    if (solves) *solves = m ? m->solves : 0;
    return 0;
}

static long lcm_of(long a0, long b0) {
    long a = a0, b = b0;
    while (b) {
        const long t = a % b;
        a = b;
        b = t;
    }
    return a0 / a * b0;
}

// The schedule of a P×Q fit over nblk block columns as fit_rank issues it — the SAME control flow, run by one host thread per
// rank without any device (operations, event records / waits, transfers with their block footprints as JSON lines):
// what tools/multi_schedule_check.py checks on a machine without a GPU.  comm: 1 = RCCL-style send/recv, 2 = copies.
extern "C" int32_t gp_multi_schedule_trace(int32_t P, int32_t Q, int32_t nblk_in, int32_t depth, int32_t comm, const char* path) {
    if (P < 1 || Q < 1 || P * Q > 64) return set_arg_err(1, "P, Q");
    if (nblk_in < 1 || nblk_in > 4096) return set_arg_err(3, "nblk");
    if (depth < 1 || depth > 3) return set_arg_err(4, "depth must be 1..3");
    if ((comm & 15) != 1 && (comm & 15) != 2) return set_arg_err(5, "comm must be 1 (send/recv) or 2 (copies), + 16 for the substitution solve, + 32 for the chain stream");
    if (!path) return set_arg_err(6, "path is NULL");
    Trace tr;
    tr.f = fopen(path, "w");
    if (!tr.f) return set_arg_err(6, "cannot open the trace file");
    gp_multi M;
    M.P = P; M.Q = Q; M.R = P * Q; M.nb = 128; M.depth = depth; M.comm = comm & 15; M.tr = &tr; M.timeout_s = 60;
    M.use_inv = !(comm & 16);  // comm + 16: the schedule with the substitution solve ("multi_trsm_inv" = 0: L_kk itself travels)
    M.dry_chain = (comm & 32) != 0;  // comm + 32: the schedule with the diagonal chain on its own stream ("multi_chain_cus" > 0)
    const int chain_flag = M.dry_chain ? 1 : 0;
    comm &= 15;
    const long lcm = lcm_of(P, Q);
    const long nblk = (nblk_in + lcm - 1) / lcm * lcm;
    Dims dm{nblk * 128, nblk * 128, nblk, 128, nblk / P, nblk / Q, 128 + 32, 1};
    M.ranks.resize((size_t)M.R);
    for (int r = 0; r < M.R; ++r) {
        MRank& rk = M.ranks[r];
        rk.r = r; rk.p = r / Q; rk.q = r % Q;
    }
    if (const int32_t erc = size_xevents(&M, nblk, true)) {
        fclose(tr.f);
        return erc;
    }
    {
        char b[160];
        snprintf(b, sizeof b, "{\"t\":\"hdr\",\"P\":%d,\"Q\":%d,\"nblk\":%ld,\"depth\":%d,\"comm\":%d,\"inv\":%d,\"chain\":%d,\"dry\":1}", P, Q, nblk, depth, comm, (int)M.use_inv, chain_flag);
        tr.line(b);
    }
    std::vector<std::thread> th;
    for (int r = 0; r < M.R; ++r)
        th.emplace_back([&, r]() {
            MRank& rk = M.ranks[r];
            rk.rc = fit_rank(&M, &rk, dm, true, 0, 1.0, nullptr, nullptr, nullptr, 1, true, false, nullptr, nullptr, nullptr, nullptr, 1);
            if (rk.rc != 0) {
                rk.err = gp_last_error();
                M.abort.store(1);
            }
        });
    for (auto& t : th) t.join();
    fclose(tr.f);
    tr.f = nullptr;
    for (int r = 0; r < M.R; ++r)
        if (M.ranks[r].rc != 0 && M.ranks[r].rc != -1999) return set_err_text(M.ranks[r].rc, "rank " + std::to_string(r) + ": " + M.ranks[r].err);
    return 0;
}

// fit on a multi-device ctx (called with the main ctx locked).  fp64, at most RHS_ROWS right-hand sides (the caller routes
// everything else to the single-device engine on devices[0]).
int32_t multi_fit(gp_ctx* c, const gp_kernel* k, const gp_points* x, const gp_noise* noise, const void* mean_or_null, const void* Yv,
                  long ldy, int ncols, double* logpdf_out, double* terms_out, gp_post* post, void* alpha_out) {
    gp_multi* M = c->multi;
    if (k->dtype != 0) return set_arg_err(2, "multi-device fits compute in fp64 (kernel dtype must be 0)");
    if (ncols > RHS_ROWS) return set_arg_err(8, "at most 128 right-hand sides per multi-device fit");
    const int P = M->P, Q = M->Q, R = M->R;
    const long n = x->n, NB = M->nb;
    const int d = x->d;
    const long lcm = lcm_of(P, Q);
    long nblk = (n + NB - 1) / NB;
    nblk = (nblk + lcm - 1) / lcm * lcm;  // every rank owns the same number of block rows / columns
    Dims dm{n, nblk * NB, nblk, NB, nblk / P, nblk / Q, NB + 32, d};
    const long npad = dm.npad;
    const double* Y = (const double*)Yv;
    const double* mean = (const double*)mean_or_null;

    // ---- host marshalling (once, shared read-only by the rank threads)
    std::vector<double> xs_h((size_t)d * npad, 0.0), noise_h((size_t)npad, 0.0), rhs_h((size_t)ncols * npad, 0.0);
    auto pt = [&](long i, int dd) -> double {
        const double* pd = (const double*)x->data;
        return x->layout == 0 ? pd[i] : (x->layout == 1 ? pd[(long)dd + i * x->d] : pd[i + (long)dd * x->n]);
    };
    for (int dd = 0; dd < d; ++dd) {
        const double s = k->nscale == 1 ? k->scale[0] : (k->nscale > 1 ? k->scale[dd] : 1.0);
        for (long i = 0; i < n; ++i) xs_h[(size_t)dd * npad + i] = s * pt(i, dd);
    }
    for (long i = 0; i < n; ++i) noise_h[i] = noise->kind == 0 ? noise->s : ((const double*)noise->diag)[i];
    {
        // "multi_trsm_inv": a product with an explicit inverse carries an error of order cond(L_kk)·ε where substitution is backward stable (the guard of the
        // single-device inverse-block solves, gpmi355.hip trsm_cached, reads the factor's diagonal; here the decision must precede the fit).  Every pivot
        // of K + Σy lies in [min Σy_ii, variance + max Σy_ii] (Schur complements of K are positive semi-definite and below K), so
        // max L_ii / min L_ii <= sqrt((variance + max Σy) / min Σy): beyond 1e5 this fit keeps the substitution recursion.
        double lo = noise_h[0], hi = noise_h[0];
        for (long i = 1; i < n; ++i) {
            lo = std::min(lo, noise_h[i]);
            hi = std::max(hi, noise_h[i]);
        }
        M->use_inv = M->trsm_inv != 0 && lo > 0 && std::sqrt((k->variance + hi) / lo) <= 1e5;
    }
    for (int s = 0; s < ncols; ++s)
        for (long i = 0; i < n; ++i) rhs_h[(size_t)s * npad + i] = Y[(size_t)s * ldy + i] - (mean ? mean[i] : 0.0);
    double* alpha_pin = nullptr;
    MCHK(hipSetDevice(c->device));
    MCHK(hipHostMalloc((void**)&alpha_pin, sizeof(double) * (size_t)npad, hipHostMallocPortable));
    memset(alpha_pin, 0, sizeof(double) * (size_t)npad);

    {
        const int32_t erc = size_xevents(M, nblk, false);
        if (erc != 0) {
            (void)hipHostFree(alpha_pin);
            return erc;
        }
    }
    (void)hipSetDevice(c->device);
    const bool keep = post != nullptr;

    Trace tr;
    if (const char* tp = getenv("GPMI_TRACE_SCHEDULE")) {
        tr.f = fopen(tp, "w");  // the LAST fit of the process is what the file holds
        if (tr.f) {
            char b[160];
            snprintf(b, sizeof b, "{\"t\":\"hdr\",\"P\":%d,\"Q\":%d,\"nblk\":%ld,\"depth\":%d,\"comm\":%d,\"inv\":%d,\"chain\":%d,\"dry\":0}", P, Q, nblk, M->depth, M->comm, (int)M->use_inv, M->chain_cus > 0 ? 1 : 0);
            tr.line(b);
        }
    }
    M->tr = tr.f ? &tr : nullptr;

    std::vector<std::vector<double>> scal((size_t)R, std::vector<double>(8 + RHS_ROWS, 0.0));
    std::vector<int> infos((size_t)R, 0);
    std::vector<std::unique_ptr<DevBufs>> bufs((size_t)R);
    auto release_bufs = [&]() {  // everything of the attempt goes back to the rank caches — the factor pieces a successful rank thread
        for (int r = 0; r < R; ++r) {  // had already taken out of its DevBufs for the posterior handle included
            MRank& rk = M->ranks[r];
            std::lock_guard<std::mutex> l(rk.c->mu);
            (void)hipSetDevice(rk.device);
            if (keep && rk.rc == 0 && rk.A) ctx_release(rk.c, rk.A, 0);
            rk.A = nullptr;
            bufs[r].reset();
        }
        (void)hipSetDevice(c->device);
    };
    // "multi_verify" (default on): the result is checked before it is handed out — δᵀα against ‖L⁻¹δ‖² (forward against backward
    // solve) and (K + Σy)α = δ on every row, K·α recomputed from the inputs on the devices — and the fit is repeated ONCE when the
    // check (or the factorisation: a spurious non-positive pivot) fails.  Why: with 18+ streams of several rank threads on ONE
    // device, kernels occasionally ran with stale arguments on this ROCm stack (profiles/r3/first_fit.md; 20 % of
    // the first fits of fresh 8-rank contexts, 2.5 % with HIP_FORCE_DEV_KERNARG=0, ~1 % once every queue exists before the fit).
    const bool verify = M->verify != 0;
    const bool want_alpha = keep || verify;
    auto verify_result = [&](double zz, std::string& why) -> bool {
        double da = 0, dn = 0;
        for (long i = 0; i < n; ++i) {
            da += rhs_h[i] * alpha_pin[i];
            dn += std::fabs(rhs_h[i] * alpha_pin[i]);
        }
        if (!(std::fabs(da - zz) <= 1e-7 * (dn + std::fabs(zz)))) {
            char b[160];
            snprintf(b, sizeof b, "delta'alpha = %.15g but ||L^-1 delta||^2 = %.15g", da, zz);
            why = b;
            return false;
        }
        // (K + Σy) α = δ on EVERY row: a wrong tile perturbs the factored matrix in one block only, and the residual is then confined
        // to that block's rows and columns (a sample of rows missed 3 of 27 corrupted fits, profiles/r3/first_fit.md) — so every rank
        // evaluates its share of the rows of K·α on its device straight from the inputs (kvec: no matrix involved)
        std::vector<double> Ka((size_t)n, 0.0);
        std::vector<int32_t> vrc((size_t)R, 0);
        {
            M->run_ranks([&](int r) {
                    MRank& rk = M->ranks[r];
                    std::lock_guard<std::mutex> l(rk.c->mu);
                    vrc[r] = [&]() -> int32_t {
                        MCHK(hipSetDevice(rk.device));
                        const long i0 = n * r / R, i1 = n * (r + 1) / R;
                        MCHK(hipMemcpyAsync(rk.alpha_blk, alpha_pin, sizeof(double) * (size_t)npad, hipMemcpyHostToDevice, rk.c->sm));
                        RC(eng_kvec(rk.c, rk.c->sm, rk.xs + i0, npad, rk.xs, npad, d, k->kind, k->variance, n, rk.alpha_blk, rk.ver + i0, i1 - i0));
                        MCHK(hipMemcpyAsync(Ka.data() + i0, rk.ver + i0, sizeof(double) * (size_t)(i1 - i0), hipMemcpyDeviceToHost, rk.c->sm));
                        MCHK(hipStreamSynchronize(rk.c->sm));
                        return 0;
                    }();
            });
        }
        (void)hipSetDevice(c->device);
        for (int r = 0; r < R; ++r)
            if (vrc[r] != 0) {
                why = "the residual pass failed on rank " + std::to_string(r);
                return false;
            }
        double amax = 0, dmax = 0;
        for (long i = 0; i < n; ++i) {
            amax = std::max(amax, std::fabs(alpha_pin[i]));
            dmax = std::max(dmax, std::fabs(rhs_h[i]));
        }
        const double tol = 1e-9 * ((double)n * k->variance * amax + dmax);  // ≥ 1e5 × the rounding of a backward-stable solve, ≤ 1e-3 × the damage seen
        for (long i = 0; i < n; ++i) {
            const double res = Ka[i] + noise_h[i] * alpha_pin[i] - rhs_h[i];
            if (!(std::fabs(res) <= tol)) {
                char b[200];
                snprintf(b, sizeof b, "row %ld of (K + Sigma_y) alpha = delta is off by %.3e (tolerance %.3e)", i, res, tol);
                why = b;
                return false;
            }
        }
        return true;
    };
    long seq = 0;
    double wall_ms = 0;
    int32_t rc = 0;
    for (int attempt = 0;; ++attempt) {
        seq = ++M->seq;
        M->abort.store(0);
        M->fits++;
        for (int r = 0; r < R; ++r) bufs[r].reset(new DevBufs(M->ranks[r].c));
        auto t0 = std::chrono::steady_clock::now();
        {
            M->run_ranks([&](int r) {
                    MRank& rk = M->ranks[r];
                    std::lock_guard<std::mutex> l(rk.c->mu);
                    rk.rc = fit_rank(M, &rk, dm, false, k->kind, k->variance, xs_h.data(), noise_h.data(), rhs_h.data(), ncols, want_alpha, keep,
                                     alpha_pin, scal[r].data(), &infos[r], bufs[r].get(), seq);
                    if (rk.rc != 0) {
                        rk.err = gp_last_error();
                        M->abort.store(1);
                        if (rk.rc != -1992) {  // (a stream that never drains would block these as well)
                            (void)hipStreamSynchronize(rk.c->sm);
                            (void)hipStreamSynchronize(rk.c->sp);
                            (void)hipStreamSynchronize(rk.sc);
                        }
                    }
            });
        }
        wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        (void)hipSetDevice(c->device);
        rc = 0;
        for (int r = 0; r < R; ++r)
            if (M->ranks[r].rc != 0 && M->ranks[r].rc != -1999) {
                rc = set_err_text(M->ranks[r].rc, "rank " + std::to_string(r) + ": " + M->ranks[r].err);
                break;
            }
        if (rc == 0)
            for (int r = 0; r < R; ++r)
                if (M->ranks[r].rc != 0) rc = set_err_text(M->ranks[r].rc, M->ranks[r].err);
        if (rc != 0) break;  // hard error: no second attempt
        int info = 0;  // the FIRST failing leading minor (LAPACK dpotrf info; PosDefException(info) in the reference)
        for (int r = 0; r < R; ++r)
            if (infos[r] > 0 && (info == 0 || infos[r] < info)) info = infos[r];
        std::string why;
        bool ok = info == 0;
        if (ok && M->inject_fault) {  // diagnostic ("multi_inject_fault"): spoil one entry of alpha once, as a kernel run on stale arguments would
            alpha_pin[n / 3] += 1.0;
            M->inject_fault = 0;
        }
        if (ok && verify) {
            double zz = 0;
            for (int r = 0; r < R; ++r) zz += scal[r][8];
            ok = verify_result(zz, why);
        } else if (!ok) {
            why = "leading minor of order " + std::to_string(info) + " not positive definite";
        }
        if (ok) break;
        if (verify && attempt == 0) {  // once more, from the inputs
            M->retries++;
            if (getenv("GPMI_VERBOSE")) fprintf(stderr, "[gpmi355] multi-device fit repeated: %s\n", why.c_str());
            release_bufs();
            for (int r = 0; r < R; ++r) {
                std::fill(scal[r].begin(), scal[r].end(), 0.0);
                infos[r] = 0;
            }
            memset(alpha_pin, 0, sizeof(double) * (size_t)npad);
            continue;
        }
        rc = info != 0 ? info : set_err_text(-1991, "multi-device fit failed its self-check twice: " + why);
        break;
    }
    M->tr = nullptr;
    if (tr.f) {
        fclose(tr.f);
        tr.f = nullptr;
    }
    if (rc != 0) {
        release_bufs();
        (void)hipHostFree(alpha_pin);
        return rc;
    }
    double logdet_half = 0;
    std::vector<double> ss((size_t)ncols, 0.0);
    for (int r = 0; r < R; ++r) {
        logdet_half += scal[r][0];
        for (int s = 0; s < ncols; ++s) ss[s] += scal[r][8 + s];
    }
    const double LOG2PI_ = 1.8378770664093454835606594728112;
    for (int s = 0; s < ncols; ++s) logpdf_out[s] = -0.5 * ((double)n * LOG2PI_ + 2.0 * logdet_half + ss[s]);
    if (terms_out) {
        terms_out[0] = 2.0 * logdet_half;
        for (int s = 0; s < ncols; ++s) terms_out[1 + s] = ss[s];
    }
    // timings of rank 0 for gp_get_timings (bench roofline)
    c->tm = gp_timings{};
    c->tm.total_ms = wall_ms;
    c->tm.potrf_ms = wall_ms;
    c->tm.gemm_ms = M->ranks[0].gemm_ms;
    c->tm.gemm_flops = M->ranks[0].gemm_flops;
    c->tm.gemm_launches = M->ranks[0].gemm_launches;

    if (post) {
        if (alpha_out) memcpy(alpha_out, alpha_pin, sizeof(double) * (size_t)n);
        // the posterior handle: inputs + α on the ctx's first device in the single-device layout; the factor stays as pieces
        const long np = (n + 127) / 128 * 128;
        DevBufs mb(c);
        void *xs_v = 0, *al_v = 0;
        int32_t rc2 = mb.get(sizeof(double) * (size_t)d * np, &xs_v);
        if (rc2 == 0) rc2 = mb.get(sizeof(double) * (size_t)np, &al_v);
        if (rc2 == 0) {
            std::vector<double> xs1((size_t)d * np, 0.0), al1((size_t)np, 0.0);
            for (int dd = 0; dd < d; ++dd) memcpy(&xs1[(size_t)dd * np], &xs_h[(size_t)dd * npad], sizeof(double) * (size_t)n);
            memcpy(al1.data(), alpha_pin, sizeof(double) * (size_t)n);
            if (hipMemcpy(xs_v, xs1.data(), sizeof(double) * xs1.size(), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(al_v, al1.data(), sizeof(double) * al1.size(), hipMemcpyHostToDevice) != hipSuccess)
                rc2 = set_err_text(-1994, "upload of the posterior vectors failed");
        }
        if (rc2 != 0) {
            release_bufs();
            (void)hipHostFree(alpha_pin);
            return rc2;
        }
        gp_multi_post* mp = new gp_multi_post();
        mp->P = P; mp->Q = Q; mp->n = n; mp->npad = npad; mp->nblk = nblk; mp->nb = NB;
        for (long kb = 0; kb < nblk; ++kb) mp->valid.push_back(std::max(0L, std::min(NB, n - kb * NB)));
        for (int r = 0; r < R; ++r) {
            MRank& rk = M->ranks[r];
            mp->pieces.push_back({rk.c, rk.A, rk.ld, rk.m_loc, rk.n_loc});
            rk.c->refs++;  // the piece keeps its rank context alive
        }
        post->ctx = c;
        post->dtype = 0;
        post->n = n; post->np = np; post->ld = np + c->ldpad; post->mtot = np + 128; post->d = d;
        post->kind = k->kind; post->variance = k->variance; post->nscale = k->nscale;
        post->scale.clear();
        if (k->scale && k->nscale > 0) post->scale.assign(k->scale, k->scale + k->nscale);
        post->A = nullptr; post->A_bytes = 0;
        post->xs = mb.keep(xs_v); post->xs_bytes = sizeof(double) * (size_t)d * np;
        post->alpha = mb.keep(al_v); post->alpha_bytes = sizeof(double) * (size_t)np;
        post->logdet_half = logdet_half;
        post->pieces = mp;
    }
    for (int r = 0; r < R; ++r) {  // everything but the kept factor pieces goes back to the rank caches
        std::lock_guard<std::mutex> l(M->ranks[r].c->mu);
        (void)hipSetDevice(M->ranks[r].device);
        bufs[r].reset();
    }
    (void)hipSetDevice(c->device);
    (void)hipHostFree(alpha_pin);
    return 0;
}

// The schedule of a solve on the distributed factor for a P×Q grid over nblk block columns, traced without a device (see
// gp_multi_schedule_trace).  flags: 1 = given right-hand sides (instead of K(x*, x)), 2 = the rows also go to the new block rows of an
// extended factor (sequential update; lcm(P, Q) new block rows), 4 = two backward sweeps after the forward pass (gp_posterior_solve).
extern "C" int32_t gp_multi_solve_trace_ex(int32_t P, int32_t Q, int32_t nblk_in, int32_t flags, const char* path) {
    if (P < 1 || Q < 1 || P * Q > 64) return set_arg_err(1, "P, Q");
    if (nblk_in < 1 || nblk_in > 4096) return set_arg_err(3, "nblk");
    if (!path) return set_arg_err(5, "path is NULL");
    Trace tr;
    tr.f = fopen(path, "w");
    if (!tr.f) return set_arg_err(5, "cannot open the trace file");
    gp_multi M;
    M.P = P; M.Q = Q; M.R = P * Q; M.nb = 128; M.comm = 2; M.tr = &tr; M.timeout_s = 60;
    const long lcm = lcm_of(P, Q);
    const long nblk = (nblk_in + lcm - 1) / lcm * lcm;
    M.ranks.resize((size_t)M.R);
    for (int r = 0; r < M.R; ++r) {
        MRank& rk = M.ranks[r];
        rk.r = r; rk.p = r / Q; rk.q = r % Q;
    }
    int32_t rc = size_xevents(&M, nblk, true);
    if (rc == 0) {
        char b[200];
        snprintf(b, sizeof b, "{\"t\":\"hdr\",\"P\":%d,\"Q\":%d,\"nblk\":%ld,\"depth\":0,\"comm\":2,\"dry\":1,\"mode\":\"solve\",\"flags\":%d}", P, Q, nblk, flags);
        tr.line(b);
        SolveDims sd{nblk * 128, nblk * 128, nblk, 128, 128, (flags & 2) ? lcm * 128 : 128, 1};
        sd.rhs = (flags & 1) != 0;
        sd.sink_nb2 = (flags & 2) ? lcm : 0;
        sd.nbwd = (flags & 4) ? 2 : 0;
        sd.seq_bwd0 = 2;
        std::vector<std::thread> th;
        for (int r = 0; r < M.R; ++r)
            th.emplace_back([&, r]() {
                MRank& rk = M.ranks[r];
                rk.rc = solve_rank(&M, &rk, sd, true, nullptr, 0, 1.0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1);
                if (rk.rc != 0) {
                    rk.err = gp_last_error();
                    M.abort.store(1);
                }
            });
        for (auto& t : th) t.join();
    }
    fclose(tr.f);
    tr.f = nullptr;
    if (rc != 0) return rc;
    for (int r = 0; r < M.R; ++r)
        if (M.ranks[r].rc != 0 && M.ranks[r].rc != -1999) return set_err_text(M.ranks[r].rc, "rank " + std::to_string(r) + ": " + M.ranks[r].err);
    return 0;
}
extern "C" int32_t gp_multi_solve_trace(int32_t P, int32_t Q, int32_t nblk_in, const char* path) { return gp_multi_solve_trace_ex(P, Q, nblk_in, 0, path); }

// can this posterior be worked on where the fit left it (block-cyclic pieces, same grid still alive)?
bool multi_can_solve(gp_post* post) {
    gp_multi_post* mp = post->pieces;
    gp_multi* M = post->ctx ? post->ctx->multi : nullptr;
    return mp && M && M->dist_predict && M->P == mp->P && M->Q == mp->Q && (int)mp->pieces.size() == M->R && post->dtype == 0;
}

namespace {
// compact index of the first real point of every block (off[nblk] = number of real points)
std::vector<long> block_offsets(const std::vector<long>& valid) {
    std::vector<long> off(valid.size() + 1, 0);
    for (size_t k = 0; k < valid.size(); ++k) off[k + 1] = off[k] + valid[k];
    return off;
}
// compact [rows][ldc] (real points only) -> padded block layout [rows][npad]
void to_padded(const double* comp, long ldc, long rows, const std::vector<long>& valid, long NB, double* pad, long npad) {
    const std::vector<long> off = block_offsets(valid);
    for (long r = 0; r < rows; ++r)
        for (size_t k = 0; k < valid.size(); ++k)
            if (valid[k] > 0) memcpy(pad + (size_t)r * npad + k * NB, comp + (size_t)r * ldc + off[k], sizeof(double) * (size_t)valid[k]);
}
void to_compact(const double* pad, long npad, long rows, const std::vector<long>& valid, long NB, double* comp, long ldc) {
    const std::vector<long> off = block_offsets(valid);
    for (long r = 0; r < rows; ++r)
        for (size_t k = 0; k < valid.size(); ++k)
            if (valid[k] > 0) memcpy(comp + (size_t)r * ldc + off[k], pad + (size_t)r * npad + k * NB, sizeof(double) * (size_t)valid[k]);
}

// One solve_rank pass over all ranks (one host thread each): generations, trace file, error collection, buffers back to the caches.
// part[r]: rank r's partial Σ_c X² per row; cpart[r] (want_cov): its partial X Xᵀ; z_host (sd.nbwd rows of npad): backward results.
int32_t run_solve(gp_ctx* c, SolveDims sd, const std::vector<gp_multi_post::Piece>& pieces, int kind, double variance, const double* x_h,
                  const double* xs_h, const double* rhs_h, bool want_cov, double* z_host, std::vector<std::vector<double>>& part,
                  std::vector<std::vector<double>>& cpart) {
    gp_multi* M = c->multi;
    const int R = M->R;
    RC(size_xevents(M, sd.nblk, false));
    (void)hipSetDevice(c->device);
    const long seq = ++M->seq;
    sd.seq_bwd0 = M->seq + 1;
    M->seq += 2 * (long)sd.nbwd + 2;
    M->abort.store(0);
    Trace tr;
    if (const char* tp = getenv("GPMI_TRACE_SCHEDULE")) {
        tr.f = fopen(tp, "w");
        if (tr.f) {
            char b[200];
            snprintf(b, sizeof b, "{\"t\":\"hdr\",\"P\":%d,\"Q\":%d,\"nblk\":%ld,\"depth\":0,\"comm\":2,\"dry\":0,\"mode\":\"solve\",\"flags\":%d}", M->P, M->Q, sd.nblk,
                     (sd.rhs ? 1 : 0) | (sd.sink_nb2 ? 2 : 0) | (sd.nbwd ? 4 : 0));
            tr.line(b);
        }
    }
    M->tr = tr.f ? &tr : nullptr;
    part.assign((size_t)R, std::vector<double>((size_t)sd.nsp, 0.0));
    cpart.assign((size_t)(want_cov ? R : 0), std::vector<double>((size_t)sd.nsp * sd.nsp, 0.0));
    std::vector<std::unique_ptr<DevBufs>> bufs((size_t)R);
    for (int r = 0; r < R; ++r) bufs[r].reset(new DevBufs(M->ranks[r].c));
    {
        M->run_ranks([&](int r) {
                MRank& rk = M->ranks[r];
                std::lock_guard<std::mutex> l(rk.c->mu);
                rk.rc = solve_rank(M, &rk, sd, false, &pieces[r], kind, variance, x_h, xs_h, rhs_h, part[r].data(), want_cov ? cpart[r].data() : nullptr,
                                   z_host, bufs[r].get(), seq);
                if (rk.rc != 0) {
                    rk.err = gp_last_error();
                    M->abort.store(1);
                    if (rk.rc != -1992) (void)hipStreamSynchronize(rk.c->sm);
                }
        });
    }
    M->tr = nullptr;
    if (tr.f) fclose(tr.f);
    for (int r = 0; r < R; ++r) {
        std::lock_guard<std::mutex> l(M->ranks[r].c->mu);
        (void)hipSetDevice(M->ranks[r].device);
        bufs[r].reset();
    }
    (void)hipSetDevice(c->device);
    for (int r = 0; r < R; ++r)
        if (M->ranks[r].rc != 0 && M->ranks[r].rc != -1999) return set_err_text(M->ranks[r].rc, "rank " + std::to_string(r) + ": " + M->ranks[r].err);
    for (int r = 0; r < R; ++r)
        if (M->ranks[r].rc != 0) return set_err_text(M->ranks[r].rc, M->ranks[r].err);
    M->solves++;
    return 0;
}

// scaled training inputs of a posterior in the padded block layout of its pieces: [d][npad]
int32_t padded_inputs(gp_post* post, std::vector<double>& x_h) {
    gp_multi_post* mp = post->pieces;
    gp_ctx* c = post->ctx;
    const int d = post->d;
    const long np = post->np;
    std::vector<double> x1((size_t)d * np);
    MCHK(hipSetDevice(c->device));
    MCHK(hipMemcpy(x1.data(), post->xs, sizeof(double) * x1.size(), hipMemcpyDeviceToHost));
    x_h.assign((size_t)d * mp->npad, 0.0);
    to_padded(x1.data(), np, d, mp->valid, mp->nb, x_h.data(), mp->npad);
    return 0;
}
}  // namespace

// var_sub[s] = Σ_c (K_*x L⁻ᵀ)[s][c]² for the ns test points xs_h (scaled, dimension-major [d][ns_ld]) — the amount the posterior
// variance lies below the prior variance — on the block-cyclic pieces of the factor.  Called with the main ctx locked.
int32_t multi_predict_var(gp_post* post, const double* xs_scaled, long ns_ld, long ns, double* var_sub, double* cov_sub) {
    gp_multi_post* mp = post->pieces;
    gp_ctx* c = post->ctx;
    gp_multi* M = c->multi;
    const int R = M->R, d = post->d;
    std::vector<double> x_h;
    RC(padded_inputs(post, x_h));
    // cov_sub (nullable): ns×ns, (X Xᵀ)[s][t] — all test points in ONE chunk then (the caller bounds ns)
    const long CH = cov_sub ? std::max(ns, 1L) : 4096;
    for (long s0 = 0; s0 < ns; s0 += CH) {
        const long nsc = std::min(CH, ns - s0), nsp = (nsc + 127) / 128 * 128;
        std::vector<double> xs_h((size_t)d * nsp, 0.0);
        for (int dd = 0; dd < d; ++dd) memcpy(&xs_h[(size_t)dd * nsp], xs_scaled + (size_t)dd * ns_ld + s0, sizeof(double) * (size_t)nsc);
        SolveDims sd{mp->n, mp->npad, mp->nblk, mp->nb, nsc, nsp, d};
        sd.valid = mp->valid.data();
        std::vector<std::vector<double>> part, cpart;
        RC(run_solve(c, sd, mp->pieces, post->kind, post->variance, x_h.data(), xs_h.data(), nullptr, cov_sub != nullptr, nullptr, part, cpart));
        for (long i = 0; i < nsc; ++i) {
            double acc = 0;
            for (int r = 0; r < R; ++r) acc += part[r][i];
            var_sub[s0 + i] = acc;
        }
        if (cov_sub)  // lower triangles of the partial products (row-major, row stride nsp) -> the full symmetric ns×ns sum
            for (long i = 0; i < nsc; ++i)
                for (long j = 0; j <= i; ++j) {
                    double acc = 0;
                    for (int r = 0; r < R; ++r) acc -= cpart[r][(size_t)i * nsp + j];  // (the SYRK kernel computes C −= X Xᵀ from zero)
                    cov_sub[(size_t)i * ns + j] = cov_sub[(size_t)j * ns + i] = acc;
                }
    }
    return 0;
}

// out[:, s] = C \ B[:, s] on the block-cyclic pieces (C = the fitted K + Σy): the columns of B travel as rows, forward pass
// X = Bᵀ L⁻ᵀ and one backward sweep per column.  B, out: n×ncols column-major host arrays.  Called with the main ctx locked.
int32_t multi_solve(gp_post* post, const double* B, int ncols, double* out) {
    gp_multi_post* mp = post->pieces;
    gp_ctx* c = post->ctx;
    const long n = mp->n, npad = mp->npad, nsp = (ncols + 127) / 128 * 128;
    std::vector<double> rhs_h((size_t)nsp * npad, 0.0), z_h((size_t)ncols * npad, 0.0);
    to_padded(B, n, ncols, mp->valid, mp->nb, rhs_h.data(), npad);
    SolveDims sd{n, npad, mp->nblk, mp->nb, ncols, nsp, post->d};
    sd.valid = mp->valid.data();
    sd.rhs = true;
    sd.nbwd = ncols;
    std::vector<std::vector<double>> part, cpart;
    RC(run_solve(c, sd, mp->pieces, post->kind, post->variance, nullptr, nullptr, rhs_h.data(), false, z_h.data(), part, cpart));
    to_compact(z_h.data(), npad, ncols, mp->valid, mp->nb, out, n);
    return 0;
}

// out[:, s] = L ξ[:, s] = C.U' ξ (the sampling transform, src/finite_gp_projection.jl:233-237, 271-277) on the block-cyclic pieces: every
// rank multiplies the blocks it holds with its share of ξ (one launch: the reach of every local row is contiguous in the piece), the
// process rows' partial products are summed on the host.  No exchange between the ranks.  Called with the main ctx locked.
int32_t multi_factor_mul(gp_post* post, const double* xi, int ncols, double* out) {
    gp_multi_post* mp = post->pieces;
    gp_ctx* c = post->ctx;
    gp_multi* M = c->multi;
    const int P = mp->P, Q = mp->Q, R = M->R;
    const long n = mp->n, npad = mp->npad, NB = mp->nb, nlb_r = mp->nblk / P, nlb_c = mp->nblk / Q;
    const long m_loc = nlb_r * NB, n_loc = nlb_c * NB;
    std::vector<double> xi_pad((size_t)ncols * npad, 0.0), out_pad((size_t)ncols * npad, 0.0);
    to_padded(xi, n, ncols, mp->valid, NB, xi_pad.data(), npad);
    std::vector<std::vector<double>> part((size_t)R);
    std::vector<int32_t> rcs((size_t)R, 0);
    std::vector<std::string> errs((size_t)R);
    {
        M->run_ranks([&](int r) {
                MRank& rk = M->ranks[r];
                std::lock_guard<std::mutex> l(rk.c->mu);
                rcs[r] = [&]() -> int32_t {
                    const int q = r % Q, p = r / Q;
                    const auto& pc = mp->pieces[r];
                    MCHK(hipSetDevice(rk.device));
                    std::vector<double> v_loc((size_t)ncols * n_loc);
                    for (int s = 0; s < ncols; ++s)
                        for (long lc = 0; lc < nlb_c; ++lc)
                            memcpy(&v_loc[(size_t)s * n_loc + lc * NB], &xi_pad[(size_t)s * npad + (lc * Q + q) * NB], sizeof(double) * (size_t)NB);
                    DevBufs b(rk.c);
                    void *v_v = 0, *o_v = 0;
                    RC(b.get(sizeof(double) * v_loc.size(), &v_v));
                    RC(b.get(sizeof(double) * (size_t)ncols * m_loc, &o_v));
                    hipStream_t sm = rk.c->sm;
                    part[r].assign((size_t)ncols * m_loc, 0.0);
                    MCHK(hipMemcpyAsync(v_v, v_loc.data(), sizeof(double) * v_loc.size(), hipMemcpyHostToDevice, sm));
                    hipLaunchKernelGGL(mk_rowdot_kernel, dim3((unsigned)m_loc, (unsigned)ncols), dim3(256), 0, sm, (const double*)pc.A, pc.ld, NB, P, p, Q, q,
                                       (const double*)v_v, n_loc, (double*)o_v, m_loc);
                    MCHK(hipGetLastError());
                    MCHK(hipMemcpyAsync(part[r].data(), o_v, sizeof(double) * part[r].size(), hipMemcpyDeviceToHost, sm));
                    MCHK(hipStreamSynchronize(sm));
                    return 0;
                }();
                if (rcs[r] != 0) errs[r] = gp_last_error();
        });
    }
    (void)hipSetDevice(c->device);
    for (int r = 0; r < R; ++r)
        if (rcs[r] != 0) return set_err_text(rcs[r], "rank " + std::to_string(r) + ": " + errs[r]);
    for (int r = 0; r < R; ++r) {
        const int p = r / Q;
        for (int s = 0; s < ncols; ++s)
            for (long li = 0; li < nlb_r; ++li) {
                const double* src = &part[r][(size_t)s * m_loc + li * NB];
                double* dst = &out_pad[(size_t)s * npad + (li * P + p) * NB];
                for (long t = 0; t < NB; ++t) dst[t] += src[t];
            }
    }
    to_compact(out_pad.data(), npad, ncols, mp->valid, NB, out, n);
    M->solves++;
    return 0;
}

// Sequential conditioning on the pieces (src/exact_gpr_posterior.jl:46-56, update_chol src/util/common_covmat_ops.jl:38-42): the factor
// is EXTENDED where it lives.  New observations become new block rows (their own padded blocks) of a new set of pieces:
//   U12ᵀ = K(x2, x1) L11⁻ᵀ     the forward solve on the old pieces; every rank keeps the rows of the blocks it owns ("sink")
//   U22ᵀ = chol(C22 − U12ᵀU12)  n2 ≤ 4 096: X Xᵀ is summed from the ranks' SYRKs, factored on the first device, its blocks sent to their owners
//   α    = L⁻ᵀ L⁻¹ δ           forward + backward sweep over the extended pieces
// Called with the main ctx locked.  Status −1991: the forward / backward consistency check failed (the caller gathers instead).
int32_t multi_update(gp_post* old, const gp_points* x2, const gp_noise* noise2, const void* delta_all, gp_post* post, void* alpha_out,
                     double* logpdf_out) {
    gp_multi_post* mp0 = old->pieces;
    gp_ctx* c = old->ctx;
    gp_multi* M = c->multi;
    const int P = M->P, Q = M->Q, R = M->R, d = old->d;
    const long NB = mp0->nb, nblk0 = mp0->nblk, npad0 = mp0->npad, n0 = mp0->n;
    const long n2 = x2->n, nsp = (n2 + 127) / 128 * 128;
    const long lcm = lcm_of(P, Q);
    const long nb2 = ((n2 + NB - 1) / NB + lcm - 1) / lcm * lcm;
    const long nblk1 = nblk0 + nb2, npad1 = nblk1 * NB, n1 = n0 + n2, Sp = nb2 * NB, lds = Sp + 32;
    const long nlb_r0 = nblk0 / P, nlb_c0 = nblk0 / Q, nlb_r1 = nblk1 / P, nlb_c1 = nblk1 / Q;
    const long ld2 = nlb_c1 * NB + 32, m_loc2 = nlb_r1 * NB, n_loc2 = nlb_c1 * NB;
    const size_t A2_b = sizeof(double) * (size_t)(m_loc2 + 128) * ld2;
    std::vector<long> valid1 = mp0->valid;
    for (long t = 0; t < nb2; ++t) valid1.push_back(std::max(0L, std::min(NB, n2 - t * NB)));
    const double* dl = (const double*)delta_all;

    // ---- host marshalling: scaled new inputs (as the fit scaled the old ones), their noise, the old inputs in block layout
    std::vector<double> xs2_h((size_t)d * nsp, 0.0), xs2S_h((size_t)d * Sp, 0.0), nz_h((size_t)Sp, 0.0), x0_h;
    {
        const double* pd = (const double*)x2->data;
        for (int dd = 0; dd < d; ++dd) {
            const double sc = old->nscale == 1 ? old->scale[0] : (old->nscale > 1 ? old->scale[dd] : 1.0);
            for (long i = 0; i < n2; ++i) {
                const double v = sc * (x2->layout == 0 ? pd[i] : (x2->layout == 1 ? pd[(long)dd + i * x2->d] : pd[i + (long)dd * x2->n]));
                xs2_h[(size_t)dd * nsp + i] = v;
                xs2S_h[(size_t)dd * Sp + i] = v;
            }
        }
        for (long i = 0; i < n2; ++i) nz_h[i] = noise2->kind == 0 ? noise2->s : ((const double*)noise2->diag)[i];
    }
    RC(padded_inputs(old, x0_h));

    // ---- new pieces: zero, the old piece copied in (device-local), the new block rows filled below
    std::vector<void*> A2((size_t)R, nullptr);
    auto drop_new = [&]() {
        for (int r = 0; r < R; ++r) {
            MRank& rk = M->ranks[r];
            std::lock_guard<std::mutex> l(rk.c->mu);
            (void)hipSetDevice(rk.device);
            if (A2[r]) ctx_release(rk.c, A2[r], 0);
            A2[r] = nullptr;
            rk.A2 = nullptr;
        }
        (void)hipSetDevice(c->device);
    };
    int32_t rc = 0;
    for (int r = 0; r < R && rc == 0; ++r) {
        MRank& rk = M->ranks[r];
        std::lock_guard<std::mutex> l(rk.c->mu);
        rc = [&]() -> int32_t {
            MCHK(hipSetDevice(rk.device));
            RC(ctx_alloc(rk.c, A2_b, &A2[r]));
            const auto& pc = mp0->pieces[r];
            MCHK(hipMemsetAsync(A2[r], 0, A2_b, rk.c->sm));
            MCHK(hipMemcpy2DAsync(A2[r], sizeof(double) * ld2, pc.A, sizeof(double) * pc.ld, sizeof(double) * nlb_c0 * NB, nlb_r0 * NB, hipMemcpyDeviceToDevice, rk.c->sm));
            rk.A2 = (double*)A2[r];
            rk.ld2 = ld2;
            return 0;
        }();
    }
    (void)hipSetDevice(c->device);
    if (rc != 0) {
        drop_new();
        return rc;
    }

    // ---- U12ᵀ = K(x2, x1) L11⁻ᵀ on the old pieces, rows kept by their new owners; X Xᵀ from the ranks' SYRKs
    std::vector<std::vector<double>> part, cpart;
    {
        SolveDims sd{n0, npad0, nblk0, NB, n2, nsp, d};
        sd.valid = mp0->valid.data();
        sd.sink_nb2 = nb2;
        rc = run_solve(c, sd, mp0->pieces, old->kind, old->variance, x0_h.data(), xs2_h.data(), nullptr, true, nullptr, part, cpart);
    }
    // ---- U22ᵀ = chol(K(x2, x2) + Σy2 − X Xᵀ) on the first device (identity padding up to the block boundary), blocks to their owners
    double logdet2_half = 0;
    if (rc == 0) {
        std::vector<double> nxx((size_t)nsp * nsp, 0.0);
        for (long i = 0; i < n2; ++i)
            for (long j = 0; j <= i; ++j) {
                double acc = 0;
                for (int r = 0; r < R; ++r) acc += cpart[r][(size_t)i * nsp + j];  // (= −(X Xᵀ)[i][j]: the SYRK kernel computes C −= X Xᵀ from zero)
                nxx[(size_t)i * nsp + j] = nxx[(size_t)j * nsp + i] = acc;
            }
        DevBufs mb(c);
        void *x2_v = 0, *nz_v = 0, *S_v = 0, *T_v = 0;
        int info_h = 0;
        double scal_h[2] = {0, 0};
        rc = [&]() -> int32_t {
            MCHK(hipSetDevice(c->device));
            RC(mb.get(sizeof(double) * (size_t)d * Sp, &x2_v));
            RC(mb.get(sizeof(double) * (size_t)Sp, &nz_v));
            RC(mb.get(sizeof(double) * (size_t)(Sp + 128) * lds, &S_v));
            RC(mb.get(sizeof(double) * (size_t)nsp * nsp, &T_v));
            if (!c->info_dev) MCHK(hipMalloc((void**)&c->info_dev, sizeof(int)));
            RC(ctx_scal(c, 16));
            hipStream_t sm = c->sm;
            MCHK(hipMemcpyAsync(x2_v, xs2S_h.data(), sizeof(double) * (size_t)d * Sp, hipMemcpyHostToDevice, sm));
            MCHK(hipMemcpyAsync(nz_v, nz_h.data(), sizeof(double) * (size_t)Sp, hipMemcpyHostToDevice, sm));
            MCHK(hipMemcpyAsync(T_v, nxx.data(), sizeof(double) * (size_t)nsp * nsp, hipMemcpyHostToDevice, sm));
            MCHK(hipMemsetAsync(c->info_dev, 0, sizeof(int), sm));
            MCHK(hipMemsetAsync(c->scal_dev, 0, sizeof(double) * 16, sm));
            MCHK(hipMemsetAsync(S_v, 0, sizeof(double) * (size_t)(Sp + 128) * lds, sm));
            RC(eng_assemble(c, sm, old->kind, old->variance, (const double*)x2_v, n2, Sp, d, (const double*)nz_v, plain_map(1, 0, 0), (double*)S_v, lds, Sp, Sp));
            hipLaunchKernelGGL(mk_addmat_kernel, dim3((unsigned)std::min<long>(1024, (nsp * nsp + 255) / 256)), dim3(256), 0, sm, (double*)S_v, lds, (const double*)T_v,
                               nsp, nsp, nsp);
            MCHK(hipGetLastError());
            RC(eng_potrf(c, sm, (double*)S_v, lds, Sp, Sp, c->info_dev, 0, n2, c->scal_dev));
            MCHK(hipMemcpyAsync(&info_h, c->info_dev, sizeof(int), hipMemcpyDeviceToHost, sm));
            MCHK(hipMemcpyAsync(scal_h, c->scal_dev, sizeof(double), hipMemcpyDeviceToHost, sm));
            MCHK(hipStreamSynchronize(sm));
            if (info_h != 0) return (int32_t)n0 + info_h;  // order of the failing leading minor of the bordered matrix
            for (long t = 0; t < nb2; ++t)
                for (long u = 0; u <= t; ++u) {
                    const long I = nblk0 + t, J = nblk0 + u;
                    MRank& rk = M->ranks[(size_t)(I % P) * Q + (size_t)(J % Q)];
                    MCHK(hipMemcpy2DAsync(rk.A2 + (I / P) * NB * ld2 + (J / Q) * NB, sizeof(double) * ld2, (const double*)S_v + t * NB * lds + u * NB,
                                          sizeof(double) * lds, sizeof(double) * NB, NB, hipMemcpyDefault, sm));
                }
            MCHK(hipStreamSynchronize(sm));
            return 0;
        }();
        if (rc != 0) (void)hipStreamSynchronize(c->sm);
        logdet2_half = scal_h[0];
    }
    // ---- α = L⁻ᵀ L⁻¹ δ over the extended pieces; ‖L⁻¹δ‖² from the forward pass
    std::vector<gp_multi_post::Piece> pieces1;
    for (int r = 0; r < R; ++r) pieces1.push_back({M->ranks[r].c, A2[r], ld2, m_loc2, n_loc2});
    std::vector<double> z_h((size_t)npad1, 0.0), al((size_t)n1, 0.0);
    double sqm = 0;
    if (rc == 0) {
        std::vector<double> rhs_h((size_t)128 * npad1, 0.0);
        to_padded(dl, n1, 1, valid1, NB, rhs_h.data(), npad1);
        SolveDims sd{n1, npad1, nblk1, NB, 1, 128, d};
        sd.valid = valid1.data();
        sd.rhs = true;
        sd.nbwd = 1;
        rc = run_solve(c, sd, pieces1, old->kind, old->variance, nullptr, nullptr, rhs_h.data(), false, z_h.data(), part, cpart);
        if (rc == 0) {
            for (int r = 0; r < R; ++r) sqm += part[r][0];
            to_compact(z_h.data(), npad1, 1, valid1, NB, al.data(), n1);
            double da = 0, dn = 0;
            for (long i = 0; i < n1; ++i) {
                da += dl[i] * al[i];
                dn += std::fabs(dl[i] * al[i]);
            }
            if (!(std::fabs(da - sqm) <= 1e-7 * (dn + std::fabs(sqm)))) {
                char b[200];
                snprintf(b, sizeof b, "sequential update on the pieces: delta'alpha = %.15g but ||L^-1 delta||^2 = %.15g", da, sqm);
                rc = set_err_text(-1991, b);
            }
        }
    }
    for (int r = 0; r < R; ++r) M->ranks[r].A2 = nullptr;
    // ---- the new handle: inputs + α on the first device (compact), the factor as the new pieces
    const long np1 = (n1 + 127) / 128 * 128;
    DevBufs mb(c);
    void *xs_v = 0, *al_v = 0;
    if (rc == 0) {
        rc = [&]() -> int32_t {
            MCHK(hipSetDevice(c->device));
            RC(mb.get(sizeof(double) * (size_t)d * np1, &xs_v));
            RC(mb.get(sizeof(double) * (size_t)np1, &al_v));
            MCHK(hipMemset(xs_v, 0, sizeof(double) * (size_t)d * np1));
            MCHK(hipMemset(al_v, 0, sizeof(double) * (size_t)np1));
            MCHK(hipMemcpy2D(xs_v, sizeof(double) * np1, old->xs, sizeof(double) * old->np, sizeof(double) * n0, d, hipMemcpyDeviceToDevice));
            MCHK(hipMemcpy2D((double*)xs_v + n0, sizeof(double) * np1, xs2_h.data(), sizeof(double) * nsp, sizeof(double) * n2, d, hipMemcpyHostToDevice));
            MCHK(hipMemcpy(al_v, al.data(), sizeof(double) * (size_t)n1, hipMemcpyHostToDevice));
            // Self-check of the NEW rows through an independent path (as multi_fit checks every row): the identity above,
            // δᵀα = ‖L⁻¹δ‖², holds for ANY triangular L, so a wrong U12 (sink rows) or U22 block would pass it.  The rows of
            //   (K([x1; x2], [x1; x2]) + Σy) α = δ   that belong to x2 involve U12 and U22:  K(x2, [x1; x2]) α + Σy2 α2 = δ2,
            // with K·α evaluated from the inputs (kvec kernel, no factor).  A mismatch returns −1991: the caller gathers instead.
            void* kv_v = nullptr;
            RC(mb.get(sizeof(double) * (size_t)nsp, &kv_v));
            RC(eng_kvec(c, c->sm, (const double*)xs_v + n0, np1, (const double*)xs_v, np1, d, old->kind, old->variance, n1, (const double*)al_v,
                        (double*)kv_v, n2));
            std::vector<double> kv((size_t)n2);
            MCHK(hipMemcpyAsync(kv.data(), kv_v, sizeof(double) * (size_t)n2, hipMemcpyDeviceToHost, c->sm));
            MCHK(hipStreamSynchronize(c->sm));
            double res = 0, amax = 0, dmax = 0;
            for (long i = 0; i < n1; ++i) {
                amax = std::max(amax, std::fabs(al[(size_t)i]));
                dmax = std::max(dmax, std::fabs(dl[i]));
            }
            for (long i = 0; i < n2; ++i) {
                const double a2 = al[(size_t)(n0 + i)], lhs = kv[(size_t)i] + nz_h[(size_t)i] * a2;
                res = std::max(res, std::fabs(lhs - dl[n0 + i]));
            }
            const double scale = (double)n1 * old->variance * amax + dmax;  // the bound multi_fit's every-row check uses
            if (!(res <= 1e-9 * scale)) {
                char b[220];
                snprintf(b, sizeof b, "sequential update on the pieces: rows of the new observations miss (K + Sigma) alpha = delta by %.3g (scale %.3g)", res, scale);
                return set_err_text(-1991, b);
            }
            return 0;
        }();
        if (rc != 0) (void)hipStreamSynchronize(c->sm);
    }
    if (rc != 0) {
        drop_new();
        return rc;
    }
    if (alpha_out) memcpy(alpha_out, al.data(), sizeof(double) * (size_t)n1);
    gp_multi_post* mp = new gp_multi_post();
    mp->P = P; mp->Q = Q; mp->n = n1; mp->npad = npad1; mp->nblk = nblk1; mp->nb = NB;
    mp->valid = valid1;
    mp->pieces = pieces1;
    for (int r = 0; r < R; ++r) M->ranks[r].c->refs++;  // every piece keeps its rank context alive
    post->ctx = c;
    post->dtype = 0;
    post->n = n1; post->np = np1; post->ld = np1 + c->ldpad; post->mtot = np1 + 128; post->d = d;
    post->kind = old->kind; post->variance = old->variance; post->nscale = old->nscale;
    post->scale = old->scale;
    post->A = nullptr; post->A_bytes = 0;
    post->xs = mb.keep(xs_v); post->xs_bytes = sizeof(double) * (size_t)d * np1;
    post->alpha = mb.keep(al_v); post->alpha_bytes = sizeof(double) * (size_t)np1;
    post->logdet_half = old->logdet_half + logdet2_half;
    post->pieces = mp;
    if (logpdf_out) *logpdf_out = -0.5 * ((double)n1 * 1.8378770664093454835606594728112 + 2.0 * post->logdet_half + sqm);
    return 0;
}

void multi_post_release(gp_post* post) {
    gp_multi_post* mp = post->pieces;
    if (!mp) return;
    for (auto& pc : mp->pieces) {
        {
            std::lock_guard<std::mutex> l(pc.c->mu);
            (void)hipSetDevice(pc.c->device);
            ctx_release(pc.c, pc.A, 0);
        }
        ctx_unref(pc.c);
    }
    (void)hipSetDevice(post->ctx->device);
    delete mp;
    post->pieces = nullptr;
}

// Block-cyclic pieces -> the single-device row-major lower factor on the ctx's first device (288 GB of HBM hold the 34 GB of
// N = 65 536 many times over), for what is not done on the pieces (C.U handed out, factor_mul, > 4 096 test points with a full
// covariance).  Only the real points travel: the padding inside the blocks (the tail of a fit, the tail of every batch of a
// sequential update) is dropped and the single-device identity padding up to np is written instead.  Called with the main ctx locked.
__global__ void mk_unit_diag_kernel(double* A, long ld, long i0, long i1) {
    const long i = i0 + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < i1) A[i * ld + i] = 1.0;
}
int32_t multi_gather(gp_post* post) {
    gp_multi_post* mp = post->pieces;
    if (!mp) return 0;
    gp_ctx* c = post->ctx;
    const long NB = mp->nb, n = post->n, np = post->np, ld = post->ld, mtot = post->mtot;
    const int P = mp->P, Q = mp->Q;
    const std::vector<long> off = block_offsets(mp->valid);
    MCHK(hipSetDevice(c->device));
    void* A_v = nullptr;
    const size_t A_b = sizeof(double) * (size_t)(mtot + 128) * ld;
    RC(ctx_alloc(c, A_b, &A_v));
    double* A = (double*)A_v;
    int32_t rc = [&]() -> int32_t {
        MCHK(hipMemsetAsync(A + n * ld, 0, sizeof(double) * (size_t)(mtot - n + 128) * ld, c->sm));
        for (size_t r = 0; r < mp->pieces.size(); ++r) {
            auto& pc = mp->pieces[r];
            const int p = (int)r / Q, q = (int)r % Q;
            for (long li = 0; li < mp->nblk / P; ++li) {
                const long gi = li * P + p, rows = mp->valid[gi];
                if (rows <= 0) continue;
                for (long lj = 0; lj < mp->nblk / Q; ++lj) {
                    const long gj = lj * Q + q;
                    if (gj > gi) break;
                    const long cols = mp->valid[gj];
                    if (cols <= 0) continue;
                    MCHK(hipMemcpy2DAsync(A + off[gi] * ld + off[gj], sizeof(double) * ld, (const double*)pc.A + li * NB * pc.ld + lj * NB,
                                          sizeof(double) * pc.ld, sizeof(double) * cols, rows, hipMemcpyDefault, c->sm));
                }
            }
        }
        if (np > n) {
            hipLaunchKernelGGL(mk_unit_diag_kernel, dim3((unsigned)((np - n + 255) / 256)), dim3(256), 0, c->sm, A, ld, n, np);
            MCHK(hipGetLastError());
        }
        MCHK(hipStreamSynchronize(c->sm));
        return 0;
    }();
    if (rc != 0) {
        (void)hipStreamSynchronize(c->sm);
        ctx_release(c, A_v, 0);
        return rc;
    }
    post->A = A_v;
    post->A_bytes = A_b;
    multi_post_release(post);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// gp_rccl_selftest: the transport's library on ONE device, without a multi-device context — dlopen (librccl, or GPMI_RCCL_LIB), the
// seven entry points, ncclCommInitAll(1), and one grouped ncclSend / ncclRecv pair of `count` doubles from the rank to itself on a
// stream (RCCL serves a self-pair as a local copy).  Proves on any GPU box what the multi-device driver assumes about the library:
// the symbols exist with the prototypes used, grouped posting works on a communicator made by ncclCommInitAll, and NCCL_FLOAT64 is
// the 8-byte element type (a wrong constant would move a different number of bytes — every element is compared).
// max_abs_err_out (nullable): largest |received − sent|.  Status −1996: library unavailable; −1998: an RCCL call failed.
// ------------------------------------------------------------------------------------------------
extern "C" int32_t gp_rccl_selftest(int32_t device, int64_t count, double* max_abs_err_out) {
    if (count < 1) return set_err_text(-2, "count must be >= 1");
    MCHK(hipSetDevice(device));
    ncclComm_t_ comm = nullptr;
    const int dev = device;
    int nrc = 0;
    {   // the global lock covers the loader, communicator creation and the live counter only: while `live` > 0 the loader refuses to switch
        // libraries, so the entry points used below stay valid without it — a concurrent gp_ctx_create_multi / gp_ctx_destroy is not held
        // up by this test's allocations, transfer and copies
        std::lock_guard<std::mutex> l(g_rccl_mu);
        if (!g_rccl.load()) return set_err_text(-1996, "RCCL is unavailable: " + g_rccl.err);
        nrc = g_rccl.CommInitAll(&comm, 1, &dev);
        if (nrc != 0) return set_err_text(-1998, std::string("ncclCommInitAll(1): ") + g_rccl.GetErrorString(nrc));
        ++g_rccl.live;
    }
    double *a = nullptr, *b = nullptr;
    hipStream_t st = nullptr;
    int32_t rc = [&]() -> int32_t {
        MCHK(hipMalloc((void**)&a, sizeof(double) * (size_t)count));
        MCHK(hipMalloc((void**)&b, sizeof(double) * (size_t)(count + 8)));
        MCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        std::vector<double> h((size_t)count);
        for (int64_t i = 0; i < count; ++i) h[(size_t)i] = 1.0 + (double)i * 0.5;
        MCHK(hipMemcpy(a, h.data(), sizeof(double) * (size_t)count, hipMemcpyHostToDevice));
        MCHK(hipMemset(b, 0xff, sizeof(double) * (size_t)(count + 8)));  // NaN pattern: untouched elements are detected
        nrc = g_rccl.GroupStart();
        if (nrc == 0) nrc = g_rccl.Send(a, (size_t)count, NCCL_FLOAT64, 0, comm, st);
        if (nrc == 0) nrc = g_rccl.Recv(b, (size_t)count, NCCL_FLOAT64, 0, comm, st);
        const int nrc2 = g_rccl.GroupEnd();
        if (nrc == 0) nrc = nrc2;
        if (nrc != 0) return set_err_text(-1998, std::string("grouped ncclSend/ncclRecv to self: ") + g_rccl.GetErrorString(nrc));
        MCHK(hipStreamSynchronize(st));
        std::vector<double> g((size_t)count + 8);
        MCHK(hipMemcpy(g.data(), b, sizeof(double) * (size_t)(count + 8), hipMemcpyDeviceToHost));
        double worst = 0;
        for (int64_t i = 0; i < count; ++i) {
            const double e = std::fabs(g[(size_t)i] - h[(size_t)i]);
            worst = (e == e) ? std::max(worst, e) : 1e300;  // NaN = element never written
        }
        for (int64_t i = count; i < count + 8; ++i)
            if (g[(size_t)i] == g[(size_t)i]) worst = 1e300;  // wrote past the count: the element type is wider than 8 bytes
        if (max_abs_err_out) *max_abs_err_out = worst;
        return 0;
    }();
    if (st) (void)hipStreamDestroy(st);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    {
        std::lock_guard<std::mutex> l(g_rccl_mu);
        (void)g_rccl.CommDestroy(comm);
        --g_rccl.live;
    }
    return rc;
}

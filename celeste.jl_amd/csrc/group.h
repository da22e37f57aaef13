// group.h -- one process, N devices: the source-partition loop of the reference behind the C ABI.
//
// The reference drains one source list with N workers inside one process (one_node_single_infer, ParallelRun.jl:546-607;
// process_sources_dynamic!, :302-369).  A celeste_group_t is that loop over HIP devices: the images are replicated on every
// member device (one celeste_ctx_t each), a call's targets are cost-sharded over the members (estimate_time = sum of active
// pixels, ParallelRun.jl:45-56, longest first onto the least loaded member), every member runs on its own worker thread
// and stream, and the per-source results are exchanged with ONE ncclAllGather (RCCL over xGMI) on communicators from
// ncclCommInitAll -- the "catalog gather", the only exchange of the path.  Included by celeste_abi.hip (one translation unit).
//
// Exchange modes: RCCL when the members sit on distinct devices (also a group of one: the collective then runs with one
// rank); PEER -- plain hipMemcpyAsync between the members' buffers -- when a device appears twice (RCCL refuses duplicate
// devices in one communicator).  PEER exists so that the shard / gather bookkeeping can be exercised on a one-GPU box; both
// modes fill the same buffers with the same bytes.  CELESTE_GROUP_EXCHANGE=rccl forces the RCCL branch whatever the devices: real
// RCCL then refuses a repeated device at ncclCommInitAll; tests/fake_rccl.c (preloaded, strict host rendezvous) accepts it, which
// is how the RCCL branch's multi-rank ordering is executed on a one-GPU box (tests/test_gpu_group_rccl_branch.py).
//
// Failure protocol (who waits where is in INTEGRATION.md section 3):
//   * per-source failures are statuses, never errors of a member;
//   * a member whose LAUNCH fails still enqueues every collective of the call (the others' results are sound) and the call
//     returns its error;
//   * a member that leaves a call WITHOUT a collective the others enqueue (a HIP call failed in front of it) would leave their
//     streams inside an all-gather that never completes.  Every member counts the collectives it has enqueued; the dispatcher
//     knows how many the dispatched work holds; a member that returns an error short of that count raises `need_abort`, and
//     whoever joins the workers (every entry point does) aborts ALL communicators (ncclCommAbort), which releases the streams.
//     The group is then `broken`: every later call returns CELESTE_ERR_ABORTED until it is destroyed and created again;
//   * a stream that holds a collective is waited for by polling, with ncclCommGetAsyncError consulted between naps and a
//     time limit (CELESTE_GROUP_TIMEOUT_MS, default 300 000, 0 = none): an asynchronous RCCL error or the limit aborts as above.
#pragma once
#include <rccl/rccl.h>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#define GROUP_MAX_MEMBERS 16
enum { GROUP_EXCHANGE_RCCL = 1, GROUP_EXCHANGE_PEER = 2 };

#define GROUP_ROW (CEL_P + 5)      // optimiser exchange row: {target, iterations, f_evals, elbo, status, theta[44]}
#define GROUP_JROW (CEL_P + 1)     // joint exchange row: {target, theta[44]}

// rows of the member's own targets -> its exchange block
__global__ void group_pack_kernel(const double *__restrict__ vp, const int32_t *__restrict__ targets, int n,
                                  const int32_t *__restrict__ it, const int32_t *__restrict__ ev, const double *__restrict__ el,
                                  const int32_t *__restrict__ st, double *__restrict__ block, int row) {
    const int i = blockIdx.x, k = threadIdx.x;
    if (i >= n) return;
    const int t = targets[i];
    double *out = block + (size_t)i * row;
    const int head = row - CEL_P;
    if (k == 0) {
        out[0] = (double)t;
        if (head > 1) { out[1] = it ? (double)it[i] : 0.0; out[2] = ev ? (double)ev[i] : 0.0; out[3] = el ? el[i] : 0.0; out[4] = st ? (double)st[i] : 0.0; }
    }
    if (k < CEL_P) out[head + k] = vp[(size_t)t * CEL_P + k];
}

// the other members' rows -> this member's table (its own rows are already there)
struct GroupCounts { int n[GROUP_MAX_MEMBERS]; };
__global__ void group_scatter_kernel(double *__restrict__ vp, const double *__restrict__ gathered, GroupCounts cnt, int n_members,
                                     int self, int width, int row) {
    const int j = blockIdx.y, i = blockIdx.x, k = threadIdx.x;
    if (j == self || j >= n_members || i >= cnt.n[j] || k >= CEL_P) return;
    const double *in = gathered + ((size_t)j * width + i) * row;
    const int t = (int)in[0];
    if (t < 0) return;          // (an unused row: the member packed fewer rows than its shard holds -- its launch failed)
    vp[(size_t)t * CEL_P + k] = in[row - CEL_P + k];
}

struct GroupMember {
    int index = 0, device = 0;
    celeste_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    hipStream_t comm_stream = nullptr;             // the catalog gather of sweep k overlaps the kernels of sweep k + 1
    hipEvent_t done[2] = {}, buf_free[2] = {}, t0 = nullptr, t1 = nullptr, t2 = nullptr;
    // device
    double *d_vp = nullptr, *d_vp_nbr = nullptr, *d_entry = nullptr;   // S x 44 tables
    int32_t *d_targets = nullptr, *d_idx = nullptr;     // the shard's targets; their positions in the caller's list
    double *d_pos = nullptr;
    int32_t *d_it = nullptr, *d_ev = nullptr, *d_st = nullptr;
    double *d_el = nullptr;
    size_t tgt_cap = 0;
    double *d_block[2] = {nullptr, nullptr}, *d_gathered = nullptr, *d_h = nullptr;
    size_t block_cap = 0, gathered_cap = 0, h_cap = 0;
    // page-locked host
    double *p_gathered = nullptr, *p_h = nullptr;
    size_t p_gathered_cap = 0, p_h_cap = 0;
    // this call's shard
    std::vector<int32_t> idx;        // positions in the caller's target list, ascending
    std::vector<int32_t> tg;         // the targets themselves
    int64_t n_chunks = 0;
    // worker thread
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> task;
    bool has_task = false, quit = false;
    std::atomic<bool> busy{false};              // (written under mu; read without it by the abort check)
    int result = CELESTE_OK;
    // collectives: enqueued so far on this member's communicator; inside the enqueue call right now
    std::atomic<uint64_t> enq{0};
    std::atomic<int> in_coll{0}, in_wait{0};    // ... waiting for a stream that holds one
};

struct celeste_group {
    int n = 0, n_devices = 0, exchange = 0, rccl_ranks = 0;
    int S = 0;
    std::vector<GroupMember *> mem;
    std::vector<celeste_images *> images;      // one handle per distinct device
    std::vector<int64_t> cost;                 // per source: estimate_time
    double *p_vp = nullptr, *p_vp_nbr = nullptr;   // page-locked S x 44 (portable: every member DMAs from it)
    bool threads = false;
    // peer-mode barrier
    std::mutex bmu; std::condition_variable bcv; int b_count = 0; uint64_t b_gen = 0; bool b_abort = false;
    // the planned sweep
    int32_t plan_n = 0; uint32_t plan_flags = 0; int plan_width = 0; size_t plan_blk = 0; uint64_t sweep_k = 0; bool planned = false;
    bool timing = false;
    std::atomic<int> abort_rc{0};              // joint inference: a member's launch failed -- every member leaves after the exchange
    std::mutex call_mu;                        // one call per group at a time
    // failure protocol (see the head of this file)
    uint64_t enq_expected = 0;                 // collectives every member will have enqueued when the dispatched work is through
    std::atomic<bool> need_abort{false}, force_abort{false}, broken{false};
    int64_t timeout_ms = 300000;
    // fault injection, tests only: CELESTE_GROUP_FAULT="member,site[,nth[,delay_ms]]" -- the member's nth (1-based, default 1)
    // arrival at `site` fails, after delay_ms (so that the others are certainly past the point the test is about): "sweep" = a HIP call in front of a sweep's collective, "rows" = between the host barrier and a row
    // exchange's collective, "barrier" = in front of that barrier, "launch" = the member's own launch (it still takes part)
    int fault_member = -1, fault_site = 0, fault_delay_ms = 0;
    std::atomic<int> fault_nth{0};
};
enum { GROUP_FAULT_SWEEP = 1, GROUP_FAULT_ROWS = 2, GROUP_FAULT_BARRIER = 3, GROUP_FAULT_LAUNCH = 4 };
static bool group_fault(celeste_group *g, const GroupMember *m, int site) {
    if (g->fault_member != m->index || g->fault_site != site) return false;
    if (g->fault_nth.fetch_sub(1) != 1) return false;
    fprintf(stderr, "celeste_mi355x: injected fault (CELESTE_GROUP_FAULT): member %d, site %d\n", m->index, site);
    if (g->fault_delay_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(g->fault_delay_ms));   // (the others get ahead)
    return true;
}

// All members meet here (host side; in front of every row exchange, and in PEER mode behind it as well -- RCCL's collective
// is the second meeting point otherwise).  A member that leaves a call with an error never arrives: group_fail wakes the ones
// that wait, and they leave with an error too instead of hanging.
static bool group_barrier(celeste_group *g) {
    if (g->n <= 1) return true;
    std::unique_lock<std::mutex> lk(g->bmu);
    if (g->b_abort) return false;
    const uint64_t gen = g->b_gen;
    if (++g->b_count == g->n) { g->b_count = 0; ++g->b_gen; g->bcv.notify_all(); return true; }
    g->bcv.wait(lk, [&] { return g->b_gen != gen || g->b_abort; });
    // (a barrier that completed has completed, whatever was flagged afterwards: a member that fails BEHIND it must find the
    // others inside the collective -- the abort protocol's case -- not half of them turned back here)
    return g->b_gen != gen;
}
static void group_fail(celeste_group *g) {
    std::lock_guard<std::mutex> lk(g->bmu);
    g->b_abort = true;
    g->bcv.notify_all();
}
static void group_barrier_reset(celeste_group *g) {
    std::lock_guard<std::mutex> lk(g->bmu);
    g->b_abort = false; g->b_count = 0;
}

static void group_worker(GroupMember *m) {
    (void)hipSetDevice(m->device);
    std::unique_lock<std::mutex> lk(m->mu);
    for (;;) {
        m->cv.wait(lk, [&] { return m->has_task || m->quit; });
        if (m->quit) return;
        std::function<int()> fn = std::move(m->task);
        m->has_task = false;
        lk.unlock();
        int r;
        try { r = fn(); } catch (const std::bad_alloc &) { r = CELESTE_ERR_ALLOC; } catch (...) { r = CELESTE_ERR_HIP; }   // (nothing escapes a worker)
        lk.lock();
        if (m->result == CELESTE_OK) m->result = r;      // (sticky until the next join: sweeps are dispatched back to back)
        m->busy = false;
        m->cv.notify_all();
    }
}

// ---- abort: a member left without a collective the others enqueue (or RCCL reported an asynchronous error, or a wait timed out)
// Called by the thread that holds call_mu (the one that dispatches and joins).  `broken` goes up first, so a member that has
// not reached its enqueue yet skips it (group_exchange); a member that is inside the enqueue gets a moment to leave it (real RCCL
// returns in microseconds; the strict test fake blocks until aborted); then every communicator is aborted, which ends the
// collectives the other members' streams sit in.
static void group_abort_comms(celeste_group *g, const char *why) {
    if (g->broken.exchange(true)) return;
    {   // (the host barriers of the row exchanges too)
        std::lock_guard<std::mutex> lk(g->bmu);
        g->b_abort = true;
        g->bcv.notify_all();
    }
    fprintf(stderr, "celeste_mi355x: device group aborted (%s): communicators torn down, the group must be recreated\n", why);
    for (GroupMember *m : g->mem) {
        if (!m->comm) continue;
        const auto t0 = std::chrono::steady_clock::now();
        while (m->in_coll.load() && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(200))
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        (void)ncclCommAbort(m->comm);
        m->comm = nullptr;
    }
}
// Does the group have to be aborted?  force_abort: yes (an asynchronous RCCL error, a wait that timed out).  need_abort: a
// member returned an error short of the collectives its work holds (from then on it enqueues nothing: group_exchange).  While
// other members still run, that is fatal as soon as one of them is blocked -- inside the enqueue, or waiting for a stream -- on
// a collective BEHIND the last one the returned members enqueued: nobody will ever complete it.  (A member blocked on a
// collective everybody did enqueue is merely waiting; a member that is still computing will reach the host barrier or its own
// collective by itself.)  Once all have returned, it is fatal exactly when their counts differ: equal counts mean everybody
// left in front of the same collective (the host barrier's case, or a member whose launch failed and who told the others),
// and the group stays usable.
static void group_check_abort(celeste_group *g, bool all_idle) {
    if (g->broken.load()) return;
    if (g->force_abort.load()) { group_abort_comms(g, "asynchronous RCCL error or time limit while waiting for a collective"); return; }
    if (!g->need_abort.load()) return;
    if (all_idle) {
        const uint64_t e0 = g->mem[0]->enq.load();
        bool same = true;
        for (GroupMember *m : g->mem) same = same && m->enq.load() == e0;
        if (same) { g->need_abort.store(false); g->enq_expected = e0; return; }
        group_abort_comms(g, "a member left a call without a collective the others enqueued");
        return;
    }
    uint64_t low = ~(uint64_t)0;
    for (GroupMember *m : g->mem) if (!m->busy.load()) low = std::min(low, m->enq.load());
    if (low == ~(uint64_t)0) return;
    for (GroupMember *m : g->mem) {
        if (!m->busy.load()) continue;
        const uint64_t e = m->enq.load();                    // (read before the flags: an enqueue that completes in between
        const bool in_c = m->in_coll.load() != 0, in_w = m->in_wait.load() != 0;   // makes this member look less stuck, never more)
        if ((in_c && e + 1 > low) || (in_w && e > low)) {
            group_abort_comms(g, "a member left a call without its collective; another waits inside that collective");
            return;
        }
    }
}
// the member's worker is idle (its last task has returned); while waiting, serve an abort request -- the task may be one that
// waits for a collective a failed member never enqueued
static void group_wait_idle(celeste_group *g, GroupMember *m, std::unique_lock<std::mutex> &lk) {
    while (m->busy.load()) {
        if ((g->need_abort.load() || g->force_abort.load()) && !g->broken.load()) { lk.unlock(); group_check_abort(g, false); lk.lock(); }
        if (m->busy.load()) m->cv.wait_for(lk, std::chrono::milliseconds(2));
    }
}

// fn(member) on every member, concurrently; dispatch returns at once, join returns the first non-OK status.
// n_collectives: the collectives fn enqueues on the member's communicator when nothing fails.
static void group_dispatch(celeste_group *g, const std::function<int(GroupMember *)> &fn, uint64_t n_collectives = 0) {
    g->enq_expected += n_collectives;
    const uint64_t expected = g->enq_expected;
    for (GroupMember *m : g->mem) {
        if (!g->threads) {
            (void)hipSetDevice(m->device);
            int r;
            try { r = fn(m); } catch (const std::bad_alloc &) { r = CELESTE_ERR_ALLOC; } catch (...) { r = CELESTE_ERR_HIP; }
            if (m->result == CELESTE_OK) m->result = r;
            continue;
        }
        celeste_group *const gg = g;
        std::unique_lock<std::mutex> lk(m->mu);
        group_wait_idle(g, m, lk);
        m->task = [fn, m, gg, expected] {
            int r;
            try { r = fn(m); } catch (const std::bad_alloc &) { r = CELESTE_ERR_ALLOC; } catch (...) { r = CELESTE_ERR_HIP; }
            // (per-source failures are statuses, not errors of the call: the other members go on)
            if (r != CELESTE_OK && r != CELESTE_ERR_NONFINITE_INPUT && r != CELESTE_ERR_NONFINITE_RESULT) {
                group_fail(gg);
                // short of the collectives this work holds: the others' streams will wait for this member for ever
                if (gg->exchange == GROUP_EXCHANGE_RCCL && gg->n > 1 && m->enq.load() < expected) gg->need_abort.store(true);
            }
            return r;
        };
        m->has_task = true; m->busy = true;
        m->cv.notify_all();
    }
}
static int group_join(celeste_group *g) {
    int rc = CELESTE_OK;
    for (GroupMember *m : g->mem) {
        if (g->threads) { std::unique_lock<std::mutex> lk(m->mu); group_wait_idle(g, m, lk); }
        if (m->result != CELESTE_OK && rc == CELESTE_OK) rc = m->result;
        m->result = CELESTE_OK;
    }
    group_check_abort(g, true);      // (every task has returned; streams may still sit in a collective the failed member skipped)
    if (g->broken.load()) rc = CELESTE_ERR_ABORTED;
    return rc;
}
static int group_run(celeste_group *g, const std::function<int(GroupMember *)> &fn, uint64_t n_collectives = 0) {
    group_dispatch(g, fn, n_collectives);
    return group_join(g);
}

// Wait for a stream that may hold a collective.  Plain synchronisation where no other rank can be missing; otherwise poll, look
// at the communicator between naps, and give up at the time limit -- both abort the group (which releases the stream).
// joiner = the thread that holds call_mu waits itself (group_quiesce): it is also the one that has to tear the communicators down
static int group_stream_wait(celeste_group *g, GroupMember *m, hipStream_t s, bool joiner = false) {
    if (g->exchange != GROUP_EXCHANGE_RCCL || g->n <= 1 || g->broken.load() || !m->comm) {
        if (hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); return CELESTE_ERR_HIP; }
        return g->broken.load() ? CELESTE_ERR_ABORTED : CELESTE_OK;
    }
    struct Waiting { std::atomic<int> &f; Waiting(std::atomic<int> &x) : f(x) { f.store(1); } ~Waiting() { f.store(0); } } waiting(m->in_wait);
    const auto t0 = std::chrono::steady_clock::now();
    bool reported = false;
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return g->broken.load() ? CELESTE_ERR_ABORTED : CELESTE_OK;
        if (e != hipErrorNotReady) { (void)hipGetLastError(); return CELESTE_ERR_HIP; }
        if (g->broken.load()) { (void)hipStreamSynchronize(s); (void)hipGetLastError(); return CELESTE_ERR_ABORTED; }   // (aborted: the stream is released)
        const auto el = std::chrono::steady_clock::now() - t0;
        if (el < std::chrono::milliseconds(2)) continue;          // (a sweep is through long before)
        if (!reported) {
            ncclResult_t ae = ncclSuccess;
            ncclComm_t comm = m->comm;
            if (comm && (ncclCommGetAsyncError(comm, &ae) != ncclSuccess || (ae != ncclSuccess && ae != ncclInProgress))) {
                fprintf(stderr, "celeste_mi355x: member %d: asynchronous RCCL error: %s\n", m->index, ncclGetErrorString(ae));
                g->force_abort.store(true); reported = true;
            } else if (g->timeout_ms > 0 && el > std::chrono::milliseconds(g->timeout_ms)) {
                fprintf(stderr, "celeste_mi355x: member %d: a stream holding a collective did not complete within %lld ms\n", m->index,
                        (long long)g->timeout_ms);
                g->force_abort.store(true); reported = true;
            }
        }
        // (a worker keeps waiting until whoever joins the workers has torn the communicators down; the joiner does it itself)
        if (joiner && g->force_abort.load()) group_check_abort(g, true);
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
}

#define NCCL_TRY(expr)                                                                                                   \
    do {                                                                                                                 \
        ncclResult_t r__ = (expr);                                                                                       \
        if (r__ != ncclSuccess) {                                                                                        \
            fprintf(stderr, "celeste_mi355x: %s failed: %s (%s:%d)\n", #expr, ncclGetErrorString(r__), __FILE__, __LINE__); \
            return CELESTE_ERR_HIP;                                                                                      \
        }                                                                                                                \
    } while (0)

template <class T>
static int group_grow(T **p, size_t *cap, size_t n, hipStream_t drain) {
    if (n <= *cap && *p) return CELESTE_OK;
    if (*p) { if (drain) HIP_TRY(hipStreamSynchronize(drain)); (void)hipFree(*p); *p = nullptr; }
    *cap = 0;
    HIP_TRY(hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    *cap = std::max<size_t>(n, 1);
    return CELESTE_OK;
}
template <class T>
static int group_grow_pinned(T **p, size_t *cap, size_t n) {
    if (n <= *cap && *p) return CELESTE_OK;
    if (*p) { (void)hipHostFree(*p); *p = nullptr; }
    *cap = 0;
    HIP_TRY(hipHostMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T), hipHostMallocPortable));
    *cap = std::max<size_t>(n, 1);
    return CELESTE_OK;
}

// the member's block (count doubles at `send`) to slot `index` of every member's gathered buffer, on `stream`
static int group_exchange(celeste_group *g, GroupMember *m, const double *send, size_t count, hipStream_t stream) {
    if (g->exchange == GROUP_EXCHANGE_RCCL) {
        ncclComm_t comm = m->comm;
        m->in_coll.store(1);
        // (need_abort: some member has left short of a collective -- pairing further ones up with the others' would hand them the
        // wrong blocks; whoever joins the workers decides whether the group survives)
        if (g->broken.load() || !comm) { m->in_coll.store(0); return CELESTE_ERR_ABORTED; }
        if (g->need_abort.load()) { m->in_coll.store(0); return CELESTE_ERR_HIP; }
        const ncclResult_t r = ncclAllGather(send, m->d_gathered, count, ncclDouble, comm, stream);
        m->in_coll.store(0);
        if (r != ncclSuccess) {
            if (g->broken.load()) return CELESTE_ERR_ABORTED;
            fprintf(stderr, "celeste_mi355x: member %d: ncclAllGather failed: %s\n", m->index, ncclGetErrorString(r));
            return CELESTE_ERR_HIP;
        }
        m->enq.fetch_add(1);
        return CELESTE_OK;
    }
    for (GroupMember *o : g->mem)
        HIP_TRY(hipMemcpyAsync(o->d_gathered + (size_t)m->index * count, send, count * sizeof(double), hipMemcpyDefault, stream));
    m->enq.fetch_add(1);
    return CELESTE_OK;
}

extern "C" void celeste_group_destroy(celeste_group_t *g) {
    if (!g) return;
    for (GroupMember *m : g->mem) {
        if (!m) continue;
        if (m->th.joinable()) {
            { std::unique_lock<std::mutex> lk(m->mu); group_wait_idle(g, m, lk); m->quit = true; m->cv.notify_all(); }
            m->th.join();
        }
        (void)hipSetDevice(m->device);
        if (m->ctx && m->ctx->stream) (void)hipStreamSynchronize(m->ctx->stream);
        if (m->comm_stream) (void)hipStreamSynchronize(m->comm_stream);
        if (m->comm) (void)ncclCommDestroy(m->comm);
        void *dp[] = {m->d_vp, m->d_vp_nbr, m->d_entry, m->d_targets, m->d_idx, m->d_pos, m->d_it, m->d_ev, m->d_st, m->d_el, m->d_block[0],
                      m->d_block[1], m->d_gathered, m->d_h};
        for (void *p : dp) if (p) (void)hipFree(p);
        if (m->p_gathered) (void)hipHostFree(m->p_gathered);
        if (m->p_h) (void)hipHostFree(m->p_h);
        hipEvent_t evs[] = {m->done[0], m->done[1], m->buf_free[0], m->buf_free[1], m->t0, m->t1, m->t2};
        for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
        stream_retire(m->device, m->comm_stream);   // (recycled, not destroyed: see stream_retire)
        if (m->ctx) celeste_ctx_destroy(m->ctx);
        delete m;
    }
    for (celeste_images *im : g->images) images_release(im);
    if (g->p_vp) (void)hipHostFree(g->p_vp);
    if (g->p_vp_nbr) (void)hipHostFree(g->p_vp_nbr);
    delete g;
}

extern "C" int celeste_group_create(const celeste_problem_t *pr, int32_t n_members, const int32_t *devices, celeste_group_t **out) try {
    if (!pr || !out || n_members < 1 || n_members > GROUP_MAX_MEMBERS) return CELESTE_ERR_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); return CELESTE_ERR_NO_DEVICE; }
    std::vector<int> dev((size_t)n_members);
    for (int r = 0; r < n_members; ++r) {
        dev[r] = devices ? devices[r] : r;
        if (dev[r] < 0 || dev[r] >= count) return CELESTE_ERR_INVALID_ARG;
    }
    celeste_group *g = new (std::nothrow) celeste_group();
    if (!g) return CELESTE_ERR_ALLOC;
    g->n = n_members; g->S = pr->n_sources;
    std::vector<int> distinct;
    for (int d : dev) if (std::find(distinct.begin(), distinct.end(), d) == distinct.end()) distinct.push_back(d);
    g->n_devices = (int)distinct.size();
    g->exchange = g->n_devices == g->n ? GROUP_EXCHANGE_RCCL : GROUP_EXCHANGE_PEER;
    if (const char *e = getenv("CELESTE_GROUP_EXCHANGE")) {   // "peer": hipMemcpyAsync between the members instead of RCCL (A/B, debugging);
        if (strcmp(e, "peer") == 0) g->exchange = GROUP_EXCHANGE_PEER;      // "rccl": the RCCL branch whatever the devices (real RCCL
        else if (strcmp(e, "rccl") == 0) g->exchange = GROUP_EXCHANGE_RCCL; // refuses repeated devices; the tests' stand-in does not)
    }
    if (const char *e = getenv("CELESTE_GROUP_TIMEOUT_MS")) g->timeout_ms = atoll(e);
    if (const char *e = getenv("CELESTE_GROUP_FAULT")) {      // tests only (see celeste_group)
        int mem = -1, nth = 1, delay = 0; char site[16] = {0};
        if (sscanf(e, "%d,%15[a-z],%d,%d", &mem, site, &nth, &delay) >= 2) {
            g->fault_member = mem; g->fault_nth.store(nth); g->fault_delay_ms = delay;
            g->fault_site = !strcmp(site, "sweep") ? GROUP_FAULT_SWEEP : !strcmp(site, "rows") ? GROUP_FAULT_ROWS
                          : !strcmp(site, "barrier") ? GROUP_FAULT_BARRIER : !strcmp(site, "launch") ? GROUP_FAULT_LAUNCH : 0;
        }
    }
    g->threads = g->n > 1 || (getenv("CELESTE_GROUP_THREADS") && atoi(getenv("CELESTE_GROUP_THREADS")) == 1);
#define GR_TRY(expr) do { int s__ = (expr); if (s__ != CELESTE_OK) { celeste_group_destroy(g); return s__; } } while (0)
#define GR_HIP(expr) do { if ((expr) != hipSuccess) { (void)hipGetLastError(); celeste_group_destroy(g); return CELESTE_ERR_HIP; } } while (0)
    // the planes once per distinct device; every member gets its own context (tables, scratch, stream) on them
    for (int d : distinct) {
        celeste_images *im = nullptr;
        GR_TRY(celeste_images_create(pr->n_images, pr->images, d, &im));
        g->images.push_back(im);
    }
    for (int r = 0; r < g->n; ++r) {
        GroupMember *m = new (std::nothrow) GroupMember();
        if (!m) { celeste_group_destroy(g); return CELESTE_ERR_ALLOC; }
        m->index = r; m->device = dev[r];
        g->mem.push_back(m);
    }
    if (g->threads) for (GroupMember *m : g->mem) m->th = std::thread(group_worker, m);
    {
        // the contexts in parallel (spline prefilter, work-item lists: host time per member)
        int rc = group_run(g, [&](GroupMember *m) -> int {
            const size_t k = (size_t)(std::find(distinct.begin(), distinct.end(), m->device) - distinct.begin());
            int st = celeste_ctx_create_on(g->images[k], pr, &m->ctx);
            if (st != CELESTE_OK) return st;
            HIP_TRY(hipSetDevice(m->device));
            HIP_TRY(stream_acquire(m->device, &m->comm_stream));
            for (int k2 = 0; k2 < 2; ++k2) {
                HIP_TRY(hipEventCreateWithFlags(&m->done[k2], hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&m->buf_free[k2], hipEventDisableTiming));
            }
            HIP_TRY(hipEventCreate(&m->t0)); HIP_TRY(hipEventCreate(&m->t1)); HIP_TRY(hipEventCreate(&m->t2));
            const size_t tb = (size_t)m->ctx->S * CEL_P * sizeof(double);
            HIP_TRY(hipMalloc((void **)&m->d_vp, tb));
            HIP_TRY(hipMalloc((void **)&m->d_vp_nbr, tb));
            HIP_TRY(hipMalloc((void **)&m->d_entry, tb));
            return CELESTE_OK;
        });
        if (rc != CELESTE_OK) { celeste_group_destroy(g); return rc; }
    }
    if (g->exchange == GROUP_EXCHANGE_RCCL) {
        std::vector<ncclComm_t> comms((size_t)g->n);
        ncclResult_t r = ncclCommInitAll(comms.data(), g->n, dev.data());
        if (r != ncclSuccess) {
            fprintf(stderr, "celeste_mi355x: ncclCommInitAll over %d device(s) failed: %s\n", g->n, ncclGetErrorString(r));
            celeste_group_destroy(g); return CELESTE_ERR_HIP;
        }
        for (int k = 0; k < g->n; ++k) g->mem[k]->comm = comms[k];
        if (ncclCommCount(comms[0], &g->rccl_ranks) != ncclSuccess) g->rccl_ranks = 0;
    } else if (g->n_devices > 1) {
        for (GroupMember *m : g->mem) {   // PEER between distinct devices: direct access where the fabric offers it
            GR_HIP(hipSetDevice(m->device));
            for (int d : distinct) if (d != m->device) { int can = 0; if (hipDeviceCanAccessPeer(&can, m->device, d) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(d, 0); (void)hipGetLastError(); }
        }
    }
    // estimate_time per source (ParallelRun.jl:45-47): the pixels of its patches
    {
        const celeste_ctx *c = g->mem[0]->ctx;
        g->cost.assign((size_t)c->S, 0);
        for (int s = 0; s < c->S; ++s)
            for (int v = c->h_vis_off[s]; v < c->h_vis_off[s + 1]; ++v) g->cost[s] += (int64_t)c->h_patches[v].H2 * c->h_patches[v].W2;
    }
    const size_t tn = (size_t)g->S * CEL_P;
    GR_HIP(hipHostMalloc((void **)&g->p_vp, tn * sizeof(double), hipHostMallocPortable));
    GR_HIP(hipHostMalloc((void **)&g->p_vp_nbr, tn * sizeof(double), hipHostMallocPortable));
#undef GR_TRY
#undef GR_HIP
    *out = g;
    return CELESTE_OK;
} ABI_CATCH

extern "C" int celeste_group_info(celeste_group_t *g, celeste_group_info_t *out) try {
    if (!g || !out) return CELESTE_ERR_INVALID_ARG;
    memset(out, 0, sizeof *out);
    out->n_members = g->n; out->n_devices = g->n_devices; out->exchange = g->exchange; out->rccl_ranks = g->rccl_ranks;
    for (int r = 0; r < g->n; ++r) out->devices[r] = g->mem[r]->device;
    return CELESTE_OK;
} ABI_CATCH

// enqueued[n_members] (may be NULL): the collectives (catalog gathers, row exchanges) each member has enqueued since the group
// was created -- equal on every member whenever no call is in flight; *aborted (may be NULL): 1 = the communicators were torn
// down (CELESTE_ERR_ABORTED from every entry point)
extern "C" int celeste_group_collectives(celeste_group_t *g, int64_t *enqueued, int32_t *aborted) try {
    if (!g) return CELESTE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(g->call_mu);
    if (enqueued) for (int r = 0; r < g->n; ++r) enqueued[r] = (int64_t)g->mem[r]->enq.load();
    if (aborted) *aborted = g->broken.load() ? 1 : 0;
    return CELESTE_OK;
} ABI_CATCH

// ---- sharding: longest processing time first onto the least loaded member (partition.shard_targets) ----------------
// weight[i] of unit i; returns for every member the ascending list of unit indices
static void group_lpt(int n_members, const std::vector<int64_t> &weight, std::vector<std::vector<int32_t>> &shards) {
    const size_t n = weight.size();
    if (n_members == 1) {      // (nothing to balance)
        shards.assign(1, std::vector<int32_t>(n));
        for (size_t i = 0; i < n; ++i) shards[0][i] = (int32_t)i;
        return;
    }
    std::vector<int32_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = (int32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return weight[a] > weight[b]; });
    std::vector<int64_t> load((size_t)n_members, 0);
    shards.assign((size_t)n_members, {});
    for (int32_t i : order) {
        int k = 0;
        for (int p = 1; p < n_members; ++p) if (load[p] < load[k]) k = p;
        shards[k].push_back(i);
        load[k] += weight[i];
    }
    for (auto &s : shards) std::sort(s.begin(), s.end());
}

static int group_check_targets(celeste_group *g, int32_t n, const int32_t *targets) {
    if (n < 0 || (n > 0 && !targets)) return CELESTE_ERR_INVALID_ARG;
    for (int t = 0; t < n; ++t) if (targets[t] < 0 || targets[t] >= g->S) return CELESTE_ERR_INVALID_ARG;
    return CELESTE_OK;
}

static void group_shard(celeste_group *g, int32_t n, const int32_t *targets) {
    std::vector<int64_t> w((size_t)n);
    for (int t = 0; t < n; ++t) w[t] = g->cost[targets[t]];
    std::vector<std::vector<int32_t>> shards;
    group_lpt(g->n, w, shards);
    for (int r = 0; r < g->n; ++r) {
        GroupMember *m = g->mem[r];
        m->idx.swap(shards[r]);
        m->tg.resize(m->idx.size());
        m->n_chunks = 0;
        for (size_t i = 0; i < m->idx.size(); ++i) { m->tg[i] = targets[m->idx[i]]; m->n_chunks += m->ctx->h_src_chunks[m->tg[i]]; }
    }
}

// per-target device buffers of a member at capacity >= n
static int group_target_buffers(GroupMember *m, size_t n) {
    if (n <= m->tgt_cap && m->d_targets) return CELESTE_OK;
    HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    HIP_TRY(hipStreamSynchronize(m->comm_stream));
    void **ps[] = {(void **)&m->d_targets, (void **)&m->d_pos, (void **)&m->d_it, (void **)&m->d_ev, (void **)&m->d_st, (void **)&m->d_el,
                   (void **)&m->d_idx};
    const size_t by[] = {sizeof(int32_t), 2 * sizeof(double), sizeof(int32_t), sizeof(int32_t), sizeof(int32_t), sizeof(double), sizeof(int32_t)};
    m->tgt_cap = 0;
    for (int k = 0; k < 7; ++k) {
        if (*ps[k]) { (void)hipFree(*ps[k]); *ps[k] = nullptr; }
        HIP_TRY(hipMalloc(ps[k], std::max<size_t>(n, 1) * by[k]));
    }
    m->tgt_cap = std::max<size_t>(n, 1);
    return CELESTE_OK;
}

// ---- the sweep: every member evaluates its shard, then the catalog gather ------------------------------------------
// block of a member, width W: v[W] | d[W x 44] | counters[W x 2] (int64) | status[W] (int32, padded to doubles)
static inline size_t sweep_block_doubles(int W) { return (size_t)W * (1 + CEL_P + 2) + ((size_t)W + 1) / 2; }

// nothing of an earlier call is in flight on any member
static int group_quiesce(celeste_group *g) {
    int rc = group_join(g);
    for (GroupMember *m : g->mem) {
        if (!m->ctx) continue;
        int s1 = CELESTE_OK;
        if (hipSetDevice(m->device) != hipSuccess) { (void)hipGetLastError(); s1 = CELESTE_ERR_HIP; }
        if (s1 == CELESTE_OK) s1 = group_stream_wait(g, m, m->comm_stream, true);
        if (s1 == CELESTE_OK || s1 == CELESTE_ERR_ABORTED) { const int s2 = group_stream_wait(g, m, m->ctx->stream, true); if (s1 == CELESTE_OK) s1 = s2; }
        if (m->ctx->copy_stream && hipStreamSynchronize(m->ctx->copy_stream) != hipSuccess) { (void)hipGetLastError(); if (s1 == CELESTE_OK) s1 = CELESTE_ERR_HIP; }
        if (s1 != CELESTE_OK && rc == CELESTE_OK) rc = s1;
    }
    if (g->broken.load()) rc = CELESTE_ERR_ABORTED;
    return rc;
}

// (call_mu held)
// sync = false: the uploads stay in flight on the members' streams (the caller enqueues its work behind them and waits then)
static int group_plan_locked(celeste_group *g, const double *vp, int32_t n_targets, const int32_t *targets, uint32_t flags, bool sync = true) {
    g->planned = false;
    (void)group_quiesce(g);     // (sweeps of an earlier plan may still be in flight: their buffers are about to move)
    if (g->broken.load()) return CELESTE_ERR_ABORTED;
    group_barrier_reset(g);
    group_shard(g, n_targets, targets);
    int W = 1;
    for (GroupMember *m : g->mem) W = std::max(W, (int)m->idx.size());
    g->plan_n = n_targets; g->plan_flags = flags; g->plan_width = W; g->plan_blk = sweep_block_doubles(W); g->sweep_k = 0;
    memcpy(g->p_vp, vp, (size_t)g->S * CEL_P * sizeof(double));
    const bool want_h = (flags & CELESTE_FLAG_HESS) != 0;
    const size_t HS = (flags & CELESTE_FLAG_PACKED_HESS) ? CELESTE_HP : (size_t)CEL_P * CEL_P;
    int rc = group_run(g, [&](GroupMember *m) -> int {
        HIP_TRY(hipSetDevice(m->device));
        hipStream_t st = m->ctx->stream;
        int s1 = group_target_buffers(m, m->tg.size());
        if (s1 != CELESTE_OK) return s1;
        const size_t blk = g->plan_blk;
        if (blk > m->block_cap) {
            HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipStreamSynchronize(m->comm_stream));
            for (int k = 0; k < 2; ++k) { if (m->d_block[k]) (void)hipFree(m->d_block[k]); m->d_block[k] = nullptr; }
            m->block_cap = 0;
            for (int k = 0; k < 2; ++k) HIP_TRY(hipMalloc((void **)&m->d_block[k], blk * sizeof(double)));
            m->block_cap = blk;
        }
        for (int k = 0; k < 2; ++k) HIP_TRY(hipMemsetAsync(m->d_block[k], 0, blk * sizeof(double), st));
        s1 = group_grow(&m->d_gathered, &m->gathered_cap, blk * g->n, m->comm_stream);
        if (s1 == CELESTE_OK && want_h) s1 = group_grow(&m->d_h, &m->h_cap, std::max<size_t>(m->tg.size(), 1) * HS, st);
        if (s1 != CELESTE_OK) return s1;
        HIP_TRY(hipMemcpyAsync(m->d_vp, g->p_vp, (size_t)g->S * CEL_P * sizeof(double), hipMemcpyHostToDevice, st));
        if (!m->tg.empty()) {
            HIP_TRY(hipMemcpyAsync(m->d_targets, m->tg.data(), m->tg.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(m->d_idx, m->idx.data(), m->idx.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
        }
        for (int k = 0; k < 2; ++k) HIP_TRY(hipEventRecord(m->buf_free[k], st));
        if (sync) HIP_TRY(hipStreamSynchronize(st));
        return CELESTE_OK;
    });
    g->planned = rc == CELESTE_OK;
    return rc;
}

extern "C" int celeste_group_sweep_plan(celeste_group_t *g, const double *vp, int32_t n_targets, const int32_t *targets,
                                        uint32_t flags) try {
    if (!g || !vp || n_targets < 1 || (flags & (CELESTE_FLAG_SPLIT))) return CELESTE_ERR_INVALID_ARG;
    if (group_check_targets(g, n_targets, targets) != CELESTE_OK) return CELESTE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(g->call_mu);
    return group_plan_locked(g, vp, n_targets, targets, flags);
} ABI_CATCH

// One sweep of the member's shard into block k, then the catalog gather on the second stream.  Whatever fails in front of the
// collective, the collective is still attempted: a member that cannot enqueue it is what the abort protocol is for.
static int group_sweep_member(celeste_group *g, GroupMember *m, int k) {
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = m->ctx->stream;
    const int W = g->plan_width;
    double *blk = m->d_block[k];
    HIP_TRY(hipStreamWaitEvent(st, m->buf_free[k], 0));      // the gather that last read this block is through
    if (g->timing) HIP_TRY(hipEventRecord(m->t0, st));
    int own_rc = CELESTE_OK;
    if (group_fault(g, m, GROUP_FAULT_LAUNCH)) own_rc = CELESTE_ERR_HIP;
    else if (!m->tg.empty())
        own_rc = launch_eval(m->ctx, m->d_vp, (int32_t)m->tg.size(), m->d_targets, g->plan_flags, blk, blk + W, m->d_h,
                             reinterpret_cast<int64_t *>(blk + (size_t)W * (1 + CEL_P)),
                             reinterpret_cast<int32_t *>(blk + (size_t)W * (1 + CEL_P + 2)), st, true, nullptr, m->n_chunks);
    // (a member whose launch failed still takes part in the gather -- the collective needs every rank, and the others' sweeps
    // are sound; the call reports this member's error)
    if (g->timing) HIP_TRY(hipEventRecord(m->t1, st));
    if (group_fault(g, m, GROUP_FAULT_SWEEP)) return CELESTE_ERR_HIP;
    HIP_TRY(hipEventRecord(m->done[k], st));
    HIP_TRY(hipStreamWaitEvent(m->comm_stream, m->done[k], 0));
    int rc = group_exchange(g, m, blk, g->plan_blk, m->comm_stream);
    if (rc != CELESTE_OK) return rc;
    HIP_TRY(hipEventRecord(m->buf_free[k], m->comm_stream));
    if (g->timing) HIP_TRY(hipEventRecord(m->t2, m->comm_stream));
    return own_rc;
}

extern "C" int celeste_group_sweep(celeste_group_t *g) try {
    if (!g) return CELESTE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(g->call_mu);
    if (g->broken.load()) return CELESTE_ERR_ABORTED;
    if (!g->planned) return CELESTE_ERR_INVALID_ARG;
    const int k = (int)(g->sweep_k++ & 1);
    group_dispatch(g, [g, k](GroupMember *m) -> int { return group_sweep_member(g, m, k); }, 1);
    if (!g->threads) return group_join(g);
    return g->broken.load() ? CELESTE_ERR_ABORTED : CELESTE_OK;     // (errors of the workers surface in celeste_group_sweep_wait)
} ABI_CATCH

extern "C" int celeste_group_sweep_wait(celeste_group_t *g) try {
    if (!g) return CELESTE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(g->call_mu);
    // every member's kernels and copies: in PEER mode a member's gathered buffer is written by the OTHER members' streams
    return group_quiesce(g);
} ABI_CATCH

// the catalog block member 0 brought down -> the caller's arrays, in the caller's order; returns the first non-OK status
static int group_catalog_to_host(celeste_group *g, const double *p_gathered, uint32_t flags, double *v, double *d, int64_t *counters,
                                 int32_t *status) {
    const int W = g->plan_width;
    const size_t blk = g->plan_blk;
    const bool want_d = d && (flags & (CELESTE_FLAG_GRAD | CELESTE_FLAG_HESS));
    int worst = CELESTE_OK;
    for (int r = 0; r < g->n; ++r) {
        const GroupMember *o = g->mem[r];
        const double *b = p_gathered + (size_t)r * blk;
        const int64_t *bc = reinterpret_cast<const int64_t *>(b + (size_t)W * (1 + CEL_P));
        const int32_t *bs = reinterpret_cast<const int32_t *>(b + (size_t)W * (1 + CEL_P + 2));
        for (size_t i = 0; i < o->idx.size(); ++i) {
            const size_t q = (size_t)o->idx[i];
            if (v) v[q] = b[i];
            if (want_d) memcpy(d + q * CEL_P, b + W + i * CEL_P, CEL_P * sizeof(double));
            if (counters) { counters[2 * q] = bc[2 * i]; counters[2 * q + 1] = bc[2 * i + 1]; }
            if (status) status[q] = bs[i];
            if (bs[i] != CELESTE_OK && worst == CELESTE_OK) worst = bs[i];
        }
    }
    return worst;
}

// a shard's Hessians, rows [lo, hi) of the member's d_h, down on `stream`: straight into the caller's array when that is
// page-locked and the shard is a run of consecutive positions (always, for a group of one), else into page-locked staging
static inline bool shard_contiguous(const GroupMember *m) { return !m->idx.empty() && (size_t)(m->idx.back() - m->idx.front()) + 1 == m->idx.size(); }
static int group_hessians_down(GroupMember *m, double *h, bool direct, size_t HS, size_t lo, size_t hi, hipStream_t stream) {
    if (hi <= lo) return CELESTE_OK;
    double *dst = direct ? h + ((size_t)m->idx[0] + lo) * HS : m->p_h + lo * HS;
    HIP_TRY(hipMemcpyAsync(dst, m->d_h + lo * HS, (hi - lo) * HS * sizeof(double), hipMemcpyDeviceToHost, stream));
    return CELESTE_OK;
}
static inline void group_hessians_scatter(const GroupMember *m, double *h, size_t HS, size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) memcpy(h + (size_t)m->idx[i] * HS, m->p_h + i * HS, HS * sizeof(double));
}

// (call_mu held; every sweep and gather is through)
static int group_results_locked(celeste_group *g, double *v, double *d, double *h, int64_t *counters, int32_t *status) {
    const size_t blk = g->plan_blk;
    const uint32_t flags = g->plan_flags;
    const bool want_h = h && (flags & CELESTE_FLAG_HESS);
    const size_t HS = (flags & CELESTE_FLAG_PACKED_HESS) ? CELESTE_HP : (size_t)CEL_P * CEL_P;
    const bool pin_h = want_h && is_pinned(h, (size_t)g->plan_n * HS * sizeof(double));
    int worst = CELESTE_OK;
    int rc = group_run(g, [&](GroupMember *m) -> int {
        HIP_TRY(hipSetDevice(m->device));
        hipStream_t st = m->ctx->stream;
        // the catalog (values, gradients, counters, status of ALL targets) is on every member: member 0 hands it to the host;
        // Hessians stay with the member that owns the target and come down from there, all members at once, in parts whose
        // host-side scatter overlaps the copies of the parts behind them
        if (m->index == 0) {
            int s1 = group_grow_pinned(&m->p_gathered, &m->p_gathered_cap, blk * g->n);
            if (s1 != CELESTE_OK) return s1;
            HIP_TRY(hipMemcpyAsync(m->p_gathered, m->d_gathered, blk * g->n * sizeof(double), hipMemcpyDeviceToHost, st));
        }
        const size_t nr = m->tg.size();
        const bool direct = pin_h && shard_contiguous(m);
        int n_parts = 0;
        size_t part_lo[celeste_ctx::MAX_PARTS + 1] = {0};
        if (want_h && nr > 0) {
            if (!direct) { int s1 = group_grow_pinned(&m->p_h, &m->p_h_cap, nr * HS); if (s1 != CELESTE_OK) return s1; }
            n_parts = direct ? 1 : (int)std::min<size_t>(celeste_ctx::MAX_PARTS, std::max<size_t>(1, nr / 192));
            for (int k = 0; k <= n_parts; ++k) part_lo[k] = nr * (size_t)k / (size_t)n_parts;
            { int s1 = ensure_part_events(m->ctx, n_parts); if (s1 != CELESTE_OK) return s1; }
            for (int k = 0; k < n_parts; ++k) {
                int s1 = group_hessians_down(m, h, direct, HS, part_lo[k], part_lo[k + 1], st);
                if (s1 != CELESTE_OK) { (void)hipStreamSynchronize(st); return s1; }
                if (hipEventRecord(m->ctx->part_copied[k], st) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(st); return CELESTE_ERR_HIP; }
            }
        }
        for (int k = 0; k < n_parts && !direct; ++k) {
            if (hipEventSynchronize(m->ctx->part_copied[k]) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(st); return CELESTE_ERR_HIP; }
            group_hessians_scatter(m, h, HS, part_lo[k], part_lo[k + 1]);
        }
        HIP_TRY(hipStreamSynchronize(st));
        if (m->index == 0) worst = group_catalog_to_host(g, m->p_gathered, flags, v, d, counters, status);
        return CELESTE_OK;
    });
    return rc != CELESTE_OK ? rc : worst;
}

extern "C" int celeste_group_sweep_results(celeste_group_t *g, double *v, double *d, double *h, int64_t *counters, int32_t *status) try {
    if (!g) return CELESTE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(g->call_mu);
    if (g->broken.load()) return CELESTE_ERR_ABORTED;
    if (!g->planned || g->sweep_k == 0) return CELESTE_ERR_INVALID_ARG;
    int rc = group_quiesce(g);
    if (rc != CELESTE_OK) return rc;
    return group_results_locked(g, v, d, h, counters, status);
} ABI_CATCH

extern "C" int celeste_group_enable_timing(celeste_group_t *g, int enable) try {
    if (!g) return CELESTE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(g->call_mu);
    (void)group_quiesce(g);
    g->timing = enable != 0;
    for (GroupMember *m : g->mem) (void)celeste_ctx_enable_timing(m->ctx, enable);
    return CELESTE_OK;
} ABI_CATCH

// celeste_ctx_last_kernel_ms of one member's last launch chain (prep / pixel / lift)
extern "C" int celeste_group_last_kernel_ms(celeste_group_t *g, int32_t member, float ms[3]) try {
    if (!g || member < 0 || member >= g->n || !ms) return CELESTE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(g->call_mu);
    HIP_TRY(hipSetDevice(g->mem[member]->device));
    return celeste_ctx_last_kernel_ms(g->mem[member]->ctx, ms);
} ABI_CATCH

// HIP-event durations of the last sweep, per member: eval_ms[r] = the member's launch chain (its shard), gather_ms[r] = from
// the end of its chain to the end of its catalog gather.  After celeste_group_sweep_wait.
extern "C" int celeste_group_last_sweep_ms(celeste_group_t *g, float *eval_ms, float *gather_ms) try {
    if (!g) return CELESTE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(g->call_mu);
    if (g->broken.load()) return CELESTE_ERR_ABORTED;
    if (!g->timing || g->sweep_k == 0) return CELESTE_ERR_INVALID_ARG;
    for (int r = 0; r < g->n; ++r) {
        GroupMember *m = g->mem[r];
        HIP_TRY(hipSetDevice(m->device));
        float a = 0, b = 0;
        HIP_TRY(hipEventSynchronize(m->t2));
        HIP_TRY(hipEventElapsedTime(&a, m->t0, m->t1));
        HIP_TRY(hipEventElapsedTime(&b, m->t1, m->t2));
        if (eval_ms) eval_ms[r] = a;
        if (gather_ms) gather_ms[r] = b;
    }
    return CELESTE_OK;
} ABI_CATCH

extern "C" int celeste_group_shard_sizes(celeste_group_t *g, int32_t *sizes, int64_t *costs) try {
    if (!g) return CELESTE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(g->call_mu);
    if (!g->planned) return CELESTE_ERR_INVALID_ARG;
    for (int r = 0; r < g->n; ++r) {
        if (sizes) sizes[r] = (int32_t)g->mem[r]->idx.size();
        if (costs) { costs[r] = 0; for (int32_t t : g->mem[r]->tg) costs[r] += g->cost[t]; }
    }
    return CELESTE_OK;
} ABI_CATCH

// elbo() for a batch of targets over all members (the drop-in call, host pointers).  As celeste_elbo_eval_batch does on one
// device, every member cuts its shard into parts that are evaluated back to back on its stream while the Hessians of the
// finished parts come down on its copy stream -- straight into the caller's array when that is page-locked and the shard is a
// run of consecutive positions (a group of one), else through page-locked staging and a host scatter that overlaps the later
// parts; the catalog gather (v, d, counters, status of all targets) follows the last part on the gather stream.
extern "C" int celeste_group_elbo_eval_batch(celeste_group_t *g, const double *vp, int32_t n_targets, const int32_t *targets,
                                             uint32_t flags, double *v, double *d, double *h, int64_t *counters, int32_t *status) try {
    if (!g || !vp || n_targets < 0 || (flags & CELESTE_FLAG_SPLIT)) return CELESTE_ERR_INVALID_ARG;
    if (n_targets == 0) return CELESTE_OK;
    if (group_check_targets(g, n_targets, targets) != CELESTE_OK) return CELESTE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(g->call_mu);
    if (g->broken.load()) return CELESTE_ERR_ABORTED;
    const bool trace = getenv("CELESTE_GROUP_TRACE") != nullptr;    // host-side phase times of member 0 to stderr
    const auto tr0 = std::chrono::steady_clock::now();
    auto tr_us = [&] { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tr0).count() * 1e-3; };
    double tr[8] = {0};
    int rc = group_plan_locked(g, vp, n_targets, targets, flags, /*sync=*/false);
    if (rc != CELESTE_OK) return rc;
    tr[0] = tr_us();
    const int W = g->plan_width;
    const size_t blk = g->plan_blk;
    const bool want_h = h && (flags & CELESTE_FLAG_HESS);
    const size_t HS = (flags & CELESTE_FLAG_PACKED_HESS) ? CELESTE_HP : (size_t)CEL_P * CEL_P;
    const bool pin_h = want_h && is_pinned(h, (size_t)n_targets * HS * sizeof(double));
    int worst = CELESTE_OK;
    g->sweep_k = 0;            // (the Hessians of this call went straight to the caller: nothing is left for celeste_group_sweep_results)
    rc = group_run(g, [&](GroupMember *m) -> int {
        HIP_TRY(hipSetDevice(m->device));
        celeste_ctx *c = m->ctx;
        hipStream_t st = c->stream, cs = c->copy_stream;
        double *b = m->d_block[0];
        const size_t nr = m->tg.size();
        // Hessians: the lift kernel writes them into page-locked host memory itself (its device address).  A page-locked array of
        // the caller's takes every member's shard directly, each Hessian at its target's position in the caller's order (the
        // lift's position map: d_idx) -- no staging, no host scatter, whatever the number of members.  A pageable array goes
        // through this member's page-locked staging block, scattered on the host part by part while the next part computes.
        // No copy engine, no second stream.
        bool direct = pin_h;
        double *h_dev = nullptr;
        if (want_h && nr > 0 && direct && (hipHostGetDevicePointer((void **)&h_dev, h, 0) != hipSuccess || !h_dev)) { (void)hipGetLastError(); h_dev = nullptr; direct = false; }
        if (want_h && nr > 0 && !direct) {
            int s1 = group_grow_pinned(&m->p_h, &m->p_h_cap, nr * HS); if (s1 != CELESTE_OK) return s1;
            if (hipHostGetDevicePointer((void **)&h_dev, m->p_h, 0) != hipSuccess || !h_dev) { (void)hipGetLastError(); h_dev = nullptr; }
        }
        if (m->index == 0) { int s1 = group_grow_pinned(&m->p_gathered, &m->p_gathered_cap, blk * g->n); if (s1 != CELESTE_OK) return s1; }
        // parts (staged Hessians only): at least 192 targets each so that a part still fills the chip
        int n_parts = 1;
        if (want_h && !direct && h_dev && !g->timing) n_parts = (int)std::min<size_t>(celeste_ctx::MAX_PARTS, std::max<size_t>(1, nr / 192));
        size_t part_lo[celeste_ctx::MAX_PARTS + 1];
        for (int k = 0; k <= n_parts; ++k) part_lo[k] = nr * (size_t)k / (size_t)n_parts;
        int own_rc = ensure_part_events(c, n_parts), copies = 0;     // (a failure here is a failed launch: the collectives still follow)
        if (g->timing) HIP_TRY(hipEventRecord(m->t0, st));
        if (group_fault(g, m, GROUP_FAULT_LAUNCH)) own_rc = CELESTE_ERR_HIP;
        for (int k = 0; k < n_parts && own_rc == CELESTE_OK && nr > 0; ++k) {
            const size_t lo = part_lo[k], cnt = part_lo[k + 1] - lo;
            int64_t n_chunks = 0;
            for (size_t i = lo; i < lo + cnt; ++i) n_chunks += c->h_src_chunks[m->tg[i]];
            // (direct: the base of the caller's array + the position map; staged / device: slot i of this member's block)
            double *hk = direct ? h_dev : (h_dev ? h_dev + lo * HS : (m->d_h ? m->d_h + lo * HS : nullptr));
            own_rc = launch_eval(c, m->d_vp, (int32_t)cnt, m->d_targets + lo, flags, b + lo, b + W + lo * CEL_P, hk,
                                 reinterpret_cast<int64_t *>(b + (size_t)W * (1 + CEL_P)) + 2 * lo,
                                 reinterpret_cast<int32_t *>(b + (size_t)W * (1 + CEL_P + 2)) + lo, st, true, nullptr, n_chunks, k > 0,
                                 nullptr, n_parts > 1, false, direct ? m->d_idx + lo : nullptr);
            if (own_rc != CELESTE_OK || !want_h) continue;
            if (!h_dev) {    // (the block could not be mapped: copy on the copy stream, as the one-device entry does for pageable arrays)
                if (hipEventRecord(c->part_done[k], st) != hipSuccess || hipStreamWaitEvent(cs, c->part_done[k], 0) != hipSuccess ||
                    group_hessians_down(m, h, false, HS, lo, lo + cnt, cs) != CELESTE_OK ||
                    hipEventRecord(c->part_copied[k], cs) != hipSuccess) { (void)hipGetLastError(); own_rc = CELESTE_ERR_HIP; break; }
            } else if (hipEventRecord(c->part_copied[k], st) != hipSuccess) { (void)hipGetLastError(); own_rc = CELESTE_ERR_HIP; break; }
            copies = k + 1;
        }
        // (a member whose launch failed still takes part in the gather: the collective needs every rank)
        int rc2 = CELESTE_OK;
        if (m->index == 0) tr[1] = tr_us();
        if (g->timing && hipEventRecord(m->t1, st) != hipSuccess) rc2 = CELESTE_ERR_HIP;
        const bool skip = group_fault(g, m, GROUP_FAULT_SWEEP);
        if (!skip && rc2 == CELESTE_OK) {
            if (hipEventRecord(m->done[0], st) != hipSuccess || hipStreamWaitEvent(m->comm_stream, m->done[0], 0) != hipSuccess) { (void)hipGetLastError(); rc2 = CELESTE_ERR_HIP; }
            if (rc2 == CELESTE_OK) rc2 = group_exchange(g, m, b, blk, m->comm_stream);
            if (rc2 == CELESTE_OK && g->timing && hipEventRecord(m->t2, m->comm_stream) != hipSuccess) rc2 = CELESTE_ERR_HIP;
            if (rc2 == CELESTE_OK && m->index == 0 && (g->exchange == GROUP_EXCHANGE_RCCL || g->n == 1) &&
                hipMemcpyAsync(m->p_gathered, m->d_gathered, blk * g->n * sizeof(double), hipMemcpyDeviceToHost, m->comm_stream) != hipSuccess) {
                (void)hipGetLastError(); rc2 = CELESTE_ERR_HIP;
            }
        } else if (skip) rc2 = CELESTE_ERR_HIP;
        if (m->index == 0) tr[2] = tr_us();
        // staged Hessians: host scatter of part k while the later parts are still in flight
        for (int k = 0; k < copies; ++k) {
            if (hipEventSynchronize(c->part_copied[k]) != hipSuccess) { (void)hipGetLastError(); if (own_rc == CELESTE_OK) own_rc = CELESTE_ERR_HIP; break; }
            if (!direct) group_hessians_scatter(m, h, HS, part_lo[k], part_lo[k + 1]);
        }
        if (m->index == 0) tr[3] = tr_us();
        // nothing of this call stays in flight into the caller's memory, whatever happened above
        if (hipStreamSynchronize(cs) != hipSuccess) { (void)hipGetLastError(); if (own_rc == CELESTE_OK) own_rc = CELESTE_ERR_HIP; }
        if (m->index == 0) tr[4] = tr_us();
        if (rc2 == CELESTE_OK) rc2 = group_stream_wait(g, m, m->comm_stream);
        if (hipStreamSynchronize(st) != hipSuccess) { (void)hipGetLastError(); if (own_rc == CELESTE_OK) own_rc = CELESTE_ERR_HIP; }
        if (m->index == 0) tr[5] = tr_us();
        if (rc2 != CELESTE_OK) return rc2;
        if (own_rc != CELESTE_OK) return own_rc;
        if (m->index == 0 && (g->exchange == GROUP_EXCHANGE_RCCL || g->n == 1)) worst = group_catalog_to_host(g, m->p_gathered, flags, v, d, counters, status);
        if (m->index == 0) tr[6] = tr_us();
        return CELESTE_OK;
    }, 1);
    if (trace) fprintf(stderr, "celeste_group_elbo_eval_batch: plan %.0f us | parts launched %.0f | gather enqueued %.0f | part copies waited %.0f | copy stream %.0f | "
                               "gather + eval streams %.0f | catalog on host %.0f | joined %.0f\n", tr[0], tr[1], tr[2], tr[3], tr[4], tr[5], tr[6], tr_us());
    if (rc != CELESTE_OK) { (void)group_quiesce(g); return g->broken.load() ? CELESTE_ERR_ABORTED : rc; }
    // PEER mode: member 0's gathered block is written by the OTHER members' streams; every member has waited for its own gather
    // stream by now, so the block is complete after the join -- it comes down here
    if (g->exchange == GROUP_EXCHANGE_PEER && g->n > 1) {
        GroupMember *m0 = g->mem[0];
        HIP_TRY(hipSetDevice(m0->device));
        HIP_TRY(hipMemcpyAsync(m0->p_gathered, m0->d_gathered, blk * g->n * sizeof(double), hipMemcpyDeviceToHost, m0->ctx->stream));
        HIP_TRY(hipStreamSynchronize(m0->ctx->stream));
        worst = group_catalog_to_host(g, m0->p_gathered, flags, v, d, counters, status);
    }
    return worst;
} ABI_CATCH

// ---- maximize! over the members (one_node_single_infer, ParallelRun.jl:546-607) ------------------------------------
// rows of the member's own targets -> its exchange block; the rows behind them are marked unused (target -1): a member whose
// launch failed packs nothing, and the others' scatter must not take its zeroed block for rows of source 0
__global__ void group_mark_unused_kernel(double *__restrict__ block, int n_used, int width, int row) {
    const int i = n_used + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < width) block[(size_t)i * row] = -1.0;
}

// after the members' optimisations: pack the updated rows, exchange them, bring every member's table up to date
static int group_exchange_rows(celeste_group *g, GroupMember *m, double *d_table, int n_own, const int32_t *d_own_targets,
                               const int32_t *d_it, const int32_t *d_ev, const double *d_el, const int32_t *d_st, int width, int row,
                               const GroupCounts &cnt) {
    hipStream_t st = m->ctx->stream;
    const size_t blk = (size_t)width * row;
    if (blk > m->block_cap) {
        HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipStreamSynchronize(m->comm_stream));
        for (int k = 0; k < 2; ++k) { if (m->d_block[k]) (void)hipFree(m->d_block[k]); m->d_block[k] = nullptr; }
        m->block_cap = 0;
        for (int k = 0; k < 2; ++k) HIP_TRY(hipMalloc((void **)&m->d_block[k], blk * sizeof(double)));
        m->block_cap = blk;
    }
    int s1 = group_grow(&m->d_gathered, &m->gathered_cap, blk * g->n, st);
    if (s1 != CELESTE_OK) return s1;
    if (group_fault(g, m, GROUP_FAULT_BARRIER)) return CELESTE_ERR_HIP;
    // Every member arrives here before anyone enqueues the exchange.  PEER mode needs it (the others copy into d_gathered: it
    // must exist on every member first); with RCCL it is what keeps a member that failed above -- an allocation, its launch's
    // set-up -- from leaving the others inside a collective that will never complete: the failing member's return wakes the
    // barrier (group_fail), and every member leaves the call with an error instead.
    if (!group_barrier(g)) return g->broken.load() ? CELESTE_ERR_ABORTED : CELESTE_ERR_HIP;
    // (a failure from here to the collective leaves the others inside theirs: the abort protocol's case)
    if (group_fault(g, m, GROUP_FAULT_ROWS)) return CELESTE_ERR_HIP;
    HIP_TRY(hipMemsetAsync(m->d_block[0], 0, blk * sizeof(double), st));
    if (n_own > 0)
        hipLaunchKernelGGL(group_pack_kernel, dim3((unsigned)n_own), dim3(64), 0, st, d_table, d_own_targets, n_own, d_it, d_ev, d_el, d_st,
                           m->d_block[0], row);
    if (n_own < width)
        hipLaunchKernelGGL(group_mark_unused_kernel, dim3((unsigned)((width - n_own + 255) / 256)), dim3(256), 0, st, m->d_block[0], n_own, width, row);
    s1 = group_exchange(g, m, m->d_block[0], blk, st);
    if (s1 != CELESTE_OK) return s1;
    if (g->exchange == GROUP_EXCHANGE_PEER) {    // every member's copies have landed before anyone reads its gathered buffer
        HIP_TRY(hipStreamSynchronize(st));
        if (!group_barrier(g)) return CELESTE_ERR_HIP;
    }
    if (g->n > 1)
        hipLaunchKernelGGL(group_scatter_kernel, dim3((unsigned)width, (unsigned)g->n), dim3(64), 0, st, d_table, m->d_gathered, cnt,
                           g->n, m->index, width, row);
    HIP_TRY(hipGetLastError());
    return CELESTE_OK;
}

extern "C" int celeste_group_maximize_batch(celeste_group_t *g, double *vp, const double *vp_neighbors, const double *pos_centers,
                                            int32_t n_targets, const int32_t *targets, const celeste_optim_config_t *cfg,
                                            int32_t *iterations, int32_t *f_evals, double *elbo, int32_t *status) try {
    if (!g || !vp || n_targets < 0) return CELESTE_ERR_INVALID_ARG;
    if (n_targets == 0) return CELESTE_OK;
    if (group_check_targets(g, n_targets, targets) != CELESTE_OK) return CELESTE_ERR_INVALID_ARG;
    {
        std::vector<uint8_t> seen;
        if (check_distinct_targets(g->mem[0]->ctx, n_targets, targets, seen) != CELESTE_OK) return CELESTE_ERR_INVALID_ARG;
        OptParams op; uint32_t fl;
        if (optim_config(cfg, &op, &fl) != CELESTE_OK) return CELESTE_ERR_INVALID_ARG;
    }
    std::lock_guard<std::mutex> lock(g->call_mu);
    if (g->broken.load()) return CELESTE_ERR_ABORTED;
    g->planned = false;
    (void)group_quiesce(g);
    if (g->broken.load()) return CELESTE_ERR_ABORTED;
    group_barrier_reset(g);
    group_shard(g, n_targets, targets);
    int W = 1;
    GroupCounts cnt; memset(&cnt, 0, sizeof cnt);
    for (GroupMember *m : g->mem) { W = std::max(W, (int)m->idx.size()); cnt.n[m->index] = (int)m->idx.size(); }
    const size_t tb = (size_t)g->S * CEL_P * sizeof(double);
    memcpy(g->p_vp, vp, tb);
    if (vp_neighbors) memcpy(g->p_vp_nbr, vp_neighbors, tb);
    const size_t blk = (size_t)W * GROUP_ROW;
    int rc = group_run(g, [&](GroupMember *m) -> int {
        HIP_TRY(hipSetDevice(m->device));
        hipStream_t st = m->ctx->stream;
        const size_t nr = m->tg.size();
        int s1 = group_target_buffers(m, nr);
        if (s1 != CELESTE_OK) return s1;
        HIP_TRY(hipMemcpyAsync(m->d_vp, g->p_vp, tb, hipMemcpyHostToDevice, st));
        if (vp_neighbors) HIP_TRY(hipMemcpyAsync(m->d_vp_nbr, g->p_vp_nbr, tb, hipMemcpyHostToDevice, st));
        int own_rc = CELESTE_OK;
        if (nr > 0) {
            HIP_TRY(hipMemcpyAsync(m->d_targets, m->tg.data(), nr * sizeof(int32_t), hipMemcpyHostToDevice, st));
            if (pos_centers) {
                std::vector<double> pc(nr * 2);
                for (size_t i = 0; i < nr; ++i) { pc[2 * i] = pos_centers[2 * (size_t)m->idx[i]]; pc[2 * i + 1] = pos_centers[2 * (size_t)m->idx[i] + 1]; }
                HIP_TRY(hipMemcpyAsync(m->d_pos, pc.data(), nr * 2 * sizeof(double), hipMemcpyHostToDevice, st));
                HIP_TRY(hipStreamSynchronize(st));   // (pc is a local)
            }
            if (group_fault(g, m, GROUP_FAULT_LAUNCH)) own_rc = CELESTE_ERR_HIP;
            else own_rc = celeste_maximize_batch_device(m->ctx, m->d_vp, vp_neighbors ? m->d_vp_nbr : nullptr, pos_centers ? m->d_pos : nullptr,
                                                        (int32_t)nr, m->d_targets, cfg, m->d_it, m->d_ev, m->d_el, m->d_st, st);
        }
        // (a member whose launch failed still takes part in the exchange: the collective needs every rank; its rows are
        // marked unused)
        s1 = group_exchange_rows(g, m, m->d_vp, own_rc == CELESTE_OK ? (int)nr : 0, m->d_targets, m->d_it, m->d_ev, m->d_el, m->d_st, W,
                                 GROUP_ROW, cnt);
        if (s1 != CELESTE_OK) return s1;
        if (m->index == 0) {
            s1 = group_grow_pinned(&m->p_gathered, &m->p_gathered_cap, blk * g->n);
            if (s1 != CELESTE_OK) return s1;
            HIP_TRY(hipMemcpyAsync(m->p_gathered, m->d_gathered, blk * g->n * sizeof(double), hipMemcpyDeviceToHost, st));
        }
        s1 = group_stream_wait(g, m, st);
        if (s1 != CELESTE_OK) return s1;
        return own_rc;      // (a fused launch that gave up shows as CELESTE_ERR_HIP in its targets' status: optim_finalize_kernel)
    }, 1);
    if (rc != CELESTE_OK) { (void)group_quiesce(g); return g->broken.load() ? CELESTE_ERR_ABORTED : rc; }
    int worst = CELESTE_OK;
    const double *gp = g->mem[0]->p_gathered;
    for (int r = 0; r < g->n; ++r) {
        const GroupMember *o = g->mem[r];
        for (size_t i = 0; i < o->idx.size(); ++i) {
            const double *row = gp + ((size_t)r * W + i) * GROUP_ROW;
            const size_t q = (size_t)o->idx[i];
            const int t = (int)row[0], st1 = (int)row[4];
            if (t != o->tg[i]) return CELESTE_ERR_HIP;            // the exchange lost a row
            memcpy(vp + (size_t)t * CEL_P, row + 5, CEL_P * sizeof(double));   // (a failed target carries its input row)
            if (iterations) iterations[q] = (int32_t)row[1];
            if (f_evals) f_evals[q] = (int32_t)row[2];
            if (elbo) elbo[q] = row[3];
            if (status) status[q] = st1;
            if (st1 != CELESTE_OK && worst == CELESTE_OK) worst = st1;
        }
    }
    return worst;
} ABI_CATCH

// ---- joint inference over the members (one_node_joint_infer, ParallelRun.jl:135-196, 302-397) -----------------------
// The connected components of a Cyclades batch never conflict (partition.jl:173-236): they are sharded over the members by
// cost, and every member runs its components' sources one after another against ITS table (celeste_joint_infer's schedule:
// layer j = the j-th sources of its components).  A member's table must be brought up to date only when the member is about
// to READ a row -- a target's own, a neighbour's -- that ANOTHER member has written since the last exchange; the host knows
// both (the shards and the neighbour graph), so the steps (sweep, batch) are cut into SEGMENTS: maximal runs in front of
// which no such read exists.  A segment is ONE launch chain per member (celeste_joint_infer's dataflow launch over the
// member's layers of all its batches) followed by ONE exchange of the rows the segment wrote.  A group of one has nobody to
// wait for: its whole schedule is one segment -- celeste_joint_infer's single launch -- and one (one-rank) exchange.  In a crowded
// field on several members every batch reads what the batch before wrote on another member: one exchange per batch.
extern "C" int celeste_group_joint_infer(celeste_group_t *g, double *vp, int32_t n_sweeps, int32_t n_batches, const int64_t *batch_offsets,
                                         const int64_t *comp_offsets, const int32_t *comp_targets, const double *pos_centers,
                                         const celeste_optim_config_t *cfg, int32_t *iterations, int32_t *f_evals, double *elbo,
                                         int32_t *status, int64_t *n_exchanges) try {
    if (!g || !vp || n_sweeps < 0 || n_batches < 0 || (n_batches > 0 && (!batch_offsets || !comp_offsets || !comp_targets)))
        return CELESTE_ERR_INVALID_ARG;
    if (n_exchanges) *n_exchanges = 0;
    if (n_batches == 0 || n_sweeps == 0) return CELESTE_OK;
    if (batch_offsets[0] != 0 || comp_offsets[0] != 0) return CELESTE_ERR_INVALID_ARG;
    // the schedule's shape first: nothing below indexes comp_offsets / comp_targets before these hold
    for (int b = 0; b < n_batches; ++b) if (batch_offsets[b + 1] < batch_offsets[b]) return CELESTE_ERR_INVALID_ARG;
    const int64_t n_comps = batch_offsets[n_batches];
    if (n_comps < 0) return CELESTE_ERR_INVALID_ARG;
    for (int64_t k = 0; k < n_comps; ++k) if (comp_offsets[k + 1] < comp_offsets[k]) return CELESTE_ERR_INVALID_ARG;
    const int64_t E = comp_offsets[n_comps];
    if (E > 0x7fffffff || group_check_targets(g, (int32_t)E, comp_targets) != CELESTE_OK) return CELESTE_ERR_INVALID_ARG;
    const celeste_ctx *c0 = g->mem[0]->ctx;
    // a batch: every source at most once, no neighbours across components
    {
        std::vector<int32_t> comp_of((size_t)g->S);
        for (int b = 0; b < n_batches; ++b) {
            std::fill(comp_of.begin(), comp_of.end(), -1);
            for (int64_t k = batch_offsets[b]; k < batch_offsets[b + 1]; ++k)
                for (int64_t e = comp_offsets[k]; e < comp_offsets[k + 1]; ++e) {
                    if (comp_of[comp_targets[e]] >= 0) return CELESTE_ERR_INVALID_ARG;
                    comp_of[comp_targets[e]] = (int32_t)(k - batch_offsets[b]);
                }
            for (int64_t k = batch_offsets[b]; k < batch_offsets[b + 1]; ++k)
                for (int64_t e = comp_offsets[k]; e < comp_offsets[k + 1]; ++e) {
                    const int t = comp_targets[e];
                    for (int64_t q = c0->h_nbr_off[t]; q < c0->h_nbr_off[t + 1]; ++q) {
                        const int32_t oc = comp_of[c0->h_nbr_idx[q]];
                        if (oc >= 0 && oc != (int32_t)(k - batch_offsets[b])) return CELESTE_ERR_INVALID_ARG;
                    }
                }
        }
    }
    std::lock_guard<std::mutex> lock(g->call_mu);
    if (g->broken.load()) return CELESTE_ERR_ABORTED;
    g->planned = false;
    (void)group_quiesce(g);
    if (g->broken.load()) return CELESTE_ERR_ABORTED;
    group_barrier_reset(g);
    g->abort_rc.store(0);
    const size_t tb = (size_t)g->S * CEL_P * sizeof(double);
    memcpy(g->p_vp, vp, tb);
    // per batch and member: its components (cost = the chunks of their sources), flattened into layers
    struct Part { std::vector<int64_t> off; std::vector<int32_t> tg; std::vector<int64_t> entry; std::vector<double> pos; };
    std::vector<std::vector<Part>> parts((size_t)n_batches, std::vector<Part>((size_t)g->n));
    for (int b = 0; b < n_batches; ++b) {
        const int64_t k0 = batch_offsets[b], nk = batch_offsets[b + 1] - k0;
        std::vector<int64_t> w((size_t)nk, 0);
        for (int64_t k = 0; k < nk; ++k)
            for (int64_t e = comp_offsets[k0 + k]; e < comp_offsets[k0 + k + 1]; ++e) w[k] += c0->h_src_chunks[comp_targets[e]];
        std::vector<std::vector<int32_t>> shards;
        group_lpt(g->n, w, shards);
        for (int r = 0; r < g->n; ++r) {
            Part &p = parts[b][r];
            size_t depth = 0;
            for (int32_t k : shards[r]) depth = std::max(depth, (size_t)(comp_offsets[k0 + k + 1] - comp_offsets[k0 + k]));
            p.off.push_back(0);
            for (size_t j = 0; j < depth; ++j) {
                for (int32_t k : shards[r]) {
                    const int64_t lo = comp_offsets[k0 + k], len = comp_offsets[k0 + k + 1] - lo;
                    if ((int64_t)j >= len) continue;
                    p.tg.push_back(comp_targets[lo + j]);
                    p.entry.push_back(lo + (int64_t)j);
                    if (pos_centers) { p.pos.push_back(pos_centers[2 * (lo + j)]); p.pos.push_back(pos_centers[2 * (lo + j) + 1]); }
                }
                p.off.push_back((int64_t)p.tg.size());
            }
        }
    }
    // segments of steps (step u = sweep u / n_batches, batch u % n_batches): a new segment starts where a member reads a row
    // another member has written since the last exchange
    const int64_t U = (int64_t)n_sweeps * n_batches;
    std::vector<int64_t> seg_lo;
    {
        std::vector<int32_t> writer((size_t)g->S, -1);
        seg_lo.push_back(0);
        for (int64_t u = 0; u < U; ++u) {
            const int b = (int)(u % n_batches);
            bool stale = false;
            for (int r = 0; r < g->n && !stale; ++r)
                for (int32_t t : parts[b][r].tg) {
                    if (writer[t] >= 0 && writer[t] != r) { stale = true; break; }
                    for (int64_t q = c0->h_nbr_off[t]; q < c0->h_nbr_off[t + 1] && !stale; ++q) {
                        const int32_t w = writer[c0->h_nbr_idx[q]];
                        if (w >= 0 && w != r) stale = true;
                    }
                    if (stale) break;
                }
            if (stale && u > seg_lo.back()) { seg_lo.push_back(u); std::fill(writer.begin(), writer.end(), -1); }
            for (int r = 0; r < g->n; ++r) for (int32_t t : parts[b][r].tg) writer[t] = r;
        }
        seg_lo.push_back(U);
    }
    const size_t n_seg = seg_lo.size() - 1;
    // per segment and member: the concatenated layers, and the rows the segment writes (every source once)
    struct Seg { std::vector<int64_t> off; std::vector<int32_t> tg, rows; std::vector<int64_t> out; std::vector<double> pos; };
    std::vector<std::vector<Seg>> segs(n_seg, std::vector<Seg>((size_t)g->n));
    std::vector<int> widths(n_seg, 1);
    {
        std::vector<uint8_t> in_rows((size_t)g->S);
        for (size_t sg = 0; sg < n_seg; ++sg)
            for (int r = 0; r < g->n; ++r) {
                Seg &q = segs[sg][(size_t)r];
                q.off.push_back(0);
                std::fill(in_rows.begin(), in_rows.end(), 0);
                for (int64_t u = seg_lo[sg]; u < seg_lo[sg + 1]; ++u) {
                    const Part &p = parts[(size_t)(u % n_batches)][(size_t)r];
                    const int64_t base = (int64_t)q.tg.size(), sw = u / n_batches;
                    q.tg.insert(q.tg.end(), p.tg.begin(), p.tg.end());
                    q.pos.insert(q.pos.end(), p.pos.begin(), p.pos.end());
                    for (int64_t e : p.entry) q.out.push_back(sw * E + e);
                    for (size_t l = 1; l < p.off.size(); ++l) q.off.push_back(base + p.off[l]);
                    for (int32_t t : p.tg) if (!in_rows[t]) { in_rows[t] = 1; q.rows.push_back(t); }
                }
                widths[sg] = std::max(widths[sg], (int)q.rows.size());
            }
    }
    std::atomic<int64_t> exchanges{0};
    int rc = group_run(g, [&](GroupMember *m) -> int {
        HIP_TRY(hipSetDevice(m->device));
        hipStream_t st = m->ctx->stream;
        HIP_TRY(hipMemcpyAsync(m->d_vp, g->p_vp, tb, hipMemcpyHostToDevice, st));
        int worst = CELESTE_OK;
        for (size_t sg = 0; sg < n_seg; ++sg) {
            const Seg &p = segs[sg][(size_t)m->index];
            const size_t ne = p.tg.size();
            int own_rc = CELESTE_OK;
            if (ne > 0) {
                std::vector<int32_t> it(ne), ev(ne), stt(ne);
                std::vector<double> el(ne);
                // (the entry state of the table, should the dataflow launch hand the schedule to the layered driver)
                HIP_TRY(hipMemcpyAsync(m->d_entry, m->d_vp, tb, hipMemcpyDeviceToDevice, st));
                if (group_fault(g, m, GROUP_FAULT_LAUNCH)) own_rc = CELESTE_ERR_HIP;
                else own_rc = joint_run(m->ctx, m->d_vp, m->d_entry, (int32_t)(p.off.size() - 1), p.off.data(), p.tg.data(),
                                        pos_centers ? p.pos.data() : nullptr, cfg, it.data(), ev.data(), el.data(), stt.data(), false);
                if (own_rc == CELESTE_OK || own_rc == CELESTE_ERR_NONFINITE_INPUT || own_rc == CELESTE_ERR_NONFINITE_RESULT)
                    for (size_t i = 0; i < ne; ++i) {
                        const size_t q = (size_t)p.out[i];
                        if (iterations) iterations[q] = it[i];
                        if (f_evals) f_evals[q] = ev[i];
                        if (elbo) elbo[q] = el[i];
                        if (status) status[q] = stt[i];
                    }
                if (own_rc == CELESTE_ERR_NONFINITE_INPUT || own_rc == CELESTE_ERR_NONFINITE_RESULT) { if (worst == CELESTE_OK) worst = own_rc; own_rc = CELESTE_OK; }
                // a launch that failed: the others would wait for this member at the next exchange -- tell them before this one
                if (own_rc != CELESTE_OK) { int z = 0; g->abort_rc.compare_exchange_strong(z, own_rc); }
            }
            int s1 = group_target_buffers(m, p.rows.size());
            if (s1 != CELESTE_OK) return s1;
            if (!p.rows.empty()) HIP_TRY(hipMemcpyAsync(m->d_targets, p.rows.data(), p.rows.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
            GroupCounts cnt; memset(&cnt, 0, sizeof cnt);
            for (int r = 0; r < g->n; ++r) cnt.n[r] = (int)segs[sg][(size_t)r].rows.size();
            s1 = group_exchange_rows(g, m, m->d_vp, own_rc == CELESTE_OK ? (int)p.rows.size() : 0, m->d_targets, nullptr, nullptr, nullptr, nullptr,
                                     widths[sg], GROUP_JROW, cnt);
            if (s1 != CELESTE_OK) return s1;
            s1 = group_stream_wait(g, m, st);
            if (s1 != CELESTE_OK) return s1;
            if (m->index == 0) exchanges.fetch_add(1);
            // (the exchange needed every member's enqueue, and a failing member raised the flag before its own)
            if (g->exchange == GROUP_EXCHANGE_PEER && !group_barrier(g)) return CELESTE_ERR_HIP;
            if (const int a = g->abort_rc.load()) return a;
        }
        if (m->index == 0) {
            HIP_TRY(hipMemcpyAsync(g->p_vp_nbr, m->d_vp, tb, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        return worst;
    }, n_seg);
    if (n_exchanges) *n_exchanges = exchanges.load();
    if (rc == CELESTE_OK || rc == CELESTE_ERR_NONFINITE_INPUT || rc == CELESTE_ERR_NONFINITE_RESULT) memcpy(vp, g->p_vp_nbr, tb);
    else { (void)group_quiesce(g); if (g->broken.load()) rc = CELESTE_ERR_ABORTED; }
    return rc;
} ABI_CATCH

/*
 * fake_rccl.c -- TEST-ONLY stand-in for librccl, preloaded (LD_PRELOAD) under the celeste_group_* tests so that the RCCL branch
 * of csrc/group.h runs with MORE THAN ONE RANK on a one-GPU box (real RCCL refuses a communicator that names a device twice).
 * Never linked into, loaded by or shipped with the product; tests/test_gpu_group_rccl_branch.py builds and preloads it.
 *
 * It is deliberately STRICTER than RCCL.  Every ncclAllGather call blocks ON THE HOST until all ranks of the communicator
 * have made their k-th call, checks that they agree on the element count, and only then lets every rank's stream copy every
 * rank's block (event-ordered after the senders' streams).  So the mistakes that would hang or corrupt a real node --
 *   - a member that skips a collective (early return on an error path),
 *   - members that issue a different NUMBER of collectives in a call,
 *   - collectives issued with different counts on different members,
 * -- deadlock or fail HERE, inside a pytest timeout, instead of on the first 8-GPU lease.  ncclCommAbort wakes every rank that
 * waits in a rendezvous (they return ncclInternalError), which is the behaviour group.h's abort protocol relies on.
 *
 * Exports: the RCCL symbols libceleste_mi355x.so imports (ncclCommInitAll, ncclAllGather, ncclCommCount, ncclCommDestroy,
 * ncclCommAbort, ncclCommGetAsyncError, ncclGetErrorString) + fake_rccl_stats for the tests.
 * Build: gcc -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include fake_rccl.c -o libfake_rccl.so -L/opt/rocm/lib -lamdhip64 -lpthread
 */
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5,
               ncclFloat16 = 6, ncclHalf = 6, ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;

#define FAKE_MAX_RANKS 16

typedef struct fake_world {
    int n, refs, aborted, mismatch;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int arrived;
    uint64_t gen;
    size_t count[FAKE_MAX_RANKS];
    const void *send[FAKE_MAX_RANKS];
    uint64_t call_no[FAKE_MAX_RANKS];
    hipEvent_t ready[FAKE_MAX_RANKS], copied[FAKE_MAX_RANKS];
} fake_world;

typedef struct ncclComm {
    fake_world *w;
    int rank, dev;
    uint64_t k;          /* collectives this rank has entered */
} *ncclComm_t;

/* process-wide counters: [0] all-gathers completed by all ranks (one per collective, not per rank), [1] of those with > 1 rank,
 * [2] communicators aborted, [3] rendezvous that an abort woke, [4] count mismatches detected */
static uint64_t g_stats[8];
static pthread_mutex_t g_stats_mu = PTHREAD_MUTEX_INITIALIZER;
static void stat_add(int k) { pthread_mutex_lock(&g_stats_mu); ++g_stats[k]; pthread_mutex_unlock(&g_stats_mu); }
void fake_rccl_stats(uint64_t out[8]) { pthread_mutex_lock(&g_stats_mu); memcpy(out, g_stats, sizeof g_stats); pthread_mutex_unlock(&g_stats_mu); }

static size_t dtype_size(ncclDataType_t t) {
    switch ((int)t) { case 0: case 1: return 1; case 6: return 2; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; default: return 0; }
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled cuda error (fake_rccl)";
    case ncclSystemError: return "unhandled system error (fake_rccl)";
    case ncclInternalError: return "internal error / communicator aborted (fake_rccl)";
    case ncclInvalidArgument: return "invalid argument (fake_rccl)";
    case ncclInvalidUsage: return "invalid usage (fake_rccl)";
    default: return "unknown result code (fake_rccl)";
    }
}
const char *ncclGetLastError(ncclComm_t comm) { (void)comm; return ""; }
ncclResult_t ncclGetVersion(int *v) { if (v) *v = 22606; return ncclSuccess; }

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist) {
    if (!comms || ndev < 1 || ndev > FAKE_MAX_RANKS) return ncclInvalidArgument;
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess) return ncclUnhandledCudaError;
    fake_world *w = (fake_world *)calloc(1, sizeof *w);
    if (!w) return ncclSystemError;
    w->n = ndev; w->refs = ndev;
    pthread_mutex_init(&w->mu, NULL);
    pthread_cond_init(&w->cv, NULL);
    for (int r = 0; r < ndev; ++r) {
        const int dev = devlist ? devlist[r] : r;      /* (repeated devices are the point of this library) */
        if (hipSetDevice(dev) != hipSuccess || hipEventCreateWithFlags(&w->ready[r], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&w->copied[r], hipEventDisableTiming) != hipSuccess) { (void)hipSetDevice(prev); return ncclUnhandledCudaError; }
        ncclComm_t c = (ncclComm_t)calloc(1, sizeof *c);
        if (!c) return ncclSystemError;
        c->w = w; c->rank = r; c->dev = dev;
        comms[r] = c;
    }
    (void)hipSetDevice(prev);
    fprintf(stderr, "fake_rccl: communicator of %d rank(s) created (test stand-in, strict host rendezvous)\n", ndev);
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = comm->w->n;
    return ncclSuccess;
}

static void world_unref(fake_world *w, int rank) {
    pthread_mutex_lock(&w->mu);
    const int left = --w->refs;
    pthread_mutex_unlock(&w->mu);
    (void)rank;
    if (left == 0) {
        for (int r = 0; r < w->n; ++r) { if (w->ready[r]) (void)hipEventDestroy(w->ready[r]); if (w->copied[r]) (void)hipEventDestroy(w->copied[r]); }
        pthread_mutex_destroy(&w->mu); pthread_cond_destroy(&w->cv);
        free(w);
    }
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclSuccess;
    world_unref(comm->w, comm->rank);
    free(comm);
    return ncclSuccess;
}

/* Aborting ANY rank's communicator aborts the world: every rank waiting in a rendezvous returns ncclInternalError, every later
 * collective on any rank fails at once.  (Real RCCL needs every rank's communicator aborted; group.h aborts them all.)  The
 * communicator object itself stays valid until the world's last reference goes -- a rank that is inside ncclAllGather on it
 * while another thread aborts it is exactly the situation group.h's abort protocol can produce. */
ncclResult_t ncclCommAbort(ncclComm_t comm) {
    if (!comm) return ncclSuccess;
    fake_world *w = comm->w;
    pthread_mutex_lock(&w->mu);
    w->aborted = 1;
    pthread_cond_broadcast(&w->cv);
    pthread_mutex_unlock(&w->mu);
    stat_add(2);
    world_unref(w, comm->rank);
    /* (comm is leaked on purpose: a rank may still be inside a call on it) */
    return ncclSuccess;
}

ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t *err) {
    if (!comm || !err) return ncclInvalidArgument;
    pthread_mutex_lock(&comm->w->mu);
    *err = comm->w->aborted ? ncclInternalError : ncclSuccess;
    pthread_mutex_unlock(&comm->w->mu);
    return ncclSuccess;
}

/* all ranks meet; returns 0, or -1 when the world was aborted while waiting */
static int rendezvous(fake_world *w) {
    pthread_mutex_lock(&w->mu);
    if (w->aborted) { pthread_mutex_unlock(&w->mu); return -1; }
    const uint64_t gen = w->gen;
    if (++w->arrived == w->n) { w->arrived = 0; ++w->gen; pthread_cond_broadcast(&w->cv); }
    else while (w->gen == gen && !w->aborted) pthread_cond_wait(&w->cv, &w->mu);
    const int ab = w->gen == gen && w->aborted;      /* (a rendezvous that completed counts even if the abort came right after) */
    if (ab) { --w->arrived; stat_add(3); }
    pthread_mutex_unlock(&w->mu);
    return ab ? -1 : 0;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    if (!comm || !sendbuff || !recvbuff) return ncclInvalidArgument;
    fake_world *w = comm->w;
    const size_t es = dtype_size(datatype);
    if (es == 0) return ncclInvalidArgument;
    const int r = comm->rank;
    const uint64_t k = ++comm->k;
    /* this rank's block is ready when its stream gets here */
    if (hipEventRecord(w->ready[r], stream) != hipSuccess) return ncclUnhandledCudaError;
    pthread_mutex_lock(&w->mu);
    w->send[r] = sendbuff; w->count[r] = sendcount * es; w->call_no[r] = k;
    pthread_mutex_unlock(&w->mu);
    if (rendezvous(w) != 0) return ncclInternalError;                 /* 1: every rank has made its k-th call */
    int bad = 0;
    for (int j = 0; j < w->n; ++j) if (w->count[j] != w->count[r] || w->call_no[j] != k) bad = 1;
    if (bad) {
        fprintf(stderr, "fake_rccl: rank %d: ncclAllGather #%llu disagrees with another rank (counts / call numbers:", r, (unsigned long long)k);
        for (int j = 0; j < w->n; ++j) fprintf(stderr, " %zu/#%llu", w->count[j], (unsigned long long)w->call_no[j]);
        fprintf(stderr, ")\n");
        if (r == 0) stat_add(4);
    }
    hipError_t e = hipSuccess;
    const size_t bytes = w->count[r];
    if (!bad)
        for (int j = 0; j < w->n && e == hipSuccess; ++j) {
            if (j != r) e = hipStreamWaitEvent(stream, w->ready[j], 0);
            if (e == hipSuccess && !(j == r && (const char *)recvbuff + (size_t)j * bytes == (const char *)sendbuff))
                e = hipMemcpyAsync((char *)recvbuff + (size_t)j * bytes, w->send[j], bytes, hipMemcpyDeviceToDevice, stream);
        }
    if (e == hipSuccess) e = hipEventRecord(w->copied[r], stream);
    if (rendezvous(w) != 0) return ncclInternalError;                 /* 2: every rank has enqueued its copies */
    /* nobody's stream runs past the collective (and overwrites its send block) before everybody has read it */
    for (int j = 0; j < w->n && e == hipSuccess; ++j) if (j != r) e = hipStreamWaitEvent(stream, w->copied[j], 0);
    if (rendezvous(w) != 0) return ncclInternalError;                 /* 3: the events may be re-recorded by the next call */
    if (r == 0) { stat_add(0); if (w->n > 1) stat_add(1); }
    if (bad) return ncclInvalidArgument;
    return e == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

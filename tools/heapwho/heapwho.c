/* LD_PRELOAD malloc interposer (diagnosis aid, not part of the product): remembers who allocated and who freed every heap
 * block whose request size lies in [HEAPWHO_LO, HEAPWHO_HI] (default 880 .. 936 bytes), so that when a block turns out to have
 * been written through a stale pointer its previous owners can be named.  heapwho_dump(ptr, path) appends the recorded events
 * of that address (allocations and frees, oldest first, six return addresses each, resolved with dladdr) to `path`.
 * HEAPWHO_QUARANTINE=<substring of a library name> (e.g. libamdhip64): blocks of 32 .. 4096 bytes FREED by that library are not
 * given back to the allocator but filled with 0xA5 and parked; heapwho_scan(path) reports every parked block that has changed
 * since, with the backtrace of its free -- a write through a stale pointer by that library, caught whether or not anything else
 * reuses the memory.  The oldest parked blocks are checked and really freed when the park (32768 blocks) is full.
 * build: gcc -O1 -g -shared -fPIC -o libheapwho.so heapwho.c -ldl -lpthread */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/syscall.h>

static void *(*real_malloc)(size_t);
static void (*real_free)(void *);
static void *(*real_calloc)(size_t, size_t);
static void *(*real_realloc)(void *, size_t);
static int (*real_posix_memalign)(void **, size_t, size_t);
static void *(*real_aligned_alloc)(size_t, size_t);

#define NFR 8
typedef struct { void *ptr; uint64_t seq; uint32_t size; int32_t tid; int kind; void *fr[NFR]; } Ev;
#define RING (1u << 18)
static Ev ring[RING];
static volatile uint64_t seq_ctr;
/* live set of tracked blocks: open addressing on the pointer */
#define LIVE (1u << 16)
static void *volatile live[LIVE];
static size_t lo = 880, hi = 936;
static __thread int busy;
static char boot[1 << 16];
static size_t boot_used;
static int inited;
static const char *q_name;     /* HEAPWHO_QUARANTINE */

static void init(void) {
    if (inited) return;
    inited = 1;
    real_malloc = dlsym(RTLD_NEXT, "malloc");
    real_free = dlsym(RTLD_NEXT, "free");
    real_calloc = dlsym(RTLD_NEXT, "calloc");
    real_realloc = dlsym(RTLD_NEXT, "realloc");
    real_posix_memalign = dlsym(RTLD_NEXT, "posix_memalign");
    real_aligned_alloc = dlsym(RTLD_NEXT, "aligned_alloc");
    const char *e = getenv("HEAPWHO_LO"); if (e) lo = (size_t)atol(e);
    e = getenv("HEAPWHO_HI"); if (e) hi = (size_t)atol(e);
    q_name = getenv("HEAPWHO_QUARANTINE");
    if (q_name && !*q_name) q_name = NULL;
    busy = 1; { void *tmp[4]; backtrace(tmp, 4); } busy = 0;     /* (the first backtrace() loads libgcc: allocates) */
}

static void record(void *p, size_t size, int kind) {
    if (busy) return;
    busy = 1;
    const uint64_t s = __sync_fetch_and_add(&seq_ctr, 1);
    Ev *e = &ring[s & (RING - 1)];
    e->ptr = p; e->seq = s; e->size = (uint32_t)size; e->kind = kind; e->tid = (int32_t)syscall(SYS_gettid);
    memset(e->fr, 0, sizeof e->fr);
    void *fr[NFR + 2];
    int n = backtrace(fr, NFR + 2);
    for (int i = 2; i < n; ++i) e->fr[i - 2] = fr[i];
    busy = 0;
}
static int live_find(void *p, int insert) {
    uint32_t h = (uint32_t)(((uintptr_t)p >> 4) * 2654435761u) & (LIVE - 1);
    for (int k = 0; k < 64; ++k, h = (h + 1) & (LIVE - 1)) {
        if (live[h] == p) return (int)h;
        if (insert && live[h] == NULL && __sync_bool_compare_and_swap(&live[h], NULL, p)) return (int)h;
    }
    return -1;
}
static void on_alloc(void *p, size_t size) {
    if (!p || size < lo || size > hi) return;
    live_find(p, 1);
    record(p, size, 1);
}
static void on_free(void *p) {
    if (!p) return;
    int h = live_find(p, 0);
    if (h < 0) return;
    live[h] = NULL;
    record(p, 0, 2);
}

/* ---- quarantine of the blocks one library frees ---- */
#include <link.h>
#include <malloc.h>
typedef struct { void *ptr; uint32_t size; int32_t tid; void *fr[NFR]; } Park;
#define PARK (1u << 15)
static Park park[PARK];
static volatile uint64_t park_ctr;
static uintptr_t q_lo, q_hi, cxx_lo, cxx_hi;
static int q_scans;
static pthread_mutex_t q_mu = PTHREAD_MUTEX_INITIALIZER;
static volatile uint64_t q_hits;

static int phdr_cb(struct dl_phdr_info *info, size_t sz, void *data) {
    (void)sz; (void)data;
    if (!info->dlpi_name) return 0;
    const int is_q = q_name && strstr(info->dlpi_name, q_name) != NULL;
    const int is_cxx = strstr(info->dlpi_name, "libstdc++") != NULL;
    if (!is_q && !is_cxx) return 0;
    uintptr_t lo = ~(uintptr_t)0, hi = 0;
    for (int i = 0; i < info->dlpi_phnum; ++i) {
        if (info->dlpi_phdr[i].p_type != PT_LOAD || !(info->dlpi_phdr[i].p_flags & PF_X)) continue;
        uintptr_t a = info->dlpi_addr + info->dlpi_phdr[i].p_vaddr, b = a + info->dlpi_phdr[i].p_memsz;
        if (a < lo) lo = a;
        if (b > hi) hi = b;
    }
    if (hi > lo) { if (is_q) { q_lo = lo; q_hi = hi; } else { cxx_lo = lo; cxx_hi = hi; } }
    return 0;
}
static void park_report(FILE *f, const Park *k, const unsigned char *b) {
    fprintf(f, "heapwho: a block of %u bytes at %p CHANGED after it was freed by thread %d:", k->size, k->ptr, k->tid);
    int shown = 0;
    for (uint32_t i = 0; i < k->size && shown < 12; ++i) if (b[i] != 0xA5) { fprintf(f, " [%u]=%02x", i, b[i]); ++shown; }
    fprintf(f, "\n  freed at:\n");
    for (int i = 0; i < NFR && k->fr[i]; ++i) {
        Dl_info di;
        if (dladdr(k->fr[i], &di) && di.dli_fname)
            fprintf(f, "      %s  %s+%#lx\n", di.dli_fname, di.dli_sname ? di.dli_sname : "?",
                    (unsigned long)((char *)k->fr[i] - (char *)(di.dli_saddr ? di.dli_saddr : di.dli_fbase)));
    }
}
static int park_check(Park *k, FILE *f) {       /* 1: changed */
    const unsigned char *b = (const unsigned char *)k->ptr;
    for (uint32_t i = 0; i < k->size; ++i)
        if (b[i] != 0xA5) { if (f) park_report(f, k, b); return 1; }
    return 0;
}
/* 1: parked (the caller must not free it) */
static int maybe_park(void *p, void *ret0) {
    if (!q_name || busy) return 0;
    if (!q_lo) {
        if ((++q_scans & 1023) != 1) return 0;
        busy = 1; dl_iterate_phdr(phdr_cb, NULL); busy = 0;
        if (!q_lo) return 0;
    }
    const uintptr_t r = (uintptr_t)ret0;
    const int direct = r >= q_lo && r < q_hi;
    if (!direct && !(r >= cxx_lo && r < cxx_hi)) return 0;
    busy = 1;
    void *fr[NFR + 2];
    const int n = backtrace(fr, NFR + 2);
    int from_q = direct;
    for (int i = 2; i < n && i < 5 && !from_q; ++i) from_q = (uintptr_t)fr[i] >= q_lo && (uintptr_t)fr[i] < q_hi;
    const size_t us = malloc_usable_size(p);
    if (!from_q || us < 32 || us > 4096) { busy = 0; return 0; }
    memset(p, 0xA5, us);
    pthread_mutex_lock(&q_mu);
    Park *k = &park[park_ctr++ & (PARK - 1)];
    Park old = *k;
    k->ptr = p; k->size = (uint32_t)us; k->tid = (int32_t)syscall(SYS_gettid);
    memset(k->fr, 0, sizeof k->fr);
    for (int i = 2; i < n; ++i) k->fr[i - 2] = fr[i];
    pthread_mutex_unlock(&q_mu);
    if (old.ptr) {      /* the park is full: the oldest block is checked and really freed */
        if (park_check(&old, NULL)) {
            __sync_fetch_and_add(&q_hits, 1);
            const char *path = getenv("HEAPWHO_REPORT");
            FILE *f = fopen(path ? path : "/tmp/heapwho_report.txt", "a");
            if (f) { park_check(&old, f); fclose(f); }
        }
        real_free(old.ptr);
    }
    busy = 0;
    return 1;
}
/* checks every parked block; returns the number of changed ones (reported to `path`, and re-armed) */
int heapwho_scan(const char *path) {
    busy = 1;
    FILE *f = fopen(path, "a");
    int n = 0, parked = 0;
    pthread_mutex_lock(&q_mu);
    for (uint32_t i = 0; i < PARK; ++i) {
        if (!park[i].ptr) continue;
        ++parked;
        if (park_check(&park[i], f)) { ++n; memset(park[i].ptr, 0xA5, park[i].size); }
    }
    pthread_mutex_unlock(&q_mu);
    if (f) { fprintf(f, "heapwho_scan: %d parked blocks of %s, %d changed (%llu more found when their slot was recycled)\n", parked,
                     q_name ? q_name : "(nothing)", n, (unsigned long long)q_hits); fclose(f); }
    busy = 0;
    return n + (int)q_hits;
}

void *malloc(size_t n) {
    if (!real_malloc) { if (inited) { void *p = boot + boot_used; boot_used += (n + 15) & ~(size_t)15; return p; } init(); }
    void *p = real_malloc(n);
    on_alloc(p, n);
    return p;
}
void free(void *p) {
    if (!p) return;
    if ((char *)p >= boot && (char *)p < boot + sizeof boot) return;
    if (!real_free) init();
    on_free(p);
    if (maybe_park(p, __builtin_return_address(0))) return;
    real_free(p);
}
void *calloc(size_t a, size_t b) {
    if (!real_calloc) {
        if (inited) { void *p = boot + boot_used; boot_used += (a * b + 15) & ~(size_t)15; return p; }   /* dlsym's own calloc */
        init();
        if (!real_calloc) { void *p = boot + boot_used; boot_used += (a * b + 15) & ~(size_t)15; return p; }
    }
    void *p = real_calloc(a, b);
    on_alloc(p, a * b);
    return p;
}
void *realloc(void *p, size_t n) {
    if (!real_realloc) init();
    if (p && (char *)p >= boot && (char *)p < boot + sizeof boot) { void *q = real_malloc(n); if (q) memcpy(q, p, n); return q; }
    if (p) on_free(p);
    void *q = real_realloc(p, n);
    on_alloc(q, n);
    return q;
}
int posix_memalign(void **out, size_t al, size_t n) {
    if (!real_posix_memalign) init();
    int r = real_posix_memalign(out, al, n);
    if (r == 0) on_alloc(*out, n);
    return r;
}
void *aligned_alloc(size_t al, size_t n) {
    if (!real_aligned_alloc) init();
    void *p = real_aligned_alloc(al, n);
    on_alloc(p, n);
    return p;
}

int heapwho_dump(void *ptr, const char *path) {
    busy = 1;
    FILE *f = fopen(path, "a");
    if (!f) { busy = 0; return -1; }
    const uint64_t now = seq_ctr;
    fprintf(f, "heapwho: events of block %p (newest event %llu, ring of %u)\n", ptr, (unsigned long long)now, RING);
    const uint64_t first = now > RING ? now - RING : 0;
    int n = 0;
    for (uint64_t s = first; s < now; ++s) {
        const Ev *e = &ring[s & (RING - 1)];
        if (e->ptr != ptr || e->seq != s) continue;
        fprintf(f, "  #%llu %s %u bytes, thread %d\n", (unsigned long long)s, e->kind == 1 ? "ALLOC" : "FREE ", e->size, e->tid);
        for (int i = 0; i < NFR && e->fr[i]; ++i) {
            Dl_info di;
            if (dladdr(e->fr[i], &di) && di.dli_fname)
                fprintf(f, "      %s  %s+%#lx\n", di.dli_fname, di.dli_sname ? di.dli_sname : "?",
                        (unsigned long)((char *)e->fr[i] - (char *)(di.dli_saddr ? di.dli_saddr : di.dli_fbase)));
            else fprintf(f, "      %p\n", e->fr[i]);
        }
        ++n;
    }
    fprintf(f, "  (%d events; this thread %d)\n", n, (int)syscall(SYS_gettid));
    fclose(f);
    busy = 0;
    return n;
}

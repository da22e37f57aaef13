/* LD_PRELOAD malloc interposer (diagnosis aid, not part of the product): remembers who allocated and who freed every heap
 * block whose request size lies in [HEAPWHO_LO, HEAPWHO_HI] (default 880 .. 936 bytes), so that when a block turns out to have
 * been written through a stale pointer its previous owners can be named.  heapwho_dump(ptr, path) appends the recorded events
 * of that address (allocations and frees, oldest first, six return addresses each, resolved with dladdr) to `path`.
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

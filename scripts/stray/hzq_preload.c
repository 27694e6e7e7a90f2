/* hzq_preload.c -- LD_PRELOAD heap tripwire used to root-cause the "stray element" of
 * NOTES_NEXT.md (a 4-byte zero that appeared in a freshly allocated 912-byte host block after
 * hz_horizon_gridded had returned).  Debug tooling only; never linked into the product.
 *
 * Every heap block whose size lies in [HZQ_MIN, HZQ_MAX] (default 256..4096 bytes) is served from
 * its own page-aligned mapping.  free() does not return the pages: it makes them inaccessible
 * (mprotect PROT_NONE) and parks them in a FIFO of HZQ_CAP mappings.  Any CPU thread that writes
 * (or reads) such a block after it was freed takes a SIGSEGV, and the handler prints
 *   - the backtrace of the faulting thread   (= the stray writer),
 *   - the backtraces of the block's allocation and of its free().
 * HZQ_MODE=fill keeps freed blocks readable instead, fills them with 0xA5 and reports blocks
 * whose pattern was damaged (offset, bytes, owner backtraces) when they leave the FIFO, on
 * hzq_check() and at exit: that also catches a writer that is not a CPU thread (DMA).
 *
 * build: gcc -O1 -g -fPIC -shared -o libhzq.so hzq_preload.c -ldl -lpthread
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <link.h>
#include <execinfo.h>
#include <pthread.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#define NBT 14
typedef struct {
    void *user;            /* address handed to the program (0: slot empty, 1: tombstone) */
    size_t size, maplen;
    void *abt[NBT], *fbt[NBT];
    int na, nf, state;     /* state 1 live, 2 quarantined */
} ent_t;

#define TAB_BITS 19
#define TAB_N (1u << TAB_BITS)
static ent_t *tab;                                 /* open addressing, keyed by user address */
static size_t *fifo; static size_t fifo_cap = 30000, fifo_head, fifo_n;
static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
static size_t q_min = 256, q_max = 4096;
static int mode_fill, want_bt = 1, ready, reports;
static __thread int in_hook;

static void *(*real_malloc)(size_t);
static void (*real_free)(void *);
static void *(*real_calloc)(size_t, size_t);
static void *(*real_realloc)(void *, size_t);
static size_t (*real_usable)(void *);

/* libgcc's unwinder calls malloc / free while it holds its own mutex (__register_frame_info & co.); a backtrace
 * from inside those calls would take that mutex again.  Calls that come from libgcc_s are not backtraced. */
static uintptr_t gcc_lo, gcc_hi;
static int find_libgcc(struct dl_phdr_info *info, size_t sz, void *data) {
    (void)sz; (void)data;
    if (info->dlpi_name && strstr(info->dlpi_name, "libgcc_s")) {
        for (int i = 0; i < info->dlpi_phnum; i++) if (info->dlpi_phdr[i].p_type == PT_LOAD) {
            const uintptr_t a = info->dlpi_addr + info->dlpi_phdr[i].p_vaddr, b = a + info->dlpi_phdr[i].p_memsz;
            if (!gcc_lo || a < gcc_lo) gcc_lo = a;
            if (b > gcc_hi) gcc_hi = b;
        }
    }
    return 0;
}
static int from_libgcc(void *ra) { return (uintptr_t)ra >= gcc_lo && (uintptr_t)ra < gcc_hi; }

static char boot[1 << 16]; static size_t boot_off;   /* dlsym() calls calloc before we are resolved */
static int is_boot(void *p) { return (char *)p >= boot && (char *)p < boot + sizeof(boot); }

static void init(void) {
    static int once;
    if (once) return;
    once = 1;
    in_hook++;
    real_malloc = dlsym(RTLD_NEXT, "malloc"); real_free = dlsym(RTLD_NEXT, "free");
    real_calloc = dlsym(RTLD_NEXT, "calloc"); real_realloc = dlsym(RTLD_NEXT, "realloc");
    real_usable = dlsym(RTLD_NEXT, "malloc_usable_size");
    const char *e;
    if ((e = getenv("HZQ_MIN"))) q_min = (size_t)atol(e);
    if ((e = getenv("HZQ_MAX"))) q_max = (size_t)atol(e);
    if ((e = getenv("HZQ_CAP"))) fifo_cap = (size_t)atol(e);
    if ((e = getenv("HZQ_MODE")) && strcmp(e, "fill") == 0) mode_fill = 1;
    if ((e = getenv("HZQ_BT"))) want_bt = atoi(e);
    tab = mmap(NULL, sizeof(ent_t) * TAB_N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    fifo = mmap(NULL, sizeof(size_t) * fifo_cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    void *warm[4]; backtrace(warm, 4);              /* loads libgcc now, not inside a hook */
    dl_iterate_phdr(find_libgcc, NULL);
    in_hook--;
    ready = 1;
}

static size_t hash_of(void *p) { return (size_t)(((uintptr_t)p >> 12) * 0x9E3779B97F4A7C15ull >> (64 - TAB_BITS)); }
static ent_t *find(void *p) {
    for (size_t i = hash_of(p), n = 0; n < TAB_N; i = (i + 1) & (TAB_N - 1), n++) {
        if (tab[i].user == p) return &tab[i];
        if (tab[i].user == NULL) return NULL;
    }
    return NULL;
}
static ent_t *slot_for(void *p) {
    for (size_t i = hash_of(p), n = 0; n < TAB_N; i = (i + 1) & (TAB_N - 1), n++)
        if (tab[i].user == NULL || tab[i].user == (void *)1) return &tab[i];
    return NULL;
}

static void print_bt(const char *what, void **bt, int n) {
    fprintf(stderr, "[hzq]   %s\n", what);
    if (n > 0) backtrace_symbols_fd(bt, n, 2);
}

static void report_damage(ent_t *e) {
    const unsigned char *u = e->user;
    size_t first = e->size, last = 0, cnt = 0;
    for (size_t k = 0; k < e->size; k++) if (u[k] != 0xA5) { if (k < first) first = k; last = k; cnt++; }
    if (!cnt) return;
    reports++;
    fprintf(stderr, "[hzq] DAMAGED freed block %p size %zu: %zu bytes differ, offsets %zu..%zu, bytes:", e->user, e->size, cnt, first, last);
    for (size_t k = first; k <= last && k < first + 32; k++) fprintf(stderr, " %02x", u[k]);
    fprintf(stderr, "\n");
    print_bt("allocated at:", e->abt, e->na);
    print_bt("freed at:", e->fbt, e->nf);
}

static void evict_one(void) {          /* mu held */
    ent_t *e = &tab[fifo[fifo_head]];
    fifo_head = (fifo_head + 1) % fifo_cap; fifo_n--;
    if (mode_fill) report_damage(e);
    munmap((char *)e->user, e->maplen);
    e->user = (void *)1; e->state = 0;
}

static void segv(int sig, siginfo_t *si, void *ctx) {
    (void)ctx;
    in_hook++;
    void *a = si->si_addr;
    ent_t *hit = NULL;
    for (size_t i = 0; i < TAB_N && !hit; i++)
        if (tab[i].state == 2 && (char *)a >= (char *)tab[i].user && (char *)a < (char *)tab[i].user + tab[i].maplen) hit = &tab[i];
    fprintf(stderr, "[hzq] signal %d at address %p", sig, a);
    if (hit) fprintf(stderr, ": FREED block %p size %zu, offset %zd -- use after free\n", hit->user, hit->size, (char *)a - (char *)hit->user);
    else fprintf(stderr, " (not a quarantined block)\n");
    void *bt[32]; int n = backtrace(bt, 32);
    print_bt("faulting thread:", bt, n);
    if (hit) { print_bt("allocated at:", hit->abt, hit->na); print_bt("freed at:", hit->fbt, hit->nf); }
    _exit(hit ? 97 : 98);
}

__attribute__((constructor)) static void ctor(void) {
    init();
    if (!mode_fill) {
        struct sigaction sa; memset(&sa, 0, sizeof(sa));
        sa.sa_sigaction = segv; sa.sa_flags = SA_SIGINFO | SA_NODEFER;
        sigaction(SIGSEGV, &sa, NULL); sigaction(SIGBUS, &sa, NULL);
    }
    fprintf(stderr, "[hzq] heap tripwire active: sizes %zu..%zu, mode %s, fifo %zu, backtraces %d\n", q_min, q_max, mode_fill ? "fill" : "page", fifo_cap, want_bt);
}

int hzq_check(void) {                 /* scan the quarantine now (fill mode); returns damaged blocks so far */
    if (!ready) return 0;
    pthread_mutex_lock(&mu);
    in_hook++;
    if (mode_fill) for (size_t k = 0, i = fifo_head; k < fifo_n; k++, i = (i + 1) % fifo_cap) {
        ent_t *e = &tab[fifo[i]];
        const unsigned char *u = e->user; int bad = 0;
        for (size_t b = 0; b < e->size; b++) if (u[b] != 0xA5) { bad = 1; break; }
        if (bad) { report_damage(e); memset(e->user, 0xA5, e->size); }
    }
    in_hook--;
    pthread_mutex_unlock(&mu);
    return reports;
}
__attribute__((destructor)) static void dtor(void) { if (ready && mode_fill) { hzq_check(); fprintf(stderr, "[hzq] exit: %d damaged blocks reported\n", reports); } }

static void *q_alloc(size_t size, int no_bt) {
    const size_t maplen = (size + 4095) & ~(size_t)4095;
    char *m = mmap(NULL, maplen, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) return NULL;
    void *bt[NBT];                       /* outside the lock: backtrace() takes the loader lock */
    const int nb = (want_bt && !no_bt) ? backtrace(bt, NBT) : 0;
    pthread_mutex_lock(&mu);
    ent_t *e = slot_for(m);
    if (!e) { pthread_mutex_unlock(&mu); munmap(m, maplen); return NULL; }
    e->user = m; e->size = size; e->maplen = maplen; e->state = 1; e->nf = 0;
    e->na = nb; memcpy(e->abt, bt, sizeof(void *) * (size_t)nb);
    pthread_mutex_unlock(&mu);
    return m;
}

static int q_free(void *p, int no_bt) {           /* 1 if p was ours */
    void *bt[NBT];
    const int nb = (want_bt && !no_bt) ? backtrace(bt, NBT) : 0;
    pthread_mutex_lock(&mu);
    ent_t *e = find(p);
    if (!e || e->state != 1) { pthread_mutex_unlock(&mu); return 0; }
    e->nf = nb; memcpy(e->fbt, bt, sizeof(void *) * (size_t)nb);
    e->state = 2;
    if (mode_fill) memset(p, 0xA5, e->size); else mprotect(p, e->maplen, PROT_NONE);
    if (fifo_n == fifo_cap) evict_one();
    fifo[(fifo_head + fifo_n) % fifo_cap] = (size_t)(e - tab); fifo_n++;
    pthread_mutex_unlock(&mu);
    return 1;
}

void *malloc(size_t size) {
    if (!real_malloc) { init(); if (!real_malloc) { void *p = boot + boot_off; boot_off += (size + 15) & ~(size_t)15; return p; } }
    if (ready && !in_hook && size >= q_min && size <= q_max) {
        in_hook++; void *p = q_alloc(size, from_libgcc(__builtin_return_address(0))); in_hook--;
        if (p) return p;
    }
    return real_malloc(size);
}
void *calloc(size_t n, size_t s) {
    if (!real_calloc) { void *p = boot + boot_off; boot_off += (n * s + 15) & ~(size_t)15; return p; }   /* zeroed static */
    const size_t size = n * s;
    if (ready && !in_hook && size >= q_min && size <= q_max && (s == 0 || size / s == n)) {
        in_hook++; void *p = q_alloc(size, from_libgcc(__builtin_return_address(0))); in_hook--;
        if (p) return p;                /* fresh anonymous pages are zero */
    }
    return real_calloc(n, s);
}
void free(void *p) {
    if (!p || is_boot(p)) return;
    if (ready && !in_hook) { in_hook++; const int ours = q_free(p, from_libgcc(__builtin_return_address(0))); in_hook--; if (ours) return; }
    else if (ready) { pthread_mutex_lock(&mu); ent_t *e = find(p); const int ours = e && e->state == 1; pthread_mutex_unlock(&mu); if (ours) { in_hook++; q_free(p, 1); in_hook--; return; } }
    real_free(p);
}
void *realloc(void *p, size_t size) {
    if (!real_realloc) init();
    if (!p) return malloc(size);
    if (is_boot(p)) { void *q = malloc(size); if (q) memcpy(q, p, size); return q; }
    size_t old = 0; int ours = 0;
    if (ready) { pthread_mutex_lock(&mu); ent_t *e = find(p); if (e && e->state == 1) { ours = 1; old = e->size; } pthread_mutex_unlock(&mu); }
    if (!ours && !(ready && !in_hook && size >= q_min && size <= q_max)) return real_realloc(p, size);
    if (size == 0) { free(p); return NULL; }
    void *q = malloc(size);
    if (!q) return NULL;
    if (!ours) old = real_usable ? real_usable(p) : size;
    memcpy(q, p, old < size ? old : size);
    free(p);
    return q;
}
size_t malloc_usable_size(void *p) {
    if (!p) return 0;
    if (ready) { pthread_mutex_lock(&mu); ent_t *e = find(p); size_t s = (e && e->state == 1) ? e->size : 0; int ours = e && e->state == 1; pthread_mutex_unlock(&mu); if (ours) return s; }
    if (is_boot(p)) return 0;
    return real_usable ? real_usable(p) : 0;
}

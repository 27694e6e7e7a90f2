/* hzwatch.c -- hardware write watchpoint on one 4-byte address for EVERY thread of the process
 * (perf_event_open breakpoints, one per thread, synchronous SIGTRAP in the writing thread).  Debug tooling for
 * the "stray element" investigation (NOTES / DESIGN): prints the backtrace of whoever stores to the address and
 * keeps per-thread hit counts (a count without a backtrace = the store was made in kernel mode on behalf of that
 * thread, e.g. the tid word the kernel clears when a thread exits).
 * build: gcc -O1 -g -fPIC -shared -o libhzwatch.so hzwatch.c */
#define _GNU_SOURCE
#include <dirent.h>
#include <execinfo.h>
#include <fcntl.h>
#include <linux/hw_breakpoint.h>
#include <linux/perf_event.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/ioctl.h>
#include <sys/syscall.h>
#include <unistd.h>

static volatile int g_hits;
static int g_fds[4096], g_tids[4096]; static char g_comm[4096][20]; static int g_nfd;
static int g_max_bt = 24;

static void on_trap(int sig, siginfo_t *si, void *ctx) {
    (void)ctx;
    if (++g_hits > g_max_bt) return;
    char buf[256];
    int n = snprintf(buf, sizeof(buf), "[hzwatch] WRITE to watched address by tid %ld (signal %d code %d addr %p), backtrace:\n",
                     (long)syscall(SYS_gettid), sig, si->si_code, si->si_addr);
    if (write(2, buf, n) < 0) return;
    void *bt[48]; int k = backtrace(bt, 48);
    backtrace_symbols_fd(bt, k, 2);
}

/* arm a 4-byte write watchpoint on addr for every thread of this process; returns the number of threads armed, <0 on error */
int hzwatch_arm(void *addr) {
    struct sigaction sa; memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_trap; sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGTRAP, &sa, NULL);
    void *warm[4]; backtrace(warm, 4);
    DIR *d = opendir("/proc/self/task");
    if (!d) return -1;
    struct dirent *e; int armed = 0, err = 0, kern_excluded = 0;
    while ((e = readdir(d))) {
        if (e->d_name[0] == '.') continue;
        pid_t tid = (pid_t)atoi(e->d_name);
        struct perf_event_attr a; memset(&a, 0, sizeof(a));
        a.type = PERF_TYPE_BREAKPOINT; a.size = sizeof(a);
        a.bp_type = HW_BREAKPOINT_W; a.bp_addr = (uint64_t)(uintptr_t)addr; a.bp_len = HW_BREAKPOINT_LEN_4;
        a.sample_period = 1; a.exclude_kernel = 0; a.exclude_hv = 1;
        a.sigtrap = 1; a.remove_on_exec = 1;   /* synchronous SIGTRAP in the thread that wrote */
        a.inherit = 1; a.inherit_thread = 1;   /* threads created later by an armed thread are covered too */
        int fd = (int)syscall(SYS_perf_event_open, &a, tid, -1, -1, PERF_FLAG_FD_CLOEXEC);
        if (fd < 0) { a.exclude_kernel = 1; fd = (int)syscall(SYS_perf_event_open, &a, tid, -1, -1, PERF_FLAG_FD_CLOEXEC); if (fd >= 0) kern_excluded++; }
        if (fd < 0) { err++; continue; }
        if (g_nfd < 4096) {
            g_fds[g_nfd] = fd; g_tids[g_nfd] = tid;
            char path[64]; snprintf(path, sizeof(path), "/proc/self/task/%d/comm", tid);
            FILE *f = fopen(path, "r"); g_comm[g_nfd][0] = 0;
            if (f) { if (fgets(g_comm[g_nfd], 20, f)) g_comm[g_nfd][strcspn(g_comm[g_nfd], "\n")] = 0; fclose(f); }
            g_nfd++;
        }
        armed++;
    }
    closedir(d);
    fprintf(stderr, "[hzwatch] armed %d threads on %p (%d failures, %d without kernel-mode coverage)\n", armed, addr, err, kern_excluded);
    return armed ? armed : -2;
}
/* per-thread hit counts (the perf counters, kernel-mode stores included), then release the watchpoints */
int hzwatch_disarm(void) {
    for (int i = 0; i < g_nfd; i++) {
        uint64_t c = 0;
        if (read(g_fds[i], &c, 8) == 8 && c) fprintf(stderr, "[hzwatch] tid %d (%s): %llu stores counted\n", g_tids[i], g_comm[i], (unsigned long long)c);
        close(g_fds[i]);
    }
    g_nfd = 0;
    fprintf(stderr, "[hzwatch] disarmed: %d stores delivered a signal\n", g_hits);
    return g_hits;
}
int hzwatch_hits(void) { return g_hits; }

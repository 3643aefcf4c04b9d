// stackdump.c -- debugging aid for stalls inside native code (tools/ddp_probe.py): a SIGUSR2
// handler that writes the receiving thread's native backtrace (glibc backtrace(), dynamic symbol
// names) to a file descriptor, and a helper that signals every thread of the process in turn.
// Host code only; never loaded by the product.
#define _GNU_SOURCE
#include <dirent.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static int g_fd = 2;

static void handler(int sig)
{
    (void)sig;
    void *frames[64];
    char head[96];
    const int n = backtrace(frames, 64);
    const int len = snprintf(head, sizeof head, "---- native stack of tid %ld ----\n", (long)syscall(SYS_gettid));
    if (write(g_fd, head, (size_t)len) < 0) return;
    backtrace_symbols_fd(frames, n, g_fd);
}

static void crash_handler(int sig)
{
    void *frames[64];
    char head[96];
    const int n = backtrace(frames, 64);
    const int len = snprintf(head, sizeof head, "---- signal %d in tid %ld: native stack ----\n", sig,
                             (long)syscall(SYS_gettid));
    if (write(g_fd, head, (size_t)len) >= 0) backtrace_symbols_fd(frames, n, g_fd);
    signal(sig, SIG_DFL);  // returning re-executes the faulting instruction: default action now
}

// also dump the faulting thread's stack on SIGSEGV / SIGBUS / SIGABRT (then die as usual)
int stackdump_install_crash(void)
{
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = crash_handler;
    sa.sa_flags = SA_NODEFER;
    int rc = sigaction(SIGSEGV, &sa, NULL);
    rc |= sigaction(SIGBUS, &sa, NULL);
    rc |= sigaction(SIGABRT, &sa, NULL);
    return rc;
}

int stackdump_install(int fd)
{
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = handler;
    sa.sa_flags = SA_RESTART;
    g_fd = fd;
    void *warm[4];
    backtrace(warm, 4);  // loads libgcc now, not inside the handler
    return sigaction(SIGUSR2, &sa, NULL);
}

// signal every thread of this process, one at a time (100 ms apart so the dumps do not interleave)
int stackdump_all(void)
{
    DIR *d = opendir("/proc/self/task");
    if (!d) return -1;
    struct dirent *e;
    const pid_t pid = getpid();
    int n = 0;
    while ((e = readdir(d))) {
        const long tid = strtol(e->d_name, NULL, 10);
        if (tid <= 0) continue;
        syscall(SYS_tgkill, pid, (pid_t)tid, SIGUSR2);
        usleep(100 * 1000);
        ++n;
    }
    closedir(d);
    return n;
}

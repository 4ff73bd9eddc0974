/* LD_PRELOAD helper: on SIGSEGV / SIGBUS / SIGABRT print a native backtrace (glibc backtrace_symbols_fd) and the mapped libraries' load
 * addresses to stderr, then _exit(139).  Python's faulthandler is already gone when a process dies inside C exit handlers.
 *   gcc -shared -fPIC -O1 -o segv_bt.so segv_bt.c && LD_PRELOAD=./segv_bt.so python3 ... */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>

static void handler(int sig, siginfo_t* si, void* ctx) {
  (void)ctx;
  char buf[128];
  int n = snprintf(buf, sizeof buf, "\n[segv_bt] signal %d at address %p, tid %d\n", sig, si ? si->si_addr : 0, (int)gettid());
  if (write(2, buf, (size_t)n) < 0) _exit(140);
  void* frames[64];
  int nf = backtrace(frames, 64);
  backtrace_symbols_fd(frames, nf, 2);
  _exit(139);
}

__attribute__((constructor)) static void install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = handler;
  sa.sa_flags = SA_SIGINFO | SA_RESETHAND;
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
  sigaction(SIGABRT, &sa, 0);
}

// Debug aid: print a native backtrace on SIGSEGV, also after a stack overflow (alternate signal stack).
// Loaded through ctypes by tools/ring_probe.py.
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static void on_segv(int sig, siginfo_t *si, void *ctx)
{
  (void)ctx;
  void *frames[96];
  int n = backtrace(frames, 96);
  fprintf(stderr, "== native backtrace (signal %d, address %p) ==\n", sig, si ? si->si_addr : 0);
  backtrace_symbols_fd(frames, n, 2);
  _exit(139);
}
void install_segv_bt(void)
{
  static char stack[1 << 18];
  stack_t ss;
  ss.ss_sp = stack;
  ss.ss_size = sizeof(stack);
  ss.ss_flags = 0;
  sigaltstack(&ss, 0);
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = on_segv;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  sigaction(SIGSEGV, &sa, 0);
}

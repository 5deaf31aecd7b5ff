// Fault diagnostics of the host side of the library (no device code).
//
// Round 4 recorded two host segmentation faults inside the HIP runtime (hipGraphLaunch of a forked graph)
// of which only the PYTHON frames survived: pytest's faulthandler prints those and nothing below them.  With
// PSEG_SEGV_BACKTRACE=1 in the environment when the library is loaded, SIGSEGV / SIGBUS / SIGFPE / SIGILL / SIGABRT
// first write the NATIVE frames of the faulting thread to stderr (backtrace_symbols_fd: async-signal-safe, no malloc),
// then hand over to whatever handler was installed before (Python's faulthandler under pytest: its Python frames follow),
// or re-raise with the default action so that the exit status and the core dump stay what they would have been.
// tests/conftest.py switches it on for every test process; bench.py / train.py leave it to the user.
//
// Alternate stack: sigaltstack is a PER-THREAD setting and install() runs on the thread that loads the library (the Python
// main thread), so only a stack-exhaustion fault on THAT thread gets its trace on the handler's own stack; any other thread
// (the autograd worker, host-function threads) runs the handler on its own stack -- ordinary faults there are traced all the
// same, a stack overflow there is not.  An alternate stack that is already installed and at least as large (Python's
// faulthandler sets one up for the main thread) is KEPT, not replaced.
#include <execinfo.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "../../include/pseg_amd.h"

namespace {

constexpr int kSignals[] = {SIGSEGV, SIGBUS, SIGFPE, SIGILL, SIGABRT};
constexpr int kNumSignals = sizeof(kSignals) / sizeof(kSignals[0]);
struct sigaction g_previous[kNumSignals];
bool g_installed = false;
// the handler's own stack: a fault from stack exhaustion still gets its trace
alignas(16) char g_altstack[64 * 1024];

void write_str(const char* s) {
  ssize_t r = write(STDERR_FILENO, s, strlen(s));
  (void)r;
}

void write_int(long v) {
  char buf[24];
  int n = 0;
  if (v == 0) buf[n++] = '0';
  const bool neg = v < 0;
  unsigned long u = neg ? (unsigned long)(-v) : (unsigned long)v;
  while (u > 0 && n < 22) {
    buf[n++] = (char)('0' + u % 10);
    u /= 10;
  }
  if (neg) buf[n++] = '-';
  for (int i = 0; i < n / 2; ++i) {
    const char t = buf[i];
    buf[i] = buf[n - 1 - i];
    buf[n - 1 - i] = t;
  }
  buf[n] = 0;
  write_str(buf);
}

void on_fault(int sig, siginfo_t* info, void* ctx) {
  write_str("\n[pseg] fatal signal ");
  write_int(sig);
  write_str(" (");
  write_str(sig == SIGSEGV ? "SIGSEGV" : sig == SIGBUS ? "SIGBUS" : sig == SIGFPE ? "SIGFPE" : sig == SIGILL ? "SIGILL" : "SIGABRT");
  write_str("), native frames of the faulting thread:\n");
  void* frames[96];
  const int n = backtrace(frames, 96);
  backtrace_symbols_fd(frames, n, STDERR_FILENO);
  write_str("[pseg] end of native frames\n");
  int slot = -1;
  for (int i = 0; i < kNumSignals; ++i)
    if (kSignals[i] == sig) slot = i;
  // the earlier handler (Python's faulthandler prints the Python frames and re-raises), else the default action
  if (slot >= 0) {
    const struct sigaction& prev = g_previous[slot];
    if ((prev.sa_flags & SA_SIGINFO) && prev.sa_sigaction != nullptr) {
      sigaction(sig, &prev, nullptr);
      prev.sa_sigaction(sig, info, ctx);
      return;
    }
    if (!(prev.sa_flags & SA_SIGINFO) && prev.sa_handler != SIG_DFL && prev.sa_handler != SIG_IGN && prev.sa_handler != nullptr) {
      sigaction(sig, &prev, nullptr);
      prev.sa_handler(sig);
      return;
    }
  }
  signal(sig, SIG_DFL);
  raise(sig);
}

int install() {
  if (g_installed) return 0;
  // backtrace() loads libgcc on first use (malloc): do that now, not inside a signal handler
  void* warm[4];
  (void)backtrace(warm, 4);
  stack_t ss, old;
  memset(&ss, 0, sizeof(ss));
  memset(&old, 0, sizeof(old));
  ss.ss_sp = g_altstack;
  ss.ss_size = sizeof(g_altstack);
  const bool have = sigaltstack(nullptr, &old) == 0 && !(old.ss_flags & SS_DISABLE) && old.ss_sp != nullptr &&
                    old.ss_size >= sizeof(g_altstack);
  if (!have) (void)sigaltstack(&ss, nullptr);      // (an installed stack of at least this size stays: see the header comment)
  for (int i = 0; i < kNumSignals; ++i) {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_fault;
    sigemptyset(&sa.sa_mask);
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
    if (sigaction(kSignals[i], &sa, &g_previous[i]) != 0) return -1;
  }
  g_installed = true;
  return 0;
}

struct AtLoad {
  AtLoad() {
    const char* e = getenv("PSEG_SEGV_BACKTRACE");
    if (e != nullptr && atoi(e) != 0) (void)install();
  }
} g_at_load;

}  // namespace

extern "C" {

int pseg_fault_backtrace_enable(void) { return install() == 0 ? PSEG_OK : PSEG_ERR_ARG; }

int pseg_fault_backtrace_enabled(void) { return g_installed ? 1 : 0; }

}  // extern "C"

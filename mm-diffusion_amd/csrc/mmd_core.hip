// Error plumbing, version and HIP-graph capture helpers of the C-ABI.
#include "mmd_common.h"
#include <string.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

static thread_local char g_err[512] = "";

int mmd_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int mmd_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
  return MMD_OK;
}

extern "C" const char* mmd_last_error(void) { return g_err; }
extern "C" int mmd_version(void) { return 100; }   // 0.1.0

// ---- HIP graphs: capture one denoising step once, replay it per step (no tracing compiler involved)
extern "C" int mmd_graph_begin(void* stream) {
  hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "graph_begin: %s", hipGetErrorString(e));
  return MMD_OK;
}
extern "C" int mmd_graph_end(void* stream, void** exec_out) {
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
  if (e != hipSuccess || !g) return mmd_set_error(MMD_ERR_LAUNCH, "graph_end: %s", hipGetErrorString(e));
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "graph_instantiate: %s", hipGetErrorString(e));
  *exec_out = (void*)ex;
  return MMD_OK;
}
extern "C" int mmd_graph_launch(void* exec, void* stream) {
  hipError_t e = hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream);
  if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "graph_launch: %s", hipGetErrorString(e));
  return MMD_OK;
}
extern "C" int mmd_graph_destroy(void* exec) {
  if (exec) (void)hipGraphExecDestroy((hipGraphExec_t)exec);
  return MMD_OK;
}

// ---- launch streams owned by the host mirror (video / audio chains, capture stream): private non-blocking streams, so the
// library's fork/join never aliases a stream of the framework's pool
extern "C" int mmd_stream_create(void** stream_out) {
  hipStream_t s = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "stream_create: %s", hipGetErrorString(e));
  *stream_out = (void*)s;
  return MMD_OK;
}
extern "C" int mmd_stream_sync(void* stream) {
  hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "stream_sync: %s", hipGetErrorString(e));
  return MMD_OK;
}
extern "C" int mmd_stream_destroy(void* stream) {
  if (stream) (void)hipStreamDestroy((hipStream_t)stream);
  return MMD_OK;
}

// ---- stream-ordered timing on the stream the kernels are launched on (bench.py roofline leg)
extern "C" int mmd_event_create(void** ev) {
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "event_create failed");
  *ev = (void*)e;
  return MMD_OK;
}
extern "C" int mmd_event_record(void* ev, void* stream) {
  if (hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "event_record failed");
  return MMD_OK;
}
extern "C" int mmd_stream_wait_event(void* stream, void* ev) {
  hipError_t e = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0);
  if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "stream_wait_event: %s", hipGetErrorString(e));
  return MMD_OK;
}
extern "C" int mmd_event_elapsed_ms(void* a, void* b, float* ms) {
  hipError_t e = hipEventSynchronize((hipEvent_t)b);
  if (e == hipSuccess) e = hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b);
  if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "event_elapsed: %s", hipGetErrorString(e));
  return MMD_OK;
}
extern "C" int mmd_event_destroy(void* ev) {
  if (ev) (void)hipEventDestroy((hipEvent_t)ev);
  return MMD_OK;
}

// ---- debugging aid for the test suite: native backtrace on a fatal signal.  Python's faulthandler prints the Python stack of a
// SIGSEGV inside a ctypes call but not WHICH native frame faulted (e.g. inside hipStreamEndCapture); this handler writes the native
// frames (module + symbol/offset) to stderr and then chains to the handler that was installed before it.
static struct sigaction g_prev_sa[3];
static const int g_sigs[3] = {SIGSEGV, SIGBUS, SIGABRT};
static void mmd_crash_handler(int sig, siginfo_t* info, void* uc) {
  static const char msg[] = "\n[libmmd] fatal signal - native backtrace:\n";
  (void)!write(2, msg, sizeof(msg) - 1);
  void* frames[64];
  const int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  for (int i = 0; i < 3; ++i) {
    if (g_sigs[i] != sig) continue;
    if (g_prev_sa[i].sa_flags & SA_SIGINFO) {
      if (g_prev_sa[i].sa_sigaction) { g_prev_sa[i].sa_sigaction(sig, info, uc); return; }
    } else if (g_prev_sa[i].sa_handler != SIG_DFL && g_prev_sa[i].sa_handler != SIG_IGN) {
      g_prev_sa[i].sa_handler(sig);
      return;
    }
    sigaction(sig, &g_prev_sa[i], nullptr);     // default action: re-raise
    raise(sig);
  }
}
extern "C" int mmd_debug_install_crash_handler(void) {
  void* warm[4];
  (void)backtrace(warm, 4);                     // loads libgcc now: backtrace() must not dlopen inside the handler
  for (int i = 0; i < 3; ++i) {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = mmd_crash_handler;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    if (sigaction(g_sigs[i], &sa, &g_prev_sa[i]) != 0) return mmd_set_error(MMD_ERR_LAUNCH, "sigaction failed");
  }
  return MMD_OK;
}

// Zero a device buffer on the stream (a memset node under capture).
extern "C" int mmd_zero(void* ptr, int64_t bytes, void* stream) {
  MMD_REQUIRE(ptr && bytes > 0, "mmd_zero: null pointer / empty");
  hipError_t e = hipMemsetAsync(ptr, 0, (size_t)bytes, (hipStream_t)stream);
  return e == hipSuccess ? MMD_OK : mmd_set_error(MMD_ERR_LAUNCH, "mmd_zero: %s", hipGetErrorString(e));
}

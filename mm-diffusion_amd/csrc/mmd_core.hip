// Error plumbing, version and HIP-graph capture helpers of the C-ABI.
#include "mmd_common.h"
#include <string.h>

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

// Error string + version entry points of the C ABI (include/clsr_hip.h).
#include "common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void clsr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int clsr_version(void) { return 100; }

// Copies the calling thread's last error message (NUL terminated); returns its length.
extern "C" int clsr_last_error(char* buf, size_t n) {
  if (buf && n) {
    strncpy(buf, g_err, n - 1);
    buf[n - 1] = 0;
  }
  return (int)strlen(g_err);
}

// Stream capture -> executable graph, so a whole training step replays as one launch.
extern "C" int clsr_graph_begin(void* stream) {
  CLSR_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return CLSR_OK;
}

extern "C" int clsr_graph_end(void* stream, void** graph_exec_out) {
  CLSR_CHECK_ARG(graph_exec_out);
  hipGraph_t graph = nullptr;
  CLSR_HIP(hipStreamEndCapture((hipStream_t)stream, &graph));
  hipGraphExec_t exec = nullptr;
  CLSR_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CLSR_HIP(hipGraphDestroy(graph));
  *graph_exec_out = (void*)exec;
  return CLSR_OK;
}

extern "C" int clsr_graph_launch(void* graph_exec, void* stream) {
  CLSR_CHECK_ARG(graph_exec);
  CLSR_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return CLSR_OK;
}

extern "C" int clsr_graph_destroy(void* graph_exec) {
  if (graph_exec) CLSR_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return CLSR_OK;
}

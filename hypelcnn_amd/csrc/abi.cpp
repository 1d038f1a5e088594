// libhypel_hip.so: error channel, device query, HIP-graph capture helpers.
#include <stdarg.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void hypel_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int hypel_version(void) { return HYPEL_ABI_VERSION; }

extern "C" const char* hypel_last_error(void) { return g_err; }

extern "C" int hypel_device_info(int32_t* n_cu, int32_t* n_xcd) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        hypel_set_error("hypel_device_info: %s", hipGetErrorString(e));
        return -2;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) {
        hypel_set_error("hypel_device_info: %s", hipGetErrorString(e));
        return -2;
    }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (n_xcd) *n_xcd = 8;  // MI355X: 8 XCDs x 32 CUs
    return 0;
}

static int dep(hipStream_t from, hipStream_t to, const char* what) {
    hipEvent_t ev = nullptr;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ev, from);
    if (e == hipSuccess) e = hipStreamWaitEvent(to, ev, 0);
    if (ev) (void)hipEventDestroy(ev);  // destruction is deferred until the event completes
    if (e != hipSuccess) {
        hypel_set_error("%s: %s", what, hipGetErrorString(e));
        return -2;
    }
    return 0;
}

extern "C" int hypel_stream_fork(hypel_stream_t main_stream, hypel_stream_t side_stream) {
    return dep((hipStream_t)main_stream, (hipStream_t)side_stream, "hypel_stream_fork");
}

extern "C" int hypel_stream_join(hypel_stream_t main_stream, hypel_stream_t side_stream) {
    return dep((hipStream_t)side_stream, (hipStream_t)main_stream, "hypel_stream_join");
}

extern "C" int hypel_graph_begin_capture(hypel_stream_t stream) {
    hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
        hypel_set_error("hypel_graph_begin_capture: %s", hipGetErrorString(e));
        return -2;
    }
    return 0;
}

extern "C" int hypel_graph_end_capture(hypel_stream_t stream, void** graph_exec_out) {
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &graph);
    if (e != hipSuccess || !graph) {
        hypel_set_error("hypel_graph_end_capture: %s", hipGetErrorString(e));
        return -2;
    }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        hypel_set_error("hypel_graph_end_capture: instantiate: %s", hipGetErrorString(e));
        return -2;
    }
    *graph_exec_out = (void*)exec;
    return 0;
}

extern "C" int hypel_graph_launch(void* graph_exec, hypel_stream_t stream) {
    hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
    if (e != hipSuccess) {
        hypel_set_error("hypel_graph_launch: %s", hipGetErrorString(e));
        return -2;
    }
    return 0;
}

extern "C" int hypel_graph_destroy(void* graph_exec) {
    if (graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
    return 0;
}

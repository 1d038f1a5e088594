// libhypel_hip.so: error channel, device query, HIP-graph capture helpers.
#include <cstdlib>
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

// CRC-32C (Castagnoli), slicing-by-8, host side: TensorFlow checkpoint bundles carry a masked CRC-32C per tensor and
// per index block (tensorflow/core/lib/hash/crc32c.h); Python has no fast implementation of this polynomial.
static uint32_t g_crc_tab[8][256];
static bool g_crc_ready = false;
static void crc_init() {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82f63b78u : c >> 1;
        g_crc_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t) g_crc_tab[t][i] = (g_crc_tab[t - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][i] & 0xff];
    g_crc_ready = true;
}

extern "C" uint32_t hypel_crc32c(uint32_t crc, const void* data, uint64_t n) {
    if (!g_crc_ready) crc_init();
    const uint8_t* p = (const uint8_t*)data;
    uint32_t c = crc ^ 0xffffffffu;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = g_crc_tab[7][lo & 0xff] ^ g_crc_tab[6][(lo >> 8) & 0xff] ^ g_crc_tab[5][(lo >> 16) & 0xff] ^
            g_crc_tab[4][lo >> 24] ^ g_crc_tab[3][hi & 0xff] ^ g_crc_tab[2][(hi >> 8) & 0xff] ^
            g_crc_tab[1][(hi >> 16) & 0xff] ^ g_crc_tab[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = g_crc_tab[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return c ^ 0xffffffffu;
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

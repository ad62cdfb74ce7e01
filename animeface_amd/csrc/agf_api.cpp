// Host-side plumbing of libagf_ops.so: error string, ABI version, device query.
#include "agf_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void agf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* agf_last_error(void) { return g_err; }

// Deterministic mode (process-wide): every launch whose reduction is normally finished with fp32 atomics from several workgroups is
// re-shaped so that each output element has ONE writer (see agf_set_deterministic in include/agf_ops.h).
static int g_deterministic = 0;
int agf_deterministic(void) { return g_deterministic; }
extern "C" int agf_set_deterministic(int on) { const int old = g_deterministic; g_deterministic = on ? 1 : 0; return old; }
extern "C" int agf_get_deterministic(void) { return g_deterministic; }
extern "C" int agf_abi_version(void) { return AGF_ABI_VERSION; }

// hipMemsetAsync on the caller's stream through the HIP runtime THIS library is linked against (the one torch's stream handle belongs
// to): under stream capture it records a memset node, which is what the caller wants it for.
extern "C" int agf_memset_node(void* buf, int value, int64_t nbytes, void* stream) {
    if (!buf || nbytes <= 0) { agf_set_error("agf_memset_node: null buffer or non-positive size"); return AGF_EINVAL; }
    const hipError_t e = hipMemsetAsync(buf, value, (size_t)nbytes, (hipStream_t)stream);
    if (e != hipSuccess) { agf_set_error("agf_memset_node: hipMemsetAsync failed: %s", hipGetErrorString(e)); return AGF_ELAUNCH; }
    return AGF_OK;
}

extern "C" int agf_device_info(int* cu_count, int* lds_bytes_per_block, int* wavefront_size) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { agf_set_error("no HIP device"); return AGF_ELAUNCH; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { agf_set_error("hipGetDeviceProperties failed"); return AGF_ELAUNCH; }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_bytes_per_block) *lds_bytes_per_block = (int)prop.sharedMemPerBlock;
    if (wavefront_size) *wavefront_size = prop.warpSize;
    return AGF_OK;
}

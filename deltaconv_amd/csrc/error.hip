// Error string + version of the C-ABI library.
#include <cstring>
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void dc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

DC_EXPORT const char* dc_last_error(void) { return g_err; }
DC_EXPORT int32_t dc_version(void) { return 100; }  // 0.1.0 -> major*10000 + minor*100 + patch

static int g_opt[DC_OPT_COUNT] = {1};  // XCD remap on by default
int dc_option(int key) { return (key >= 0 && key < DC_OPT_COUNT) ? g_opt[key] : 0; }
// Experiment switch for A/B measurements (key 0: XCD-aware block remap, default 1).
DC_EXPORT int dc_set_option(int32_t key, int32_t value) {
    if (key < 0 || key >= DC_OPT_COUNT) return DC_ERR_ARG;
    g_opt[key] = value;
    return DC_OK;
}

// ---- dynamic LDS opt-in (common.h) ---------------------------------------------------------------------------------------
static thread_local bool g_lds_failed = false;
bool dc_take_lds_failure() {
    const bool f = g_lds_failed;
    g_lds_failed = false;
    return f;
}
bool dc_ensure_lds(unsigned long long* done_mask, const void* kernel, size_t bytes, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = dev < 64 ? 1ull << dev : 0ull;
    if (bit && (__atomic_load_n(done_mask, __ATOMIC_ACQUIRE) & bit)) return true;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        dc_set_error("%s: needs %zu KiB of LDS per workgroup (MI355X / gfx950 offers 160 KiB); device %d refused: %s", what,
                     bytes / 1024, dev, hipGetErrorString(e));
        g_lds_failed = true;
        return false;
    }
    if (bit) __atomic_fetch_or(done_mask, bit, __ATOMIC_RELEASE);
    return true;
}

// ---- deferred finalisers (common.h) -------------------------------------------------------------------------------------------
static thread_local DcFinPending g_fin[DC_FIN_MAX];
static thread_local int g_fin_count = 0;
static thread_local bool g_fin_open = false, g_fin_request = false, g_gemm_request = false;
bool dc_gemm_take_request() {
    const bool take = g_fin_open && g_gemm_request && g_fin_request;   // (a queued product needs its finaliser queued behind it)
    g_gemm_request = false;
    return take;
}
bool dc_fin_take_request() {
    const bool take = g_fin_open && g_fin_request && g_fin_count < DC_FIN_MAX;
    g_fin_request = false;
    return take;
}
void dc_fin_push(int kind, const double* partial, int chunks, int C, const void* fin, size_t bytes, const void* functor,
                 size_t functor_bytes, long R, int rpc) {
    DcFinPending& e = g_fin[g_fin_count++];
    e.kind = kind; e.partial = partial; e.chunks = chunks; e.C = C;
    memcpy(e.blob, fin, bytes);
    e.stage1 = functor != nullptr; e.R = R; e.rpc = rpc;
    if (functor) memcpy(e.functor, functor, functor_bytes);
}
int dc_fin_pending(DcFinPending** out) { *out = g_fin; return g_fin_count; }
void dc_fin_clear() { g_fin_count = 0; g_fin_open = false; g_fin_request = false; g_gemm_request = false; }
DC_EXPORT int dc_finalisers_begin(void) {
    if (g_fin_open) {
        dc_set_error("dc_finalisers_begin: a batch is already open on this thread");
        return DC_ERR_ARG;
    }
    g_fin_open = true; g_fin_count = 0; g_fin_request = false;
    return DC_OK;
}
DC_EXPORT int dc_finaliser_defer_next(void) {
    g_fin_request = g_fin_open;
    return DC_OK;
}
DC_EXPORT int dc_gemm_defer_next(void) {
    g_gemm_request = g_fin_open;
    return DC_OK;
}

// ---- device-clock stamps (common.h) --------------------------------------------------------------------------------------
static unsigned long long* g_stamp_buf = nullptr;
static int g_stamp_slots = 0, g_stamp_used = 0;
static int g_stamp_tags[4096];
unsigned long long* dc_stamp_next(int tag) {
    if (!g_stamp_buf || g_stamp_used >= g_stamp_slots) return nullptr;
    g_stamp_tags[g_stamp_used] = tag;
    return g_stamp_buf + 4 * (size_t)g_stamp_used++;
}
// buf: device memory of slots x 4 x 8 bytes (NULL / 0 disarms); restarts the record counter
DC_EXPORT int dc_stamp_buffer(uint64_t* buf, int32_t slots) {
    if (slots < 0 || slots > 4096) {
        dc_set_error("dc_stamp_buffer: at most 4096 records");
        return DC_ERR_ARG;
    }
    g_stamp_buf = reinterpret_cast<unsigned long long*>(buf);
    g_stamp_slots = buf ? slots : 0;
    g_stamp_used = 0;
    return DC_OK;
}
DC_EXPORT int32_t dc_stamp_count(void) { return g_stamp_used; }
DC_EXPORT int32_t dc_stamp_tag(int32_t record) { return (record >= 0 && record < g_stamp_used) ? g_stamp_tags[record] : -1; }

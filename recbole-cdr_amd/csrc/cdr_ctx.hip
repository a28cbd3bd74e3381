// Context, error reporting, ABI version.
#include <stdarg.h>
#include <string.h>
#include "cdr_common.h"
#include <stdlib.h>

static thread_local char g_err[512] = "";

void cdr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* cdr_last_error(void) { return g_err; }
extern "C" int cdr_abi_version(void) { return CDR_ABI_VERSION; }

extern "C" int cdr_ctx_create(int device, cdr_ctx** out) {
    CDR_CHECK_ARG(out != nullptr);
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        cdr_set_error("cdr_ctx_create: device %d not available (%d visible)", device, ndev);
        return CDR_ENODEV;
    }
    int prev = 0;
    CDR_HIP(hipGetDevice(&prev));
    CDR_HIP(hipSetDevice(device));
    cdr_ctx* c = new cdr_ctx();
    c->device = device;
    c->timing_cap = c->timing_n = 0;
    c->ev0 = c->ev1 = nullptr;
    c->tags = nullptr;
    c->scratch = nullptr;
    c->scratch_bytes = 0;
    c->scrub_ptr = nullptr; c->scrub_bytes = 0;
    c->idc_user = c->idc_item = nullptr; c->idc_user_rows = c->idc_item_rows = 0; c->idc_list = nullptr; c->idc_list_bytes = 0;
    c->partials = nullptr;
    c->tickets = nullptr;
    hipError_t e = hipMalloc(&c->partials, sizeof(double) * CDR_MAX_PARTIAL_BLOCKS * CDR_PARTIAL_STRIDE);
    if (e == hipSuccess) e = hipMalloc(&c->tickets, sizeof(unsigned) * CDR_TICKETS);
    if (e == hipSuccess) e = hipMemset(c->tickets, 0, sizeof(unsigned) * CDR_TICKETS);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) {
        if (c->partials) (void)hipFree(c->partials);
        if (c->tickets) (void)hipFree(c->tickets);
        delete c;
        cdr_set_error("cdr_ctx_create: scratch allocation failed: %s", hipGetErrorString(e));
        return CDR_ENOMEM;
    }
    *out = c;
    return CDR_OK;
}

// Device scratch owned by the context, grown (never shrunk) on demand.  hipFree synchronises the device, so work still
// reading the old block has finished before it is released.  A context serves ONE stream at a time (the host binding
// keeps one per (device, stream)): its scratch and reduction partials are not shared between concurrent launches.
int cdr_ctx_scratch(cdr_ctx* ctx, size_t bytes, void** out) {
    if (ctx->scratch_bytes < bytes) {
        int prev = 0;
        CDR_HIP(hipGetDevice(&prev));
        CDR_HIP(hipSetDevice(ctx->device));
        if (ctx->scratch) (void)hipFree(ctx->scratch);
        ctx->scratch = nullptr; ctx->scratch_bytes = 0;
        const size_t want = bytes + bytes / 4;
        hipError_t e = hipMalloc(&ctx->scratch, want);
        (void)hipSetDevice(prev);
        if (e != hipSuccess) { cdr_set_error("cdr_ctx_scratch: %zu bytes: %s", want, hipGetErrorString(e)); return CDR_ENOMEM; }
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return CDR_OK;
}

extern "C" int cdr_timing_enable(cdr_ctx* ctx, int capacity) {
    CDR_CHECK_ARG(ctx && capacity >= 0);
    for (int i = 0; i < ctx->timing_cap; ++i) { (void)hipEventDestroy(ctx->ev0[i]); (void)hipEventDestroy(ctx->ev1[i]); }
    delete[] ctx->ev0; delete[] ctx->ev1; delete[] ctx->tags;
    ctx->ev0 = ctx->ev1 = nullptr; ctx->tags = nullptr;
    ctx->timing_cap = ctx->timing_n = 0;
    if (capacity == 0) return CDR_OK;
    ctx->ev0 = new hipEvent_t[capacity];
    ctx->ev1 = new hipEvent_t[capacity];
    ctx->tags = new int[capacity];
    for (int i = 0; i < capacity; ++i) { CDR_HIP(hipEventCreate(&ctx->ev0[i])); CDR_HIP(hipEventCreate(&ctx->ev1[i])); }
    ctx->timing_cap = capacity;
    return CDR_OK;
}

extern "C" int cdr_timing_collect(cdr_ctx* ctx, int* tags, float* ms, int max_n, int* n_out) {
    CDR_CHECK_ARG(ctx && tags && ms && n_out && max_n >= 0);
    int n = ctx->timing_n < max_n ? ctx->timing_n : max_n;
    for (int i = 0; i < n; ++i) {
        CDR_HIP(hipEventSynchronize(ctx->ev1[i]));
        CDR_HIP(hipEventElapsedTime(&ms[i], ctx->ev0[i], ctx->ev1[i]));
        tags[i] = ctx->tags[i];
    }
    *n_out = n;
    ctx->timing_n = 0;
    return CDR_OK;
}

extern "C" int cdr_ctx_destroy(cdr_ctx* ctx) {
    if (!ctx) return CDR_OK;
    cdr_timing_enable(ctx, 0);
    if (ctx->partials) (void)hipFree(ctx->partials);
    if (ctx->tickets) (void)hipFree(ctx->tickets);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->conet_tab_dev) (void)hipFree(ctx->conet_tab_dev);
    if (ctx->conet_tab_pin) (void)hipHostFree(ctx->conet_tab_pin);
    free(ctx->conet_tab_shadow);
    delete ctx;
    return CDR_OK;
}

// The next forward launch made with this context (cdr_bpr_fwd / cdr_point_fwd / cdr_point_fwd_pair) also zero-fills [ptr, ptr + bytes):
// the dense gradient buffers of the step's backward, cleared under the forward's gathers instead of by a launch of their own.
extern "C" int cdr_ctx_scrub_next(cdr_ctx* ctx, void* ptr, size_t bytes) {
    CDR_CHECK_ARG(ctx && ptr && bytes > 0 && (bytes & 15) == 0 && ((uintptr_t)ptr & 15) == 0);
    ctx->scrub_ptr = ptr; ctx->scrub_bytes = bytes;
    return CDR_OK;
}

// The count path of the fused BPR step (cdr_bpr_step_fused / _dev, 16,448 <= B <= 131,072): per-row occurrence counters for the two tables
// the NEXT steps on this context train (uint32 [rows] each, ALL ZERO on entry and left all zero by every step) and the duplicate list's
// workspace (cdr_id_count_workspace_bytes).  NULL counters switch back to the sorted path.  The step checks the row counts against its own.
extern "C" int cdr_ctx_set_id_counters(cdr_ctx* ctx, uint32_t* user_counts, int64_t user_rows, uint32_t* item_counts, int64_t item_rows,
                                       void* list_ws, size_t list_ws_bytes) {
    CDR_CHECK_ARG(ctx != nullptr);
    if (!user_counts || !item_counts || !list_ws) {
        ctx->idc_user = ctx->idc_item = nullptr; ctx->idc_user_rows = ctx->idc_item_rows = 0; ctx->idc_list = nullptr; ctx->idc_list_bytes = 0;
        return CDR_OK;
    }
    CDR_CHECK_ARG(user_rows > 0 && item_rows > 0 && list_ws_bytes > 0);
    ctx->idc_user = user_counts; ctx->idc_user_rows = user_rows; ctx->idc_item = item_counts; ctx->idc_item_rows = item_rows;
    ctx->idc_list = list_ws; ctx->idc_list_bytes = list_ws_bytes;
    return CDR_OK;
}

// Context, error reporting, ABI version.
#include <stdarg.h>
#include <string.h>
#include "cdr_common.h"

static thread_local char g_err[512] = "";

void cdr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* cdr_last_error(void) { return g_err; }
extern "C" int cdr_abi_version(void) { return 1; }

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
    hipError_t e = hipMalloc(&c->partials, sizeof(double) * CDR_MAX_PARTIAL_BLOCKS * CDR_PARTIAL_STRIDE);
    hipSetDevice(prev);
    if (e != hipSuccess) {
        delete c;
        cdr_set_error("cdr_ctx_create: scratch allocation failed: %s", hipGetErrorString(e));
        return CDR_ENOMEM;
    }
    *out = c;
    return CDR_OK;
}

extern "C" int cdr_ctx_destroy(cdr_ctx* ctx) {
    if (!ctx) return CDR_OK;
    if (ctx->partials) hipFree(ctx->partials);
    delete ctx;
    return CDR_OK;
}

// ABI bookkeeping: version, thread-local error text, host-side geometry helpers.
#include <stdarg.h>
#include <string.h>

#include "common.hpp"

namespace binf {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int32_t fail(int32_t code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int32_t hip_fail(hipError_t e, const char *what)
{
    snprintf(g_err, sizeof(g_err), "%s: %s (hipError_t %d)", what,
             hipGetErrorString(e), (int)e);
    return (int32_t)e;
}

}  // namespace binf

extern "C" int32_t binf_abi_version(void) { return BINF_ABI_VERSION; }

extern "C" int32_t binf_last_error(char *buf, size_t n)
{
    size_t len = strlen(binf::g_err);
    if (buf && n) {
        size_t m = len < n - 1 ? len : n - 1;
        memcpy(buf, binf::g_err, m);
        buf[m] = 0;
    }
    return (int32_t)len;
}

extern "C" int32_t binf_pairwise_tree_height(int64_t n)
{
    if (n < 0) return binf::fail(BINF_E_ARG, "pairwise_tree_height: n < 0");
    return binf::pairwise_tree_height(n);
}

extern "C" int32_t binf_pairwise_leaf(int64_t n, int32_t H, int32_t path,
                                      int64_t *off, int64_t *len,
                                      int32_t *depth, int32_t *canonical)
{
    if (n < 0 || n > 0x7fffffff || H < 0 || H > 24 || path < 0 ||
        path >= (1 << H) || !off || !len || !depth || !canonical)
        return binf::fail(BINF_E_ARG, "pairwise_leaf: bad argument");
    binf::Leaf L = binf::pairwise_leaf((int32_t)n, H, path);
    *off = L.off;
    *len = L.len;
    *depth = L.depth;
    *canonical = L.canonical;
    return 0;
}

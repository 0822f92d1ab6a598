// Error reporting shared by every translation unit of libtrajopt_grpo_hip.so.
#include "tg_common.hpp"

namespace tg {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace tg

extern "C" {
const char* tg_last_error(void) { return tg::g_err; }
int tg_abi_version(void) { return TG_ABI_VERSION; }
}

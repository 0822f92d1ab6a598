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

int tg_clock_probe_attach(int32_t family, void* d_probe) {
    switch (family) {
        case TG_PROBE_FWD_CHAIN: return tg::attach_probe_fwd_chain(d_probe);
        case TG_PROBE_BWD_CHAIN: return tg::attach_probe_bwd_chain(d_probe);
        case TG_PROBE_WEIGHT_GRAD: return tg::attach_probe_weight_grad(d_probe);
        // (the fp32 learners: H = 64 / 128 and H = 256 are different kernels of one family; a net runs one of them)
        case TG_PROBE_F32_CHAIN: { const int rc = tg::attach_probe_f32(0, d_probe); return rc ? rc : tg::attach_probe_f32w(0, d_probe); }
        case TG_PROBE_F32_WEIGHT_GRAD: { const int rc = tg::attach_probe_f32(1, d_probe); return rc ? rc : tg::attach_probe_f32w(1, d_probe); }
        case TG_PROBE_MFMA_LOOP: return tg::attach_probe_mfma_loop(d_probe);
        case TG_PROBE_FWD_CHAIN_PLAIN: return tg::attach_probe_fwd_chain_plain(d_probe);
        default: return tg::set_error(TG_ERR_ARG, "tg_clock_probe_attach: family %d", family);
    }
}
}

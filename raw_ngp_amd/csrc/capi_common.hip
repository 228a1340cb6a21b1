// Error reporting and ABI version of libngp_hip.so.
#include "ngp_common.hpp"

namespace ngp {
char *error_buffer()
{
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace ngp

extern "C" int ngp_abi_version(void) { return NGP_ABI_VERSION; }

extern "C" const char *ngp_last_error(void) { return ngp::error_buffer(); }

// Error plumbing and version query of libnextou_hip.so.
#include "common.h"
#include <cstring>

namespace nextou {

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace nextou

extern "C" int nextou_abi_version(void) { return NEXTOU_ABI_VERSION; }
extern "C" const char* nextou_last_error(void) { return nextou::error_buffer(); }

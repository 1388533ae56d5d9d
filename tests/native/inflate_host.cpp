// Host build of the DEFLATE decoder the GPU kernel runs (mustache_amd/csrc/mfk_inflate_core.h, a wave of one lane), for the
// CPU tests: tests/test_inflate.py compares it with zlib on every kind of block.  Test infrastructure, not a product path.
#include <stdint.h>
#include <string.h>
#include "../../mustache_amd/csrc/mfk_inflate_core.h"

extern "C" int32_t mfk_inflate_host(const uint8_t *in, uint32_t clen, uint8_t *out, uint32_t ulen) {
    static thread_local InflTables T;
    memset(&T, 0xA5, sizeof T);      // nothing may depend on what an earlier block left behind
    return infl_block<1>(in, clen, out, ulen, T, 0);
}

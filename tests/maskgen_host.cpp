// tests/maskgen_host.cpp — the mask-only move generator of the library (cchess_zero_amd/csrc/cz_maskgen.h, one lane = one
// position on the GPU) compiled for the HOST, so that tests/test_maskgen_cpu.py can hold the very same function to the golden
// move lists of the reference and to the C oracle on the CPU.  Test infrastructure: nothing in the product path uses it.
#include <string.h>
#include "../cchess_zero_amd/csrc/cz_maskgen.h"

extern "C" void czm_host_tables(const int16_t *lut, CzmTables *t) { czm_build_tables(lut, t); }
extern "C" int czm_host_sizeof_tables(void) { return (int)sizeof(CzmTables); }
// boards [n][90], side [n] -> mask [n][66], count [n] (-1: error)
extern "C" void czm_host_masks(const CzmTables *t, const uint8_t *boards, const uint8_t *side, int n, uint32_t *mask, int *count) {
    for (int i = 0; i < n; ++i) {
        uint32_t w[23];
        unsigned char buf[92];
        memcpy(buf, boards + (size_t)i * 90, 90);
        buf[90] = buf[91] = 0;
        memcpy(w, buf, 92);
        uint32_t *row = mask + (size_t)i * 66;
        memset(row, 0, 66 * 4);
        int emits = 0;
        count[i] = czm_position(w, side[i] ? 1 : 0, *t, [row, &emits](int bit, uint32_t field) {
            ++emits;
            czm_or_field([row](int wi, uint32_t v) { if (wi < 66) row[wi] |= v; }, bit, field);
        });
        if (emits != CZM_EMITS) count[i] = -1000 - emits;   // the kernel's record buffer relies on this number
    }
}

// boards [n][90], side [n] -> moves [n][128] labels in the reference's order (0xFFFF padding), count [n] (-1: error), and the
// mask [n][66] czm_list emits beside the list (NULL: not wanted)
extern "C" void czm_host_lists(const CzmTables *t, const uint8_t *boards, const uint8_t *side, int n, uint16_t *moves, int *count, uint32_t *mask) {
    for (int i = 0; i < n; ++i) {
        uint32_t w[23];
        unsigned char buf[92];
        memcpy(buf, boards + (size_t)i * 90, 90);
        buf[90] = buf[91] = 0;
        memcpy(w, buf, 92);
        uint16_t *row = moves + (size_t)i * 128;
        for (int k = 0; k < 128; ++k) row[k] = 0xFFFF;
        uint32_t scratch[17], dummy[66];
        uint32_t *mrow = mask ? mask + (size_t)i * 66 : dummy;
        memset(mrow, 0, 66 * 4);
        int emits = 0;
        count[i] = czm_list(w, side[i] ? 1 : 0, *t, [row](int m, int label, uint32_t b) { const int k = CZM_IGNORE_SLOT + m / 2; if (b && (m & 1) == 0 && k >= 0 && k < 128) row[k] = (uint16_t)label; return m + 2 * (int)b; },
                            [&scratch](int k) -> uint32_t & { return scratch[k]; }, [] {},
                            [mrow, &emits](int bit, uint32_t field) { ++emits; czm_or_field([mrow](int wi, uint32_t v) { if (wi < 66) mrow[wi] |= v; }, bit, field); });
        if (emits != CZM_EMITS) count[i] = -1000 - emits;
    }
}

// cz_api.hip — the extern "C" boundary of libcchess_hip.so (declared in include/cchess_hip.h).
#include "cz_internal.h"
#include "cz_maskgen.h"

#include <stdarg.h>
#include <string.h>
#include <vector>

static thread_local char g_err[512] = "";

void cz_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {

size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// carve helper over one device allocation
struct Carver {
    char *base;
    size_t off = 0;
    template <typename T> T *take(size_t n) {
        T *p = base ? (T *)(base + off) : nullptr;
        off = align_up(off + n * sizeof(T));
        return p;
    }
};

void carve_pool(Carver &c, CzPool &p, size_t n) {
    p.P = c.take<float>(n); p.W = c.take<float>(n); p.Q = c.take<float>(n);
    p.N = c.take<int32_t>(n); p.parent = c.take<int32_t>(n); p.child_begin = c.take<int32_t>(n);
    p.child_count = c.take<uint16_t>(n); p.move = c.take<uint16_t>(n); p.sd = c.take<uint16_t>(n);
}

void carve_trees(Carver &c, CzTrees &t, size_t G, size_t words) {
    t.mark_bits = c.take<unsigned long long>(G * words);
    t.mark_rank = c.take<uint32_t>(G * words);
    t.root_board = c.take<uint8_t>(G * CZD_BOARD_LDS);
    t.rec = c.take<CzTreeRec>(G);
    t.pend_moves = c.take<uint16_t>(G * CZD_MAXMOVES);
    t.pend_path = c.take<int32_t>(G * CZ_PATH_MAX);
    t.pk_kind = c.take<int32_t>(G); t.pk_leaf = c.take<int32_t>(G); t.pk_value = c.take<float>(G);
    t.pk_side = c.take<uint8_t>(G); t.pk_nmoves = c.take<uint16_t>(G);
    t.slot_of = c.take<int32_t>(G);
    t.evcnt = c.take<int32_t>(2);
    t.adv_list = c.take<int32_t>(G); t.adv_cnt = c.take<int32_t>(4);
    t.evtotal = c.take<unsigned long long>(2);
}

}  // namespace

extern "C" {

const char *cz_last_error(void) { return g_err; }

// CRC-32C (Castagnoli, reflected 0x82F63B78), host code: the checksum of TensorFlow's checkpoint blocks and tensors
// (cchess_zero_amd/tf_checkpoint.py reads and writes the reference's tf.train.Saver files; pure Python does ~5 MB/s)
unsigned int cz_crc32c(const void *data, size_t n) {
    static unsigned int T[8][256];
    static bool init = false;
    if (!init) {
        for (unsigned int i = 0; i < 256; ++i) {
            unsigned int c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            T[0][i] = c;
        }
        for (unsigned int i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) T[s][i] = (T[s - 1][i] >> 8) ^ T[0][T[s - 1][i] & 0xFFu];
        init = true;
    }
    const unsigned char *p = (const unsigned char *)data;
    unsigned int c = 0xFFFFFFFFu;
    while (n >= 8) {   // slicing-by-8
        const unsigned int lo = c ^ ((unsigned int)p[0] | ((unsigned int)p[1] << 8) | ((unsigned int)p[2] << 16) | ((unsigned int)p[3] << 24));
        c = T[7][lo & 0xFFu] ^ T[6][(lo >> 8) & 0xFFu] ^ T[5][(lo >> 16) & 0xFFu] ^ T[4][lo >> 24] ^
            T[3][p[4]] ^ T[2][p[5]] ^ T[1][p[6]] ^ T[0][p[7]];
        p += 8; n -= 8;
    }
    while (n--) c = T[0][(c ^ *p++) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
int cz_version(void) { return 200; }

int cz_tables(const int16_t **lut, const int16_t **unflip, const char **labels, const uint16_t **srcdst) {
    const CzHostTables &t = cz_host_tables();
    if (lut) *lut = t.lut;
    if (unflip) *unflip = t.unflip;
    if (labels) *labels = t.labels;
    if (srcdst) *srcdst = t.srcdst;
    return CZ_OK;
}

int cz_zobrist(const uint64_t **keys, uint64_t *side_key) {
    const CzHostTables &t = cz_host_tables();
    if (keys) *keys = t.zob;
    if (side_key) *side_key = t.zob[15 * CZ_NSQ];
    return CZ_OK;
}

int cz_create(int device, int max_games, int max_nodes_per_tree, cz_ctx **out) {
    CZ_REQUIRE(out != nullptr, "cz_create: out is NULL");
    CZ_REQUIRE(max_games > 0 && max_nodes_per_tree >= 2, "cz_create: max_games > 0 and max_nodes_per_tree >= 2 required");
    int ndev = 0;
    CZ_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) { cz_set_error("cz_create: device %d not available (%d HIP devices)", device, ndev); return CZ_EINVAL; }
    CZ_HIP(hipSetDevice(device));
    cz_ctx *c = new cz_ctx();
    memset(c, 0, sizeof(*c));
    c->device = device; c->stream = nullptr; c->max_games = max_games; c->cap = max_nodes_per_tree; c->G = 0;
    c->width = 1;

    // tables
    const CzHostTables &ht = cz_host_tables();
    {
        Carver m{nullptr};
        m.take<int16_t>(CZ_NSQ * CZ_NSQ); m.take<int16_t>(CZ_NLABELS); m.take<uint16_t>(CZ_NLABELS); m.take<uint64_t>(15 * CZ_NSQ + 1);
        if (hipMalloc(&c->tab_block, m.off) != hipSuccess) { cz_set_error("cz_create: hipMalloc(tables) failed"); delete c; return CZ_ENOMEM; }
        Carver k{(char *)c->tab_block};
        int16_t *lut = k.take<int16_t>(CZ_NSQ * CZ_NSQ);
        int16_t *unf = k.take<int16_t>(CZ_NLABELS);
        uint16_t *sd = k.take<uint16_t>(CZ_NLABELS);
        uint64_t *zb = k.take<uint64_t>(15 * CZ_NSQ + 1);
        CZ_HIP(hipMemcpy(lut, ht.lut, sizeof(ht.lut), hipMemcpyHostToDevice));
        CZ_HIP(hipMemcpy(unf, ht.unflip, sizeof(ht.unflip), hipMemcpyHostToDevice));
        CZ_HIP(hipMemcpy(sd, ht.srcdst, sizeof(ht.srcdst), hipMemcpyHostToDevice));
        CZ_HIP(hipMemcpy(zb, ht.zob, sizeof(ht.zob), hipMemcpyHostToDevice));
        c->tab.lut = lut; c->tab.unflip = unf; c->tab.srcdst = sd; c->tab.zob = zb;
        CzmTables mt = {};   // the mask-only generator's per-square tables, derived from the label LUT
        czm_build_tables(ht.lut, &mt);
        CzmTables *dmt = nullptr;
        if (hipMalloc(&dmt, sizeof(CzmTables)) != hipSuccess) { cz_set_error("cz_create: hipMalloc(mask tables) failed"); delete c; return CZ_ENOMEM; }
        CZ_HIP(hipMemcpy(dmt, &mt, sizeof(mt), hipMemcpyHostToDevice));
        c->mask_tab = dmt;
    }
    // per-tree arrays
    {
        Carver m{nullptr};
        CzTrees dummy;
        const size_t words = ((size_t)max_nodes_per_tree + 63) / 64;
        carve_trees(m, dummy, (size_t)max_games, words);
        if (hipMalloc(&c->tree_block, m.off) != hipSuccess) { cz_set_error("cz_create: hipMalloc(trees, %zu B) failed", m.off); cz_destroy(c); return CZ_ENOMEM; }
        CZ_HIP(hipMemset(c->tree_block, 0, m.off));
        Carver k{(char *)c->tree_block};
        carve_trees(k, c->t, (size_t)max_games, words);
        c->t.cap = max_nodes_per_tree;
        c->t.words = (int)words;
    }
    // node pool: ONE pool of cap nodes per tree (cz_search_advance compacts the kept subtree in place)
    const size_t n = (size_t)max_games * (size_t)max_nodes_per_tree;
    {
        Carver m{nullptr};
        CzPool dummy;
        carve_pool(m, dummy, n);
        if (hipMalloc(&c->pool_block, m.off) != hipSuccess) {
            cz_set_error("cz_create: hipMalloc(node pool, %zu B) failed", m.off);
            cz_destroy(c);
            return CZ_ENOMEM;
        }
        Carver k{(char *)c->pool_block};
        carve_pool(k, c->t.pool, n);
    }
    *out = c;
    return CZ_OK;
}

void cz_destroy(cz_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->pend_block) (void)hipFree(c->pend_block);
    if (c->tab_block) (void)hipFree(c->tab_block);
    if (c->mask_tab) (void)hipFree(const_cast<CzmTables *>(c->mask_tab));
    if (c->xc_block) (void)hipFree(c->xc_block);
    if (c->tree_block) (void)hipFree(c->tree_block);
    if (c->pool_block) (void)hipFree(c->pool_block);
    if (c->sp_block) (void)hipFree(c->sp_block);
    if (c->ec_block) (void)hipFree(c->ec_block);
    if (c->mx_xbuf) (void)hipFree(c->mx_xbuf);
    delete c;
}

int cz_set_stream(cz_ctx *c, void *s) {
    CZ_REQUIRE(c, "null ctx");
    c->stream = (hipStream_t)s;
    return CZ_OK;
}

int cz_synchronize(cz_ctx *c) {
    CZ_REQUIRE(c, "null ctx");
    CZ_HIP(hipStreamSynchronize(c->stream));
    return CZ_OK;
}

int cz_malloc(cz_ctx *c, size_t bytes, void **dptr) {
    CZ_REQUIRE(c && dptr, "cz_malloc: null argument");
    CZ_HIP(hipSetDevice(c->device));
    if (hipMalloc(dptr, bytes ? bytes : 1) != hipSuccess) { cz_set_error("cz_malloc: %zu bytes failed", bytes); return CZ_ENOMEM; }
    return CZ_OK;
}
int cz_free(cz_ctx *c, void *dptr) {
    CZ_REQUIRE(c, "null ctx");
    CZ_HIP(hipFree(dptr));
    return CZ_OK;
}
int cz_upload(cz_ctx *c, void *dst, const void *src, size_t bytes) {
    CZ_REQUIRE(c, "null ctx");
    CZ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    CZ_HIP(hipStreamSynchronize(c->stream));
    return CZ_OK;
}
int cz_download(cz_ctx *c, void *dst, const void *src, size_t bytes) {
    CZ_REQUIRE(c, "null ctx");
    CZ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    CZ_HIP(hipStreamSynchronize(c->stream));
    return CZ_OK;
}

int cz_movegen(cz_ctx *c, const uint8_t *boards, const uint8_t *side, int G, uint16_t *moves, uint16_t *count, uint32_t *mask) {
    CZ_REQUIRE(c && G >= 0, "cz_movegen: null ctx / negative G");
    if (G == 0) return CZ_OK;
    CZ_REQUIRE(boards && side && count, "cz_movegen: boards, side, count required");
    return czk_movegen(c, boards, side, G, moves, count, mask, 0);
}

int cz_movegen_ex(cz_ctx *c, const uint8_t *boards, const uint8_t *side, int G, uint16_t *moves, uint16_t *count, uint32_t *mask, int flags) {
    CZ_REQUIRE(c && G >= 0, "cz_movegen_ex: null ctx / negative G");
    CZ_REQUIRE((flags & ~CZ_MOVES_NO_PAD) == 0, "cz_movegen_ex: unknown flag");
    if (G == 0) return CZ_OK;
    CZ_REQUIRE(boards && side && count, "cz_movegen_ex: boards, side, count required");
    return czk_movegen(c, boards, side, G, moves, count, mask, flags);
}
int cz_apply_move(cz_ctx *c, uint8_t *boards, uint8_t *side, const uint16_t *label, int G, uint64_t *hash, uint8_t *captured, int8_t *terminal) {
    CZ_REQUIRE(c && G >= 0, "cz_apply_move: null ctx / negative G");
    if (G == 0) return CZ_OK;
    CZ_REQUIRE(boards && side && label, "cz_apply_move: boards, side, move_label required");
    return czk_apply_move(c, boards, side, label, G, hash, captured, terminal);
}
int cz_hash(cz_ctx *c, const uint8_t *boards, const uint8_t *side, int G, uint64_t *hash) {
    CZ_REQUIRE(c && G >= 0, "cz_hash: null ctx / negative G");
    if (G == 0) return CZ_OK;
    CZ_REQUIRE(boards && side && hash, "cz_hash: null argument");
    return czk_hash(c, boards, side, G, hash);
}
int cz_encode_planes(cz_ctx *c, const uint8_t *boards, const uint8_t *side, int G, void *planes, int dtype, int channels, int quirk) {
    CZ_REQUIRE(c && G >= 0, "cz_encode_planes: null ctx / negative G");
    if (G == 0) return CZ_OK;
    CZ_REQUIRE(boards && side && planes, "cz_encode_planes: null argument");
    CZ_REQUIRE(dtype == CZ_F32 || dtype == CZ_BF16 || dtype == CZ_F16, "cz_encode_planes: dtype must be CZ_F32, CZ_BF16 or CZ_F16");
    CZ_REQUIRE(channels >= 14 && channels <= 64, "cz_encode_planes: 14 <= channels <= 64");
    return czk_encode_planes(c, boards, side, G, planes, dtype, channels, quirk);
}

int cz_search_set_width(cz_ctx *c, int width) {
    CZ_REQUIRE(c, "null ctx");
    if (width < 1 || width > 64) { cz_set_error("cz_search_set_width: width %d outside 1..64", width); return CZ_EINVAL; }
    if (width <= c->width) return CZ_OK;
    if (c->t.ec_key) { cz_set_error("cz_search_set_width: the evaluation cache needs width 1 (cz_search_set_eval_cache(ctx, 0) first)"); return CZ_EINVAL; }
    CZ_HIP(hipStreamSynchronize(c->stream));
    const size_t n = (size_t)c->max_games * (size_t)width;
    Carver m{nullptr};
    m.take<int32_t>(n); m.take<int32_t>(n); m.take<float>(n); m.take<uint8_t>(n); m.take<uint16_t>(n); m.take<uint16_t>(n * CZD_MAXMOVES);
    void *blk = nullptr;
    if (hipMalloc(&blk, m.off) != hipSuccess) { cz_set_error("cz_search_set_width: hipMalloc(%zu B) failed", m.off); return CZ_ENOMEM; }
    CZ_HIP(hipMemset(blk, 0, m.off));
    Carver k{(char *)blk};
    c->t.pk_kind = k.take<int32_t>(n); c->t.pk_leaf = k.take<int32_t>(n); c->t.pk_value = k.take<float>(n);
    c->t.pk_side = k.take<uint8_t>(n); c->t.pk_nmoves = k.take<uint16_t>(n); c->t.pend_moves = k.take<uint16_t>(n * CZD_MAXMOVES);
    if (c->pend_block) (void)hipFree(c->pend_block);
    c->pend_block = blk;
    c->width = width;
    return CZ_OK;
}

int cz_search_set_sim_target(cz_ctx *c, int target) {
    CZ_REQUIRE(c && target >= 0, "cz_search_set_sim_target: null ctx / negative target");
    c->sim_target = target;
    return CZ_OK;
}

int cz_search_set_eval_cache(cz_ctx *c, int on) {
    CZ_REQUIRE(c, "cz_search_set_eval_cache: null ctx");
    if (on && c->width != 1) { cz_set_error("cz_search_set_eval_cache: needs width 1 (one simulation in flight per tree)"); return CZ_EINVAL; }
    const size_t per = (size_t)c->max_games * CZ_EC_ENTRIES;
    if (on && !c->ec_block) {
        // per entry: key 8 B, node 4 B, value 4 B, the packed position 48 B; per tree: the pending leaf's packed position
        const size_t bytes = per * (8 + 4 + 4 + 48) + (size_t)c->max_games * 48;
        CZ_HIP(hipSetDevice(c->device));
        if (hipMalloc(&c->ec_block, bytes) != hipSuccess) { c->ec_block = nullptr; cz_set_error("cz_search_set_eval_cache: hipMalloc(%zu B) failed", bytes); return CZ_ENOMEM; }
    }
    if (on) {
        char *b = (char *)c->ec_block;
        c->t.ec_key = (unsigned long long *)b;
        c->t.ec_node = (int32_t *)(b + per * 8);
        c->t.ec_val = (float *)(b + per * 12);
        c->t.ec_board = (uint32_t *)(b + per * 16);
        c->t.pend_board = (uint32_t *)(b + per * 64);
        if (!c->t.ec_key_mask) c->t.ec_key_mask = ~0ull;
        // an empty cache: entries of an earlier use would point into trees that no longer exist
        CZ_HIP(hipMemsetAsync(c->ec_block, 0, per * 8, c->stream));
        czk_search_clear_cache_stats(c);
        if (c->t.xc_base) CZ_HIP(hipMemsetAsync(c->xc_block, 0, ((size_t)1 << c->xc_log2_entries) * 8 + 64, c->stream));   // and the cross-tree level
    } else {
        c->t.ec_key = nullptr; c->t.ec_node = nullptr; c->t.ec_val = nullptr; c->t.ec_board = nullptr; c->t.pend_board = nullptr;
        if (c->xc_block) return cz_search_set_xcache(c, 0);   // the cross-tree level lives behind the per-tree probe
    }
    return CZ_OK;
}
int cz_search_set_xcache(cz_ctx *c, int log2_entries) {
    CZ_REQUIRE(c && (log2_entries == 0 || (log2_entries >= 6 && log2_entries <= 24)), "cz_search_set_xcache: log2_entries must be 0 (off) or 6..24");
    CZ_HIP(hipSetDevice(c->device));
    if (log2_entries == 0 || (c->xc_block && c->xc_log2_entries != log2_entries)) {
        if (c->xc_block) { CZ_HIP(hipStreamSynchronize(c->stream)); (void)hipFree(c->xc_block); }
        c->xc_block = nullptr; c->xc_log2_entries = 0;
        c->t.xc_base = nullptr; c->t.xc_mask = 0;
        if (log2_entries == 0) return CZ_OK;
    }
    if (!c->t.ec_key) { cz_set_error("cz_search_set_xcache: switch the per-tree evaluation cache on first (cz_search_set_eval_cache(ctx, 1))"); return CZ_EINVAL; }
    const size_t n = (size_t)1 << log2_entries;
    // per entry: key 8, value 4, count 4, position 48, labels 256, (src, dst) 256, priors 512 = 1088 bytes; + 32 bytes of counters per tree
    if (!c->xc_block) {
        const size_t bytes = n * 1088 + 64 + (size_t)c->max_games * 32;
        if (hipMalloc(&c->xc_block, bytes) != hipSuccess) { c->xc_block = nullptr; cz_set_error("cz_search_set_xcache: hipMalloc(%zu B) failed", bytes); return CZ_ENOMEM; }
        c->xc_log2_entries = log2_entries;
        c->t.xc_base = (char *)c->xc_block;
        c->t.xc_mask = (uint32_t)(n / 64 - 1);
    }
    // an empty table (new weights => remembered evaluations are stale): only the keys and the counters need clearing
    CZ_HIP(hipMemsetAsync(c->xc_block, 0, n * 8 + 64, c->stream));
    CZ_HIP(hipMemsetAsync(czx_tree_stats(c->t), 0, (size_t)c->max_games * 32, c->stream));
    return CZ_OK;
}
static int xcache_stats5(cz_ctx *c, unsigned long long *stats, int n) {
    for (int k = 0; k < n; ++k) stats[k] = 0;
    if (!c->t.xc_base) return CZ_OK;
    std::vector<uint32_t> per((size_t)c->max_games * 8);     // per tree: hits, lookups, written, no room, replaced, 3 spare
    CZ_HIP(hipMemcpyAsync(per.data(), czx_tree_stats(c->t), per.size() * 4, hipMemcpyDeviceToHost, c->stream));
    CZ_HIP(hipStreamSynchronize(c->stream));
    for (size_t g = 0; g < (size_t)c->max_games; ++g)
        for (int k = 0; k < n; ++k) stats[k] += per[g * 8 + k];
    return CZ_OK;
}
int cz_search_xcache_stats(cz_ctx *c, unsigned long long *stats4) {
    CZ_REQUIRE(c && stats4, "cz_search_xcache_stats: null argument");
    return xcache_stats5(c, stats4, 4);
}
int cz_search_xcache_stats5(cz_ctx *c, unsigned long long *stats5) {
    CZ_REQUIRE(c && stats5, "cz_search_xcache_stats5: null argument");
    return xcache_stats5(c, stats5, 5);
}
int cz_search_debug_eval_cache_key_bits(cz_ctx *c, int bits) {
    CZ_REQUIRE(c && (bits == 64 || (bits >= 8 && bits <= 24)), "cz_search_debug_eval_cache_key_bits: bits must be 8..24 or 64");
    // narrowed keys keep the 7-bit bucket field (bits 17..23) plus the bits - 7 lowest bits: entries still spread over the buckets
    c->t.ec_key_mask = bits == 64 ? ~0ull : ((0x7Full << 17) | ((1ull << (bits - 7)) - 1ull));
    return CZ_OK;
}
int cz_search_debug_advance_in_global_memory(cz_ctx *c, int on) {
    CZ_REQUIRE(c, "cz_search_debug_advance_in_global_memory: null context");
    c->adv_force_global = on != 0;
    return CZ_OK;
}
int cz_search_eval_cache_collisions(cz_ctx *c, unsigned long long *collisions) {
    CZ_REQUIRE(c && collisions, "cz_search_eval_cache_collisions: null argument");
    *collisions = 0;
    if (c->G <= 0) return CZ_OK;
    std::vector<CzTreeRec> h((size_t)c->G);
    CZ_HIP(hipMemcpyAsync(h.data(), c->t.rec, h.size() * sizeof(CzTreeRec), hipMemcpyDeviceToHost, c->stream));
    CZ_HIP(hipStreamSynchronize(c->stream));
    for (int g = 0; g < c->G; ++g) *collisions += h[(size_t)g].ec_collisions;
    return CZ_OK;
}
int cz_search_eval_cache_stats(cz_ctx *c, unsigned long long *hits, unsigned long long *lookups) {
    CZ_REQUIRE(c && hits && lookups, "cz_search_eval_cache_stats: null argument");
    *hits = 0; *lookups = 0;
    if (c->G <= 0) return CZ_OK;
    std::vector<CzTreeRec> h((size_t)c->G);
    CZ_HIP(hipMemcpyAsync(h.data(), c->t.rec, h.size() * sizeof(CzTreeRec), hipMemcpyDeviceToHost, c->stream));
    CZ_HIP(hipStreamSynchronize(c->stream));
    for (int g = 0; g < c->G; ++g) { *hits += h[(size_t)g].ec_hits; *lookups += h[(size_t)g].ec_lookups; }
    return CZ_OK;
}
int cz_search_set_terminal_extra(cz_ctx *c, int n) {
    CZ_REQUIRE(c && n >= 0 && n <= 64, "cz_search_set_terminal_extra: 0 <= n <= 64 required");
    c->terminal_extra = n;
    return CZ_OK;
}

int cz_search_select_k(cz_ctx *c, int mode, int k, const uint8_t *active, void *planes, int dtype, int channels, uint8_t *needs_eval) {
    CZ_REQUIRE(c && c->G > 0, "cz_search_select_k: call cz_search_reset first");
    CZ_REQUIRE(mode == 0 || mode == 1, "cz_search_select_k: mode must be 0 or 1");
    CZ_REQUIRE(dtype == CZ_F32 || dtype == CZ_BF16 || dtype == CZ_F16, "cz_search_select_k: dtype must be CZ_F32, CZ_BF16 or CZ_F16");
    CZ_REQUIRE(channels >= 14 && channels <= 64, "cz_search_select_k: 14 <= channels <= 64");
    if (k < 1 || k > c->width) { cz_set_error("cz_search_select_k: k=%d outside 1..width=%d (cz_search_set_width)", k, c->width); return CZ_EINVAL; }
    return czk_search_select_k(c, mode, k, active, planes, dtype, channels, needs_eval);
}
int cz_search_expand_backup_k(cz_ctx *c, int k, const void *logits, const void *value, int dtype) {
    CZ_REQUIRE(c && c->G > 0 && logits && value, "cz_search_expand_backup_k: null argument / no search");
    CZ_REQUIRE(dtype == CZ_F32 || dtype == CZ_BF16, "cz_search_expand_backup_k: dtype must be CZ_F32 or CZ_BF16");
    if (k < 1 || k > c->width) { cz_set_error("cz_search_expand_backup_k: k=%d outside 1..width=%d", k, c->width); return CZ_EINVAL; }
    return czk_search_expand_backup_k(c, k, logits, value, dtype);
}

int cz_search_reset(cz_ctx *c, const uint8_t *boards, const uint8_t *side, const int32_t *rr, int G) {
    CZ_REQUIRE(c && boards && side, "cz_search_reset: null argument");
    if (G <= 0 || G > c->max_games) { cz_set_error("cz_search_reset: G=%d outside 1..%d", G, c->max_games); return CZ_EINVAL; }
    c->G = G;
    return czk_search_reset(c, boards, side, rr, G, nullptr);
}
int cz_search_reload(cz_ctx *c, const uint8_t *which, const uint8_t *boards, const uint8_t *side, const int32_t *rr) {
    CZ_REQUIRE(c && c->G > 0, "cz_search_reload: call cz_search_reset first");
    CZ_REQUIRE(which && boards && side, "cz_search_reload: null argument");
    return czk_search_reset(c, boards, side, rr, c->G, which);
}
int cz_search_select(cz_ctx *c, int mode, const uint8_t *active, void *planes, int dtype, int channels, uint8_t *needs_eval) {
    CZ_REQUIRE(c && c->G > 0, "cz_search_select: call cz_search_reset first");
    CZ_REQUIRE(mode == 0 || mode == 1, "cz_search_select: mode must be 0 or 1");
    CZ_REQUIRE(dtype == CZ_F32 || dtype == CZ_BF16 || dtype == CZ_F16, "cz_search_select: dtype must be CZ_F32, CZ_BF16 or CZ_F16");
    CZ_REQUIRE(channels >= 14 && channels <= 64, "cz_search_select: 14 <= channels <= 64");
    return czk_search_select(c, mode, active, planes, dtype, channels, needs_eval);
}
int cz_search_expand_backup_fc(cz_ctx *c, const float *z, const float *value, const float *pfc_w, const float *pfc_b, int compact) {
    CZ_REQUIRE(c && c->G > 0 && z && value && pfc_w && pfc_b, "cz_search_expand_backup_fc: null argument / no search");
    return czk_search_expand_backup_fc(c, z, value, pfc_w, pfc_b, compact != 0);
}
int cz_search_select_compact(cz_ctx *c, int mode, const uint8_t *active, void *planes, int dtype, int channels,
                             const int32_t **slot_of, const int32_t **n_rows) {
    CZ_REQUIRE(c && c->G > 0 && planes, "cz_search_select_compact: call cz_search_reset first / null planes");
    CZ_REQUIRE(mode == 0 || mode == 1, "cz_search_select_compact: mode must be 0 or 1");
    CZ_REQUIRE(dtype == CZ_F32 || dtype == CZ_BF16 || dtype == CZ_F16, "cz_search_select_compact: dtype must be CZ_F32, CZ_BF16 or CZ_F16");
    CZ_REQUIRE(channels >= 14 && channels <= 64, "cz_search_select_compact: 14 <= channels <= 64");
    c->step_parity ^= 1;
    if (slot_of) *slot_of = c->t.slot_of;
    if (n_rows) *n_rows = c->t.evcnt + c->step_parity;
    return czk_search_select(c, mode, active, planes, dtype, channels, nullptr, true);
}
int cz_set_batch_count(cz_ctx *c, const int32_t *n_rows_dev) {
    CZ_REQUIRE(c, "null ctx");
    c->batch_count = n_rows_dev;
    return CZ_OK;
}
int cz_search_eval_totals(cz_ctx *c, unsigned long long *rows, unsigned long long *steps) {
    CZ_REQUIRE(c, "null ctx");
    unsigned long long h[2] = {0, 0};
    CZ_HIP(hipStreamSynchronize(c->stream));
    CZ_HIP(hipMemcpy(h, c->t.evtotal, sizeof(h), hipMemcpyDeviceToHost));
    if (rows) *rows = h[0];
    if (steps) *steps = h[1];
    return CZ_OK;
}
int cz_search_expand_backup(cz_ctx *c, const void *logits, const void *value, int dtype) {
    CZ_REQUIRE(c && c->G > 0 && logits && value, "cz_search_expand_backup: null argument / no search");
    CZ_REQUIRE(dtype == CZ_F32 || dtype == CZ_BF16, "cz_search_expand_backup: dtype must be CZ_F32 or CZ_BF16");
    return czk_search_expand_backup(c, logits, value, dtype);
}
int cz_search_root_stats(cz_ctx *c, uint16_t *label, int32_t *N, float *Q, float *P, float *W, uint16_t *count) {
    CZ_REQUIRE(c && c->G > 0, "cz_search_root_stats: no search");
    return czk_search_root_stats(c, label, N, Q, P, W, count);
}
int cz_search_advance(cz_ctx *c, const uint16_t *played) {
    CZ_REQUIRE(c && c->G > 0 && played, "cz_search_advance: null argument / no search");
    return czk_search_advance(c, played);
}
int cz_search_pick_ready(cz_ctx *c, int32_t *sim_threshold, int next_threshold, uint16_t *played, uint8_t *ready, unsigned long long *banked_sims) {
    CZ_REQUIRE(c && c->G > 0 && sim_threshold && played, "cz_search_pick_ready: null argument / no search");
    return czk_search_pick_ready(c, sim_threshold, next_threshold, played, ready, banked_sims);
}
int cz_search_reload_finished(cz_ctx *c, const uint8_t *ready, const uint16_t *played, const uint8_t *boards, const uint8_t *side,
                              const int32_t *rr, unsigned long long *reloaded) {
    CZ_REQUIRE(c && c->G > 0 && ready && played && boards && side, "cz_search_reload_finished: null argument / no search");
    return czk_search_reload_finished(c, ready, played, boards, side, rr, reloaded);
}
int cz_search_status(cz_ctx *c, int32_t *status, int32_t *nodes, int32_t *sims, int32_t *depth) {
    CZ_REQUIRE(c && c->G > 0, "cz_search_status: no search");
    return czk_search_status(c, status, nodes, sims, depth);
}

int cz_selfplay_begin(cz_ctx *c, int max_plies, const uint8_t *boards, const uint8_t *side, const int32_t *rr) {
    CZ_REQUIRE(c && c->G > 0, "cz_selfplay_begin: call cz_search_reset first");
    CZ_REQUIRE(max_plies >= 1 && max_plies <= 65535, "cz_selfplay_begin: 1 <= max_plies <= 65535");
    CZ_REQUIRE(boards == nullptr || side != nullptr, "cz_selfplay_begin: start_side required with start_boards");
    if (!c->sp_block || c->sp.max_plies != max_plies) {
        CZ_HIP(hipStreamSynchronize(c->stream));
        if (c->sp_block) { (void)hipFree(c->sp_block); c->sp_block = nullptr; }
        const size_t G = (size_t)c->max_games;
        auto carve = [&](Carver &k, CzSelfplay &sp) {
            sp.hist = k.take<uint8_t>(G * (size_t)max_plies * CZ_REC_BYTES);
            sp.ply = k.take<int32_t>(G);
            sp.stalled = k.take<uint8_t>(G);
            sp.active = k.take<uint8_t>(G);
            sp.start_board = k.take<uint8_t>(G * CZD_BOARD_LDS);
            sp.start_side = k.take<uint8_t>(G);
            sp.start_rr = k.take<int32_t>(G);
            sp.stats = k.take<long long>(CZ_SP_NSTATS);
        };
        Carver m{nullptr};
        CzSelfplay dummy;
        carve(m, dummy);
        if (hipMalloc(&c->sp_block, m.off) != hipSuccess) { c->sp_block = nullptr; cz_set_error("cz_selfplay_begin: hipMalloc(%zu B) failed", m.off); return CZ_ENOMEM; }
        Carver k{(char *)c->sp_block};
        carve(k, c->sp);
        c->sp.max_plies = max_plies;
    }
    return czk_selfplay_seed(c, boards, side, rr);
}
int cz_selfplay_active(cz_ctx *c, const uint8_t **active) {
    CZ_REQUIRE(c && c->sp_block && active, "cz_selfplay_active: call cz_selfplay_begin first");
    *active = c->sp.active;
    return CZ_OK;
}
int cz_selfplay_choose(cz_ctx *c, const float *gamma, const float *u, const uint16_t *forced, double temperature, float eps, int min_sims, uint16_t *played) {
    CZ_REQUIRE(c && c->sp_block && c->G > 0, "cz_selfplay_choose: call cz_selfplay_begin first");
    CZ_REQUIRE(u && played, "cz_selfplay_choose: u and played required");
    CZ_REQUIRE(temperature > 0.0 && eps >= 0.f && eps <= 1.f, "cz_selfplay_choose: temperature > 0 and 0 <= noise_eps <= 1 required");
    CZ_REQUIRE(min_sims >= 0, "cz_selfplay_choose: min_sims >= 0 required");
    return czk_selfplay_choose(c, gamma, u, forced, temperature, eps, min_sims, played);
}
int cz_selfplay_adjudicate(cz_ctx *c, int reseed, const uint16_t *played, int32_t *fin_n) {
    CZ_REQUIRE(c && c->sp_block && c->G > 0 && fin_n, "cz_selfplay_adjudicate: call cz_selfplay_begin first / null fin_n");
    return czk_selfplay_adjudicate(c, reseed, played, fin_n);
}
int cz_selfplay_flush(cz_ctx *c, const int32_t *fin_n, const long long *offset, uint8_t *ring, long long ring_records, const long long *read_cursor) {
    CZ_REQUIRE(c && c->sp_block && c->G > 0, "cz_selfplay_flush: call cz_selfplay_begin first");
    CZ_REQUIRE(fin_n && offset && ring && ring_records > 0, "cz_selfplay_flush: null argument / empty ring");
    return czk_selfplay_flush(c, fin_n, offset, ring, ring_records, read_cursor);
}
int cz_selfplay_stats(cz_ctx *c, long long *stats_dev) {
    CZ_REQUIRE(c && c->sp_block && stats_dev, "cz_selfplay_stats: call cz_selfplay_begin first / null argument");
    CZ_HIP(hipMemcpyAsync(stats_dev, c->sp.stats, sizeof(long long) * CZ_SP_NSTATS, hipMemcpyDeviceToDevice, c->stream));
    return CZ_OK;
}

int cz_search_tree_dump(cz_ctx *c, int g, int32_t *out, int max_records) {
    CZ_REQUIRE(c && c->G > 0 && g >= 0 && g < c->G, "cz_search_tree_dump: bad tree index");
    CZ_HIP(hipStreamSynchronize(c->stream));
    int32_t root = 0, n = 0;
    CZ_HIP(hipMemcpy(&root, &c->t.root_node[g], 4, hipMemcpyDeviceToHost));
    CZ_HIP(hipMemcpy(&n, &c->t.n_nodes[g], 4, hipMemcpyDeviceToHost));
    const CzPool &p = c->t.pool;
    const size_t base = (size_t)g * (size_t)c->cap;
    std::vector<float> P(n), W(n), Q(n);
    std::vector<int32_t> N(n), cb(n);
    std::vector<uint16_t> cc(n), mv(n);
    CZ_HIP(hipMemcpy(P.data(), p.P + base, 4 * (size_t)n, hipMemcpyDeviceToHost));
    CZ_HIP(hipMemcpy(W.data(), p.W + base, 4 * (size_t)n, hipMemcpyDeviceToHost));
    CZ_HIP(hipMemcpy(Q.data(), p.Q + base, 4 * (size_t)n, hipMemcpyDeviceToHost));
    CZ_HIP(hipMemcpy(N.data(), p.N + base, 4 * (size_t)n, hipMemcpyDeviceToHost));
    CZ_HIP(hipMemcpy(cb.data(), p.child_begin + base, 4 * (size_t)n, hipMemcpyDeviceToHost));
    CZ_HIP(hipMemcpy(cc.data(), p.child_count + base, 2 * (size_t)n, hipMemcpyDeviceToHost));
    CZ_HIP(hipMemcpy(mv.data(), p.move + base, 2 * (size_t)n, hipMemcpyDeviceToHost));
    // iterative pre-order
    struct Frame { int node, next; int depth; };
    std::vector<Frame> st;
    int nrec = 0;
    if (cb[root] >= 0) st.push_back({root, 0, 0});
    auto bits = [](float f) { int32_t i; memcpy(&i, &f, 4); return i; };
    while (!st.empty()) {
        Frame &f = st.back();
        if (f.next >= cc[f.node]) { st.pop_back(); continue; }
        const int ch = cb[f.node] + f.next++;
        const int depth = f.depth;
        if (nrec < max_records && out) {
            int32_t *r = out + (size_t)nrec * 7;
            r[0] = depth; r[1] = mv[ch]; r[2] = N[ch]; r[3] = bits(W[ch]); r[4] = bits(Q[ch]); r[5] = bits(P[ch]);
            r[6] = cb[ch] < 0 ? -1 : cc[ch];
        }
        ++nrec;
        if (cb[ch] >= 0) st.push_back({ch, 0, depth + 1});
    }
    return nrec;
}

}  // extern "C"

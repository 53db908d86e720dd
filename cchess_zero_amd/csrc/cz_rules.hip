// cz_rules.hip — stand-alone rules kernels: K1 move generation (one lane = one position: the ordered list k_movegen_list, the set
// k_movegen_mask), K2 (make move + hash + flags), the Zobrist key (one lane = one position), K3 (input planes: one wave per
// position).  Boards are staged in LDS, results leave with coalesced stores.  These are HBM / issue-bound byte kernels: no MFMA
// here by design.
#include "cz_internal.h"
#include "cz_maskgen.h"

namespace {

// K1m: the legal-move MASK (and count) without the ordered list — cz_movegen(moves = NULL).  One lane = one position; the rules
// are cz_maskgen.h's czm_position (register bit sets, no list, no LUT, no divergence on the piece kind).  A wave stages its 64
// boards (5 760 contiguous bytes) into LDS with coalesced 16-byte loads, every lane pulls its own 90 bytes out as 23 dwords
// (an odd lane's board starts on a 2-byte boundary: funnel shift).  czm_position's 15 (bit, field) results per position go to
// a record buffer in LDS (15 x 64 dwords, where the boards were) and from there to registers, transposed: lane l holds record
// l + 64 k of each half-wave.  The mask rows are then built and written HALF A WAVE AT A TIME: 32 rows of 66 words in the ABI's
// layout (8 448 bytes, the same LDS again), all 64 lanes applying the half's 480 records with ds_or_b32, then one contiguous
// block of 16-byte stores.  9.4 KB of LDS per wave instead of 17.9 for 64 rows: 12 waves per CU (3 per SIMD, the VGPR limit)
// instead of 8 — LDS is handed out in 1 280-byte granules on this chip: a first version with the records in their own 3.8 KB
// (13.3 KB per wave) got 11 waves per CU, not 12, and ran its 3 072 waves in two rounds.
// Algorithmic bytes: 90 + 1 in, 264 + 2 out per position (SURVEY 8(d) counts 312 with the board packed to 48 bytes).
#define CZK_HALF_WORDS (32 * CZ_MASK_WORDS)
// The workgroup IS one wave: its LDS instructions execute in issue order, so between the phases below the LDS counter has to
// drain and the compiler must not move memory operations across — but nothing needs the VECTOR-memory counter at zero: a
// __syncthreads() here waits for the prefetch just issued and for the previous rows' global stores (two exposed HBM round trips
// per group).
#define CZK_WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
__global__ __launch_bounds__(64) void k_movegen_mask(const CzmTables *__restrict__ gtab, const uint8_t *__restrict__ boards,
                                                     const uint8_t *__restrict__ side, int G, uint16_t *__restrict__ count,
                                                     uint32_t *__restrict__ mask) {
    __shared__ __attribute__((aligned(16))) uint32_t rows[CZK_HALF_WORDS];   // the 64 boards (1 440 words), then the records, then 32 mask rows at a time
    uint32_t *const rec = rows;                                               // [emit][lane]: field << 12 | bit (960 words) while czm_position runs
    __shared__ __attribute__((aligned(16))) CzmTables T;
    const int lane = threadIdx.x;
    if (lane < (int)(sizeof(CzmTables) / 16)) reinterpret_cast<uint4 *>(&T)[lane] = reinterpret_cast<const uint4 *>(gtab)[lane];   // hipMalloc'ed: 256-byte aligned
    const int ngroups = (G + 63) >> 6;
    const bool al16 = (reinterpret_cast<uintptr_t>(boards) & 15u) == 0, mal16 = mask && (reinterpret_cast<uintptr_t>(mask) & 15u) == 0;
    // Persistent waves (the launch has at most a chip's worth): a wave walks its groups with a stride and requests the NEXT
    // group's 5 760 board bytes (six 16-byte loads per lane) and side bytes into registers before it computes the current one,
    // so that the only HBM round trip a wave waits for is its first (SQ counters of the one-group-per-wave kernel: half of a
    // wave's life in s_waitcnt).  The prefetch needs 16-byte aligned boards (g0 * 90 is a multiple of 16); other addresses
    // take the byte path without it.
    uint4 pre[6];
    int presd = 0;
    auto prefetch = [&](int grp) {
        const int g0 = grp * 64, np = min(64, G - g0), nbytes = np * CZ_NSQ;
        const uint8_t *src = boards + (size_t)g0 * CZ_NSQ;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int i = lane + 64 * k;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (i * 16 + 16 <= nbytes) v = reinterpret_cast<const uint4 *>(src)[i];   // the ragged piece of a batch's last group: below
            pre[k] = v;
        }
        presd = (lane < np && side[g0 + lane]) ? 1 : 0;
    };
    if (al16 && (int)blockIdx.x < ngroups) prefetch(blockIdx.x);
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int g0 = grp * 64, np = min(64, G - g0), p = g0 + lane;
        const bool live = lane < np;
        CZK_WAVE_SYNC();   // the previous group's rows have left (and the tables are in place)
        int sd;
        if (al16) {   // the prefetched bytes -> LDS
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (lane + 64 * k < 64 * CZ_NSQ / 16) reinterpret_cast<uint4 *>(rows)[lane + 64 * k] = pre[k];
            if (np < 64) {   // the last group of a batch: its ragged 16-byte piece byte by byte, never past the batch (wave-uniform branch)
                const int nbytes = np * CZ_NSQ, full = nbytes & ~15;
                if (lane < nbytes - full) reinterpret_cast<uint8_t *>(rows)[full + lane] = boards[(size_t)g0 * CZ_NSQ + full + lane];
            }
            sd = presd;
        } else {
            const uint8_t *src = boards + (size_t)g0 * CZ_NSQ;
            uint8_t *dst = reinterpret_cast<uint8_t *>(rows);
            const int nbytes = np * CZ_NSQ;
            for (int i = lane; i < nbytes; i += 64) dst[i] = src[i];
            sd = (live && side[p]) ? 1 : 0;
        }
        CZK_WAVE_SYNC();
        uint32_t w[23];
        {   // the lane's 90 bytes start at byte 90 * lane: 4-aligned for even lanes, 2 (mod 4) for odd ones
            const int b0 = (CZ_NSQ * lane) >> 2, sh = (lane & 1) * 16;
            uint32_t d[24];
#pragma unroll
            for (int k = 0; k < 24; ++k) d[k] = rows[b0 + k];
#pragma unroll
            for (int k = 0; k < 23; ++k) w[k] = __builtin_amdgcn_alignbit(d[k + 1], d[k], (uint32_t)sh);
            w[22] &= 0x0000FFFFu;
            if (!live) {
#pragma unroll
                for (int k = 0; k < 23; ++k) w[k] = 0u;
            }
        }
        if (al16 && grp + (int)gridDim.x < ngroups) prefetch(grp + gridDim.x);   // in flight while this group is computed
        int ne = 0;   // wave-uniform: every lane emits the same CZM_EMITS records in the same order
        const int n = czm_position(w, sd, T, [&](int bit, uint32_t f) { rec[ne * 64 + lane] = (f << 12) | (uint32_t)bit; ++ne; });
        if (live) count[p] = n < 0 ? (uint16_t)0xFFFF : (uint16_t)n;
        if (!mask) continue;
        // the records leave LDS for registers (the rows take their place): lane l applies record r = l + 64 k of each half,
        // i.e. emit r >> 5 of the half's position r & 31
        CZK_WAVE_SYNC();
        uint32_t rv[2][8];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = lane + 64 * k;
                rv[h][k] = r < 32 * CZM_EMITS ? rec[(r >> 5) * 64 + h * 32 + (r & 31)] : 0u;
            }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h * 32 >= np) break;   // a ragged last group may have no second half (wave-uniform)
            CZK_WAVE_SYNC();   // the records are in registers (h = 0) / the first half's rows have left
#pragma unroll
            for (int k = 0; k < 9; ++k)
                if (lane + 64 * k < CZK_HALF_WORDS / 4) reinterpret_cast<uint4 *>(rows)[lane + 64 * k] = make_uint4(0, 0, 0, 0);
            CZK_WAVE_SYNC();
#pragma unroll
            for (int k = 0; k < 8; ++k) {   // the half's 32 x 15 = 480 records, one per lane and step
                const int r = lane + 64 * k;
                if (r < 32 * CZM_EMITS) {
                    const uint32_t v = rv[h][k];
                    uint32_t *row = rows + (r & 31) * CZ_MASK_WORDS;
                    czm_or_field([row](int wi, uint32_t x) { atomicOr(&row[wi], x); }, (int)(v & 0xFFFu), v >> 12);   // ds_or_b32, nothing returned
                }
            }
            CZK_WAVE_SYNC();
            const int nph = min(32, np - h * 32);
            uint32_t *dstm = mask + (size_t)(g0 + h * 32) * CZ_MASK_WORDS;
            if (mal16 && nph == 32) {   // 528 16-byte stores, statically counted (the waits on the prefetch stay counted too)
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    if (lane + 64 * k < CZK_HALF_WORDS / 4) reinterpret_cast<uint4 *>(dstm)[lane + 64 * k] = reinterpret_cast<const uint4 *>(rows)[lane + 64 * k];
            } else if (mal16) {
                for (int i = lane; i < nph * CZ_MASK_WORDS / 4; i += 64) reinterpret_cast<uint4 *>(dstm)[i] = reinterpret_cast<const uint4 *>(rows)[i];
                for (int i = (nph * CZ_MASK_WORDS / 4) * 4 + lane; i < nph * CZ_MASK_WORDS; i += 64) dstm[i] = rows[i];
            } else {
                for (int i = lane; i < nph * CZ_MASK_WORDS; i += 64) dstm[i] = rows[i];
            }
        }
    }
}

// K2: thread per game (the work per game is a handful of byte moves)
__global__ void k_apply_move(CzTables tab, uint8_t *__restrict__ boards, uint8_t *__restrict__ side,
                             const uint16_t *__restrict__ label, int G, uint64_t *__restrict__ hash,
                             uint8_t *__restrict__ captured, int8_t *__restrict__ terminal) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const uint16_t l = label[g];
    uint8_t *b = boards + (size_t)g * CZ_NSQ;
    uint8_t cap = 0;
    if (l < CZ_NLABELS) {
        const int src = tab.srcdst[l] & 0xFF, dst = tab.srcdst[l] >> 8;
        const uint8_t pc = b[src];
        cap = b[dst];
        b[dst] = pc;  // sim_do_action, main.py:671-672
        b[src] = 0;
        if (hash) {
            uint64_t h = hash[g];
            if (pc) h ^= tab.zob[pc * CZ_NSQ + src] ^ tab.zob[pc * CZ_NSQ + dst];
            if (cap) h ^= tab.zob[cap * CZ_NSQ + dst];
            h ^= tab.zob[15 * CZ_NSQ];
            hash[g] = h;
        }
        side[g] = side[g] ? 0 : 1;
    }
    if (captured) captured[g] = cap;  // is_kill_move != 0, main.py:219-227
    if (terminal) {
        bool K = false, k = false;
        for (int i = 0; i < CZ_NSQ; ++i) { K |= (b[i] == 1); k |= (b[i] == 8); }
        terminal[g] = (int8_t)((K ? 0 : 1) | (k ? 0 : 2));  // find('K') == -1 / find('k') == -1, main.py:409-413
    }
}

// K1 (ordered): GameBoard.get_legal_moves' list as labels in the reference's order — cz_movegen(moves != NULL).  One lane = one
// position, czm_list (cz_maskgen.h): the pieces kind by kind into payload registers, their counts summed in square order through
// 16 words of per-position scratch, then every piece writes its labels at its offset into the lane's row of the list buffer in
// LDS (64 rows of 128 labels; 260-byte stride: the 64 lanes' 2-byte writes spread over the banks) — the boards, the scratch and
// the list rows share the same LDS; the rows leave as 16-byte stores.  The lane = piece kernel this replaces (four positions per
// wave, staging rows, a segmented prefix sum, one LUT round trip and one LDS atomic per move) ran at 1.9 G positions/s.
#define CZK_LROW 65   /* dwords per list row in LDS: 64 + 1 */
// With mask != NULL the same launch writes the 2086-bit masks too: czm_list hands out the set's 15 (bit, field) pairs beside the
// list (registers), and once the list rows have left the same LDS holds the 64 mask rows (k_movegen_mask builds 32 at a time to
// stay at 9.4 KB; here the list rows have set the footprint already): each lane applies its own position's pairs to its own row.
// PAD = false (cz_movegen_ex, CZ_MOVES_NO_PAD; round 6): a row is written up to its count only (in 16-byte pieces: the labels behind
// `count` in the last piece are undefined) — the 0xFFFF padding is 2/3 of the list's 256 bytes (~40 moves per position), 65 LDS
// stores per lane to make and 1.97x the kernel's algorithmic HBM traffic to write.
template <bool MASK, bool PAD>
__global__ __launch_bounds__(64, MASK ? 2 : 3) void k_movegen_list(const CzmTables *__restrict__ gtab, const uint8_t *__restrict__ boards,
                                                     const uint8_t *__restrict__ side, int G, uint16_t *__restrict__ moves,
                                                     uint16_t *__restrict__ count, uint32_t *__restrict__ mask) {
    __shared__ __attribute__((aligned(16))) uint32_t rows[64 * CZ_MASK_WORDS];   // >= 64 * CZK_LROW + 4
    __shared__ __attribute__((aligned(16))) CzmTables T;
    const int lane = threadIdx.x;
    if (lane < (int)(sizeof(CzmTables) / 16)) reinterpret_cast<uint4 *>(&T)[lane] = reinterpret_cast<const uint4 *>(gtab)[lane];
    const int ngroups = (G + 63) >> 6;
    const bool al16 = (reinterpret_cast<uintptr_t>(boards) & 15u) == 0;
    uint4 pre[6];
    int presd = 0;
    auto prefetch = [&](int grp) {
        const int g0 = grp * 64, np = min(64, G - g0), nbytes = np * CZ_NSQ;
        const uint8_t *src = boards + (size_t)g0 * CZ_NSQ;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int i = lane + 64 * k;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (i * 16 + 16 <= nbytes) v = reinterpret_cast<const uint4 *>(src)[i];
            pre[k] = v;
        }
        presd = (lane < np && side[g0 + lane]) ? 1 : 0;
    };
    if (al16 && (int)blockIdx.x < ngroups) prefetch(blockIdx.x);
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int g0 = grp * 64, np = min(64, G - g0), p = g0 + lane;
        const bool live = lane < np;
        CZK_WAVE_SYNC();   // the previous group's rows have left (and the tables are in place)
        int sd;
        if (al16) {
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (lane + 64 * k < 64 * CZ_NSQ / 16) reinterpret_cast<uint4 *>(rows)[lane + 64 * k] = pre[k];
            if (np < 64) {
                const int nbytes = np * CZ_NSQ, full = nbytes & ~15;
                if (lane < nbytes - full) reinterpret_cast<uint8_t *>(rows)[full + lane] = boards[(size_t)g0 * CZ_NSQ + full + lane];
            }
            sd = presd;
        } else {
            const uint8_t *src = boards + (size_t)g0 * CZ_NSQ;
            uint8_t *dst = reinterpret_cast<uint8_t *>(rows);
            const int nbytes = np * CZ_NSQ;
            for (int i = lane; i < nbytes; i += 64) dst[i] = src[i];
            sd = (live && side[p]) ? 1 : 0;
        }
        CZK_WAVE_SYNC();
        uint32_t w[23];
        {
            const int b0 = (CZ_NSQ * lane) >> 2, sh = (lane & 1) * 16;
            uint32_t d[24];
#pragma unroll
            for (int k = 0; k < 24; ++k) d[k] = rows[b0 + k];
#pragma unroll
            for (int k = 0; k < 23; ++k) w[k] = __builtin_amdgcn_alignbit(d[k + 1], d[k], (uint32_t)sh);
            w[22] &= 0x0000FFFFu;
            if (!live) {
#pragma unroll
                for (int k = 0; k < 23; ++k) w[k] = 0u;
            }
        }
        if (al16 && grp + (int)gridDim.x < ngroups) prefetch(grp + gridDim.x);
        CZK_WAVE_SYNC();   // every lane holds its board: the bytes become scratch, then list rows
        const uint32_t slot128 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)rows + 2u * (lane * (2 * CZK_LROW) + CZM_IGNORE_SLOT);   // LDS byte address
        uint32_t recs[CZM_EMITS];
        int ne = 0;   // compile-time after unrolling: the emits sit in straight-line code
        const int n = czm_list(w, sd, T,
            // what is not a move goes to the row's padding word (u16 slot 128 of the 130) — byte offset b * m from it, one v_mad_i32_i24:
            // an unconditional store is cheaper than a store under an exec mask, the multiply-add cheaper than compare + select
            // (120 candidates per position)
            [slot128](int m, int label, uint32_t b) {
                typedef __attribute__((address_space(3))) uint16_t lds_u16;
                uint32_t at;   // (__mul24 leaves its 24-bit sign extension of m in the code: 2 more instructions per candidate)
                asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(at) : "v"(b), "v"(m), "v"(slot128));
                *reinterpret_cast<lds_u16 *>(at) = (uint16_t)label;
                asm("v_lshl_add_u32 %0, %1, 1, %0" : "+v"(m) : "v"(b));
                return m;
            },
            [&](int i) -> uint32_t & { return rows[i * 64 + lane]; },   // 0 <= i <= 16
            [&]() {   // the scratch has been read: fill the rows with the 0xFFFF padding of the ABI
                CZK_WAVE_SYNC();
                if (PAD)
                    for (int i = lane; i < 64 * CZK_LROW; i += 64) rows[i] = 0xFFFFFFFFu;
                CZK_WAVE_SYNC();
            },
            [&](int bit, uint32_t f) { if (MASK) recs[ne] = (f << 12) | (uint32_t)bit; ++ne; });
        if (live) count[p] = n < 0 ? (uint16_t)0xFFFF : (uint16_t)n;
        CZK_WAVE_SYNC();
        uint4 *dst = reinterpret_cast<uint4 *>(moves + (size_t)g0 * CZD_MAXMOVES);
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const int idx = lane + 64 * k, pp = idx >> 4, j = idx & 15;
            const int npp = PAD ? 128 : __shfl(n, pp, 64);     // position pp's count (negative: not a Xiangqi set, nothing to write)
            if (pp < np && 8 * j < npp) {
                const uint32_t *src = rows + pp * CZK_LROW + 4 * j;
                dst[idx] = make_uint4(src[0], src[1], src[2], src[3]);
            }
        }
        if constexpr (MASK) {   // all 64 mask rows at once: 16 896 bytes, the same fourteen 1 280-byte LDS granules as the list rows
            const bool mal16 = (reinterpret_cast<uintptr_t>(mask) & 15u) == 0;
            CZK_WAVE_SYNC();   // the list rows have left
            static_assert(64 * CZ_MASK_WORDS / 4 == 16 * 64 + 32, "64 mask rows are 16 1/2 rounds of 16-byte pieces");
#pragma unroll
            for (int k = 0; k < 16; ++k) reinterpret_cast<uint4 *>(rows)[lane + 64 * k] = make_uint4(0, 0, 0, 0);
            if (lane < 32) reinterpret_cast<uint4 *>(rows)[lane + 1024] = make_uint4(0, 0, 0, 0);
            CZK_WAVE_SYNC();
            {
                uint32_t *row = rows + lane * CZ_MASK_WORDS;   // the lane's own row: no other lane touches it
#pragma unroll
                for (int k = 0; k < CZM_EMITS; ++k)
                    czm_or_field([row](int wi, uint32_t x) { atomicOr(&row[wi], x); }, (int)(recs[k] & 0xFFFu), recs[k] >> 12);
            }
            CZK_WAVE_SYNC();
            uint32_t *dstm = mask + (size_t)g0 * CZ_MASK_WORDS;
            if (mal16 && np == 64) {
#pragma unroll
                for (int k = 0; k < 16; ++k) reinterpret_cast<uint4 *>(dstm)[lane + 64 * k] = reinterpret_cast<const uint4 *>(rows)[lane + 64 * k];
                if (lane < 32) reinterpret_cast<uint4 *>(dstm)[lane + 1024] = reinterpret_cast<const uint4 *>(rows)[lane + 1024];
            } else {
                for (int i = lane; i < np * CZ_MASK_WORDS; i += 64) dstm[i] = rows[i];
            }
        }
    }
}

// Zobrist key of a position (the oracle's cz_zhash): one lane = one position, as in k_movegen_mask — a wave stages
// its 64 boards through LDS with 16-byte loads, every lane pulls its 90 bytes out as 23 dwords, and the 15 x 90 + 1 keys live
// in LDS: 90 ds_read_b64 per position instead of 90 dependent byte loads and 64-bit gathers from global memory per thread
// (round 3: 1.1 G positions/s on 98 bytes per position).  Code 0's keys are zero, so empty squares need no branch.
// Round 6: the 11.5 KB key table is shared by the FOUR waves of a workgroup (one table per wave had left 9 waves per CU, and SQ
// counters showed 69 % of a wave's life parked in s_waitcnt: profiles/r05p_pmc_sq_rules.txt) — 34.6 KB per workgroup, 16 waves
// per CU — and a wave requests its NEXT group's boards into registers before it hashes the current one, as the move generators do.
#define CZK_HASH_WAVES 4
__global__ __launch_bounds__(64 * CZK_HASH_WAVES) void k_hash(CzTables tab, const uint8_t *__restrict__ boards, const uint8_t *__restrict__ side, int G,
                                                              uint64_t *__restrict__ hash) {
    __shared__ __attribute__((aligned(16))) uint32_t stage_all[CZK_HASH_WAVES][64 * CZ_NSQ / 4 + 4];
    __shared__ uint64_t Z[16 * CZ_NSQ];   // 15 x 90 keys + the side key at [15 * 90]; the rest of row 15 is zero: a byte above 14 (not a piece code) hashes as an empty square instead of reading past the table
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t *stage = stage_all[wv];
    for (int i = threadIdx.x; i < 16 * CZ_NSQ; i += 64 * CZK_HASH_WAVES) Z[i] = i <= 15 * CZ_NSQ ? tab.zob[i] : 0ull;
    __syncthreads();   // the only workgroup-wide one: from here on every wave walks its own groups
    const int ngroups = (G + 63) >> 6, nwaves = gridDim.x * CZK_HASH_WAVES, w0 = blockIdx.x * CZK_HASH_WAVES + wv;
    const bool al16 = (reinterpret_cast<uintptr_t>(boards) & 15u) == 0;
    uint4 pre[6];
    int presd = 0;
    auto prefetch = [&](int grp) {
        const int g0 = grp * 64, np = min(64, G - g0), nbytes = np * CZ_NSQ;
        const uint8_t *src = boards + (size_t)g0 * CZ_NSQ;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int i = lane + 64 * k;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (i * 16 + 16 <= nbytes) v = reinterpret_cast<const uint4 *>(src)[i];
            pre[k] = v;
        }
        presd = (lane < np && side[g0 + lane]) ? 1 : 0;
    };
    if (al16 && w0 < ngroups) prefetch(w0);
    for (int grp = w0; grp < ngroups; grp += nwaves) {
        const int g0 = grp * 64, np = min(64, G - g0), nbytes = np * CZ_NSQ;
        const uint8_t *src = boards + (size_t)g0 * CZ_NSQ;
        CZK_WAVE_SYNC();   // the previous group's bytes have been read
        int sd;
        if (al16) {
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (lane + 64 * k < 64 * CZ_NSQ / 16) reinterpret_cast<uint4 *>(stage)[lane + 64 * k] = pre[k];
            const int full = nbytes & ~15;   // the ragged piece of a batch's last group byte by byte, never past the batch
            if (np < 64 && lane < nbytes - full) reinterpret_cast<uint8_t *>(stage)[full + lane] = src[full + lane];
            sd = presd;
        } else {
            for (int i = lane; i < nbytes; i += 64) reinterpret_cast<uint8_t *>(stage)[i] = src[i];
            sd = (lane < np && side[g0 + lane]) ? 1 : 0;
        }
        CZK_WAVE_SYNC();
        uint32_t d[24];
        {
            const int b0 = (CZ_NSQ * lane) >> 2;
#pragma unroll
            for (int k = 0; k < 24; ++k) d[k] = stage[b0 + k];
        }
        if (al16 && grp + nwaves < ngroups) prefetch(grp + nwaves);   // in flight while this group is hashed
        if (lane < np) {
            const uint32_t sh = (lane & 1) * 16;
            uint64_t h = sd ? Z[15 * CZ_NSQ] : 0ull;
#pragma unroll
            for (int k = 0; k < 23; ++k) {
                const uint32_t w = __builtin_amdgcn_alignbit(d[k + 1], d[k], sh);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = 4 * k + j;
                    if (q < CZ_NSQ) { const uint32_t c = (w >> (8 * j)) & 0xFFu; h ^= Z[(c < 15u ? c : 0u) * CZ_NSQ + q]; }
                }
            }
            hash[g0 + lane] = h;
        }
    }
}

// K3: one wave per position, walking its positions with a stride.  The next board (two bytes per lane) and side byte are requested
// before the current planes are written, and between positions only the LDS counter drains (CZK_WAVE_SYNC), so a position's
// 2.9 KB of stores are not waited for before the next board is asked for.
template <typename T>
__global__ __launch_bounds__(64) void k_encode_planes(const uint8_t *__restrict__ boards, const uint8_t *__restrict__ side,
                                                      int G, T *__restrict__ planes, int C, int quirk, T one) {
    __shared__ __attribute__((aligned(16))) uint8_t b[CZD_BOARD_LDS];
    const int lane = threadIdx.x;
    uint8_t p0 = 0, p1 = 0, psd = 0;
    auto prefetch = [&](int g) {
        const uint8_t *src = boards + (size_t)g * CZ_NSQ;
        if (lane < 45) { p0 = src[2 * lane]; p1 = src[2 * lane + 1]; }
        psd = side[g];
    };
    if ((int)blockIdx.x < G) prefetch(blockIdx.x);
    for (int g = blockIdx.x; g < G; g += gridDim.x) {
        CZK_WAVE_SYNC();   // the previous board has been read
        if (lane < 48) { b[2 * lane] = lane < 45 ? p0 : (uint8_t)0; b[2 * lane + 1] = lane < 45 ? p1 : (uint8_t)0; }
        const int sd = psd ? 1 : 0;
        if (g + (int)gridDim.x < G) prefetch(g + gridDim.x);   // in flight while this position's planes leave
        CZK_WAVE_SYNC();
        czd_wave_encode_planes<T>(b, sd, quirk, planes + (size_t)g * 90 * C, C, one, lane);
    }
}

inline int grid_for(int G) { return G < 65536 ? G : 65536; }

}  // namespace

int czk_movegen(cz_ctx *c, const uint8_t *boards, const uint8_t *side, int G, uint16_t *moves, uint16_t *count, uint32_t *mask, int flags) {
    if (G == 0) return CZ_OK;
    const bool pad = !(flags & CZ_MOVES_NO_PAD);
    if (moves && (reinterpret_cast<uintptr_t>(moves) & 15u)) { cz_set_error("cz_movegen: moves must be 16-byte aligned"); return CZ_EINVAL; }
    const int ngroups = (G + 63) / 64;
    if (moves) {    // the reference's ordered list (and, from the same launch, the set): one lane per position, 8-9 persistent waves per CU
        const int chip = 256 * 8;
        const dim3 gm(ngroups < chip ? ngroups : chip), gl(ngroups < 256 * 9 ? ngroups : 256 * 9);
        if (mask && pad) hipLaunchKernelGGL((k_movegen_list<true, true>), gm, dim3(64), 0, c->stream, c->mask_tab, boards, side, G, moves, count, mask);
        else if (mask) hipLaunchKernelGGL((k_movegen_list<true, false>), gm, dim3(64), 0, c->stream, c->mask_tab, boards, side, G, moves, count, mask);
        else if (pad) hipLaunchKernelGGL((k_movegen_list<false, true>), gl, dim3(64), 0, c->stream, c->mask_tab, boards, side, G, moves, count, mask);
        else hipLaunchKernelGGL((k_movegen_list<false, false>), gl, dim3(64), 0, c->stream, c->mask_tab, boards, side, G, moves, count, mask);
    } else {        // the set alone (k_movegen_mask: mask and count; mask may be NULL: counts only): 12 persistent waves per CU
        const int chip = 256 * 12;
        hipLaunchKernelGGL(k_movegen_mask, dim3(ngroups < chip ? ngroups : chip), dim3(64), 0, c->stream, c->mask_tab, boards, side, G, count, mask);
    }
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int czk_apply_move(cz_ctx *c, uint8_t *boards, uint8_t *side, const uint16_t *label, int G, uint64_t *hash, uint8_t *captured, int8_t *terminal) {
    if (G == 0) return CZ_OK;
    hipLaunchKernelGGL(k_apply_move, dim3((G + 255) / 256), dim3(256), 0, c->stream, c->tab, boards, side, label, G, hash, captured, terminal);
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int czk_hash(cz_ctx *c, const uint8_t *boards, const uint8_t *side, int G, uint64_t *hash) {
    if (G == 0) return CZ_OK;
    { const int ngroups = (G + 63) / 64, wgs = (ngroups + CZK_HASH_WAVES - 1) / CZK_HASH_WAVES, chip = 256 * 4;   // 34.6 KB of LDS per workgroup of four waves: 4 persistent workgroups per CU
      hipLaunchKernelGGL(k_hash, dim3(wgs < chip ? wgs : chip), dim3(64 * CZK_HASH_WAVES), 0, c->stream, c->tab, boards, side, G, hash); }
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int czk_encode_planes(cz_ctx *c, const uint8_t *boards, const uint8_t *side, int G, void *planes, int dtype, int C, int quirk) {
    if (G == 0) return CZ_OK;
    if (dtype == CZ_F32)
        hipLaunchKernelGGL(k_encode_planes<float>, dim3(grid_for(G)), dim3(64), 0, c->stream, boards, side, G, (float *)planes, C, quirk, 1.0f);
    else
        hipLaunchKernelGGL(k_encode_planes<uint16_t>, dim3(grid_for(G)), dim3(64), 0, c->stream, boards, side, G, (uint16_t *)planes, C, quirk, (uint16_t)(dtype == CZ_F16 ? 0x3C00 : 0x3F80));
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

/*
 * pfac_hip.hip -- the MI355X (gfx950 / CDNA4) PFAC scan: kernels + C-ABI runtime.
 *
 * Replaces master_kernel.cu of the reference (TraceTable_kernel :92-180,
 * SUBSEG_MATCH :37-74, GPU_Malloc_Memory :188-257, GPU_TraceTable :277-455,
 * GPU_Free_memory :457-524).  Written for 64-wide wavefronts; not a
 * translation: the reference gives every offset a dense max_pat_len-slot
 * result row (4*max_pat_len bytes of HBM traffic per input byte, three times
 * over); this kernel emits compact, globally ORDERED records in a single pass
 * over the input: one 32-bit word  pos_in_tile:12 | final_state:20  per match
 * plus one 64-bit first-record index per 4 KiB tile (automata with more than
 * 2^20 final states fall back to 8-byte {pos, state} records).
 *
 * Kernel structure (one persistent workgroup per CU = up to 15 compute waves + 1 coordinator wave;
 * the unit of work is a WAVE TILE, 4 KiB of input owned by one wavefront; no workgroup barrier in the loop):
 *   ticket   the coordinator takes one BATCH of tiles (one per compute wave) per round with a single
 *            global atomic, three rounds ahead (four interleaved ticket counters)
 *   stage    each wave loads its 4 KiB with 16-B-per-lane buffer loads (hardware bounds check ->
 *            bytes past n_avail read as 0), one round ahead, and mirrors them + the max_pat_len-1
 *            halo into its private LDS region
 *   root     32-bit "has a root edge" mask per lane per 32 contiguous bytes (two halves per tile): SWAR
 *            compare + v_dot4 when the root has a single edge, else one LDS flag lookup per byte (mk.cu:41)
 *   level 2  most walks die on their second byte, so survivors are classified before any walk: DEEP when the
 *            byte pair starts a path of length 2 in the trie (bit-parallel SWAR compare for a single-edge root
 *            with <= 2 grandchildren, else one lookup per survivor in a 2-byte-prefix bitmap in LDS), SHALLOW-FINAL
 *            when they are not deep but the depth-1 state is final (exactly one record, known without a walk);
 *            everything else is dropped on the spot
 *   compact  DPP prefix sums append the kept survivors' positions, in order, to a FIFO in LDS; a tile without deep
 *            survivors writes its records straight to the staging buffer instead (no FIFO, no walk)
 *   walk     rounds of 64 (x NWALK) FIFO entries; straight-line, predicated: root row and the dense depth-1 rows
 *            from LDS, deeper states through the perfect hash  state = PHF(state, byte)  from LDS (small tables)
 *            or L2 (large: 2-4 walks per lane, slots fused with the next state's r[] so a step is one gather);
 *            final states are recorded as they are met (mk.cu:49-71); a round without deep entries skips the walk
 *   stage    records (pos:12 | state:20) go to an LDS staging buffer in (position, length) order
 *   place    per-wave counts -> coordinator, which places the round's tiles back to back in the workgroup's current
 *            CHUNK of the record heap (one atomic on the heap cursor per chunk, the next chunk always on order: no
 *            workgroup ever waits for another), writes every wave's first record index back to LDS and the tiles'
 *            index words (first record | count << 40) to memory -- all in the round the counts came in
 *   emit     one round later the wave copies its staged words to the heap, 16 B per lane.  Walking the tile index
 *            in order gives the records sorted by (position, pattern length), the reference's output order
 *            (main.cc:341-349).  Tiles with more records than the staging buffer holds are re-walked writing
 *            straight to the heap; "dense mode" (one big staging buffer, synchronous emission) takes over when most
 *            tiles are like that.
 */
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <type_traits>
#include <string>
#include <vector>

#include "pfac.h"

namespace {

// ---------------------------------------------------------------------------
// geometry.  The unit of work is a WAVE TILE: 4 KiB of input owned by one wavefront.
constexpr int WAVE = 64;
constexpr int SUB = WAVE * 16;             // bytes one wave covers with one 16-B-per-lane load (1 KiB)
constexpr int SUBS = 4;                    // such sub-tiles per wave tile
constexpr int WTILE = SUB * SUBS;          // 4096: tile-local positions fit 12 bits
constexpr int MSUBS = 2;                   // root test / compaction: the tile as 2 halves, a lane owning 32 contiguous bytes of each
constexpr int MSUB = WTILE / MSUBS;        // 2048
constexpr int MLANE = MSUB / WAVE;         // 32 bytes per lane -> one 32-bit survivor mask
constexpr int HALO_MAX = 1024;             // >= max_pat_len - 1 (patterns are < 1024 bytes), multiple of 16
constexpr int QCAP = SUB / 2 + 4 * WAVE;   // survivor FIFO: < one round (at most 4 x 64) carried over + up to 512 appended at a time
#ifndef PFAC_CAPW
#define PFAC_CAPW 256
#endif
constexpr int CAPW = PFAC_CAPW;            // records staged in LDS per tile per buffer (more -> synchronous re-walk)
constexpr int PACK_STATE_BITS = 20;        // staged record = pos:12 | state:20 (larger automata re-walk)
constexpr int MAX_WAVES_PER_BLOCK = 16;     // 15 compute waves + the coordinator
constexpr int LDS_TOTAL = 160 * 1024;
constexpr int LDS_TABLE_MAX = 40 * 1024;   // PHF tables up to this size are staged in LDS (variant 0)

// shared (per workgroup) LDS: root row, 4 pre-shifted byte-wide flag tables (root-edge flag << k | can-be-a-second-byte
// flag << (k + 4), one table per byte of a dword), then the PHF tables (variant 0)
constexpr int SH_HDR = 0;                  // round rings (H_* below)
constexpr int SH_S0 = SH_HDR + 2048;
constexpr int SH_FTAB = SH_S0 + 256 * 4;
constexpr int SH_D1IDX = SH_FTAB + 4 * 256;  // 256 x u8: dense-row index of the depth-1 state reached on each root byte
constexpr int SH_FIN = SH_D1IDX + 256;     // 256 x u8: 1 where the depth-1 state reached on that root byte is final
constexpr int SH_COLMAP = SH_FIN + 256;    // 256 x u8: column of the dense rows a second byte maps to (the last column = no edge)
constexpr int SH_D1 = SH_COLMAP + 256;     // d1_rows dense rows int32[d1_stride] (the hot first-level transition rows), only the
                                           // columns of bytes that ARE the second byte of some pattern + one "no edge" column
constexpr int D1_LDS_MAX = 32 * 1024;      // the depth-1 states get dense rows when rows x columns x 4 B fit this (else none do)
constexpr int D1_STATE_BITS = 20;          // packed dense-row entry (FUSED): state | index of its r[] << 20
constexpr int D1_N2_MAX = 2048;            // ... so at most this many depth-2 states (the entry stays positive)
// the PHF tables (variant 0) follow the dense rows: SH_D1 + d1_lds_bytes; the 2-byte-prefix bitmap (bm2_rows x 32
// bytes) sits at ScanArgs::sh_bm2, behind them
#ifndef PFAC_L2F_UNROLL
#define PFAC_L2F_UNROLL 1
#endif
constexpr int L2F_UNROLL = PFAC_L2F_UNROLL;  // survivors classified per trip of the level-2 lookup loop
#ifndef PFAC_MASK_GATHERS
#define PFAC_MASK_GATHERS 1
#endif
constexpr bool MASK_GATHERS = PFAC_MASK_GATHERS != 0;   // fused walks: exec-mask the table gathers of dead walkers
#ifndef PFAC_LOAD_AUX
#define PFAC_LOAD_AUX 2
#endif
constexpr int LOAD_AUX = PFAC_LOAD_AUX;    // cache policy of the tile loads (0 default, 2 = nt: the input is read once)
#ifndef PFAC_PAIR_DENSITY_PCT
#define PFAC_PAIR_DENSITY_PCT 40              // level-2 filter: flags only (mode 3) from this share of (first, second) byte combinations on
#endif
constexpr int QDEEP = 0x8000;              // FIFO entry = tile-local position | QDEEP when the survivor needs a walk
// per-wave LDS: tile bytes + halo | survivor FIFO | nbuf record staging buffers.  A tile's records leave nbuf - 1 rounds
// after it was scanned (its record base needs every count of its round).  nbuf = 3 where LDS allows: the stores then
// go out at the TOP of a round, right behind the next tile's loads, and have the whole round to complete -- with two
// buffers they can only go at the END (the base of the previous round is not in earlier), and since loads and stores
// retire through one in-order counter (vmcnt) the next round's wait for its tile also waits for their write
// acknowledgements (measured: 6 % of the headline kernel).
constexpr int NBUF_MAX = 3;
constexpr int CAPW3_MIN = 192;             // the three-buffer layout needs room for this many records per buffer
constexpr int PW_FIXED_1BUF = WTILE + QCAP * 2;              // + nbuf * stage_cap * 4
// Dense mode (most tiles hold more matches than CAPW, e.g. a dictionary on text): ONE big staging buffer per
// wave and synchronous emission -- fewer waves fit, but a tile is walked once instead of twice.
constexpr int CAPW_DENSE = 2048;
constexpr int PW_FIXED_DENSE = PW_FIXED_1BUF + CAPW_DENSE * 4;
// Dense mode, second form (dense2_tile below: walker slots refilled the moment a walk ends, records logged out of order
// and put in order on their way out).  Per wave, behind the tile + halo: ring of pending walkers | match count of every
// tile position (4 bits each) | record prefix of every 8 positions (u16).
constexpr int D2_RING = 256;               // ring entries of 8 bytes (a power of two, >= 3 front-end batches of 64)
constexpr int D2_CNT_OFF = D2_RING * 8;
constexpr int D2_PREF_OFF = D2_CNT_OFF + WTILE / 2;
constexpr int D2_AUX = D2_PREF_OFF + (WTILE / 8) * 2;
static_assert(D2_AUX >= QCAP * 2 && D2_AUX % 16 == 0, "the survivor FIFO of the fallback pass lies in the same bytes");
constexpr int PW_FIXED_DENSE2 = WTILE + D2_AUX;              // (the record log is in device memory)
constexpr unsigned D2_LOG_CAP = 4096;     // words of log per wave (device memory; a tile with more records goes the classic way)

// Batch tickets: one address sustains ~85 M atomics/s, and at 4 TB/s with 60 KiB per ticket the workgroups ask for
// 65 M/s -- the single counter was the floor of the whole kernel (3.9 us per round with the scan compiled out).  So
// there are TICKET_WAYS counters, 256 bytes apart: workgroup j draws from counter j % ways and holds batches
// ticket * ways + j % ways.  Ids stay monotone per workgroup (it leaves when ITS counter runs past the end) and every
// batch below the end is drawn by somebody as long as every residue class has a workgroup (ways <= grid).  Nothing
// waits for a batch: a workgroup's record placement depends on no other workgroup (see "Record placement").
#ifndef PFAC_TICKET_WAYS
#define PFAC_TICKET_WAYS 4
#endif
constexpr unsigned TICKET_WAYS = PFAC_TICKET_WAYS;
constexpr unsigned CTL_WORDS = 64u * (TICKET_WAYS + 1);   // control header in 32-bit words: one 256-byte line per ticket counter + the cursor's
constexpr unsigned SPIN_MAX = 1u << 22;    // bounded spins (default; PFAC_SPIN_MAX): ~0.5 s of LDS polls, seconds of global polls
// words of the control header (first ticket line) the kernel reports through: device memory, ordinary device atomics
constexpr unsigned CTL_ERR = 32, CTL_OVF = 33, CTL_DONE = 34, CTL_OVF2 = 35, CTL_TOTAL = 36 /* u64: matches */;
constexpr unsigned CTL_CURSOR = 64u * TICKET_WAYS;     // u64: first free record of the heap (on a line of its own)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ScanArgs {
    const unsigned char *in;
    unsigned long long n_owned, n_avail;
    void *out;                            // the record heap: out_cap records of rec_bytes bytes
    unsigned long long out_cap;
    unsigned long long *tile_index;       // [n_tiles]: first record of the tile | its record count << 40
    unsigned chunk;                       // records a workgroup takes from the heap cursor at a time (0: exact allocations only)
    unsigned rec_bytes;                   // record form of this scan: 2 (pos:12 | state:4), 4 (pos:12 | state:20) or 8 (pfac_record)
    const int *s0;
    const int *r;
    const int2 *T;
    const int4 *T4;                       // FUSED: {owner row, next state, r[row of next state], 0} per slot (tables via L2)
    int r_words, t_entries;
    int ht_size, wbit, num_final, halo;   // halo: bytes readable past a tile, multiple of 16
    int rn_bias;                          // fused tables: T4 starts at slot -rn_bias (displacements may be negative); every index
                                          // the fused walk forms is slot + rn_bias >= 0
    int shared_bytes, pw_bytes;           // LDS carve: shared region, then one region per wave
    const int *d1;                        // dense rows of the depth-1 states, d1_rows x 256 (or null); in LDS: d1_rows x d1_stride
    const unsigned char *d1idx;           // root byte -> dense row index
    const unsigned char *d1_colmap;       // [256] second byte -> LDS column (d1_stride - 1: no pattern has it as second byte)
    const unsigned char *d1_colbyte;      // [d1_ncols] the byte of each LDS column
    int d1_rows;                          // 0: no dense level
    int d1_stride, d1_ncols, d1_lds_bytes;  // LDS row length in words (columns, + 1 "no edge" unless all 256 are used);
                                          // columns; bytes of the LDS rows (multiple of 16)
    const int2 *d1r2;                     // FUSED + packed dense rows: {r[], child mask} of the depth-2 states, d1_n2 entries (else null)
    int d1_n2;                            // > 0: a dense-row entry is  state | index into d1r2 << 20  (or -1)
    unsigned root_byte;                   // ROOT == 1: the only byte with a root edge, replicated x4
    int root_state;                       // ROOT == 1: the state that byte leads to (every survivor starts there)
    // level-2 filter (which survivors can go beyond their second byte):
    //   0 off: every survivor is walked        1 (ROOT == 1): SWAR compare with <= 2 child bytes of root_state
    //   2: one lookup per survivor in the 2-byte-prefix bitmap (bm2_rows = 256 rows of 32 bytes; ROOT == 1: its one row)
    //   3 (ROOT == 0, sec_filter): second-byte flags only -- survivors whose next byte can be a second byte are all walked
    int l2f_mode;
    unsigned child0, child1;              // mode 1: the child bytes, replicated x4 (n_child 1: child1 == child0)
    int n_child;                          // mode 1: 0 (nothing is ever deep), 1 or 2
    const unsigned char *bm2;             // mode 2: bit (b0 << 8 | b1) set iff a path b0 b1 leaves the root
    int bm2_rows, sh_bm2;                 // rows staged in LDS (256, or 1 = the root byte's row) at this LDS offset
    int sh_t0;                            // packed dense rows: LDS offset of the root table of dense2_tile (256 words), else 0
    const unsigned char *sec2;            // [256]: 1 where the byte is the second byte of some pattern (column OR of bm2)
    int sec_filter;                       // ROOT == 0, mode 2: pre-filter the lookups with sec2 (no 1-byte patterns)
    unsigned stage_cap;                   // records one staging buffer holds (0: final states do not fit the packed word)
    unsigned nbuf;                        // staging buffers per wave: 3 / 2 = emit two / one round(s) later, 1 = emit at once (dense mode)
    int dense2;                           // dense mode on fused tables with packed dense rows: the refilled-walker form (dense2_tile)
    unsigned *d2log;                      // ... its record logs: d2log_cap words per compute wave of the grid (device memory)
    unsigned d2log_cap;
    unsigned sparse_cap;                  // tiles with more matches than this are counted in res[3] (mode adaptation) ...
    unsigned small_cap;                   // ... and than this (the three-buffer capacity) in res[6]
    unsigned n_tiles;
    unsigned spin_max;             // bound of every spin loop
    unsigned fault;                // test knob (PFAC_FAULT): bit 0 = workgroup 1 never publishes the record bases of its second round
    unsigned *ctl;                 // control header (device memory): batch ticket counters 64 words apart; [CTL_ERR/OVF/DONE/TOTAL]:
                                   // error flags, tiles denser than sparse_cap, workgroups that have left, matches; [CTL_CURSOR]
    unsigned ticket_ways;          // counters in use: min(TICKET_WAYS, grid)
    uint4 *zero_next;              // the slot's OTHER control header: this launch zeroes it for the next one ...
    unsigned zero_vec;             // ... this many 16-byte units (no memset between back-to-back scans)
    unsigned *res;                 // host-mapped pinned words the host reads after the stream sync, no D2H copy, written with
                                   // plain stores by the last workgroup to leave: [0..1] matches (u64), [2] error flags,
                                   // [3] tiles denser than sparse_cap, [4..5] heap cursor = records of capacity used (u64)
    unsigned long long *dbg;       // PFAC_TRACE_BUILD only: per-round timestamps (10 ns units), else null
};

// ---------------------------------------------------------------------------
// wave helpers (wave = 64 lanes)

// Inclusive prefix sum over the 64 lanes with DPP (no LDS round trips): Hillis-Steele inside each
// row of 16 lanes (row_shr 1,2,4,8), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows
// 2 and 3.  Lanes without a source keep `old` = 0 (bound_ctrl off), i.e. add the identity.
__device__ __forceinline__ unsigned wave_incl_scan(unsigned x) {
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}
__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, WAVE);
    return x;
}
__device__ __forceinline__ unsigned bcast_last(unsigned x) { return __builtin_amdgcn_readlane(x, WAVE - 1); }
// LDS written by some lanes of a wave and read by others of the SAME wave:
// the LDS executes one wave's instructions in order; this keeps the compiler
// from moving accesses across and drains lgkmcnt.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// Workgroup organisation.  A workgroup = NC compute waves + 1 coordinator wave, and proceeds in
// ROUNDS: in round r compute wave c scans tile  batch(r) * NC + c.  Batches are handed out in order by
// one global atomic per round (a ticket per 4 KiB tile would hit the ~88 M atomics/s an address
// sustains on MI355X and cap the scan at 0.36 TB/s), taken AHEAD rounds ahead by the coordinator.
// The coordinator also sums the round's per-wave match counts and places the round's tiles in the
// workgroup's chunk of the record heap, while the compute waves are already scanning the next round;
// they pick their first record index up from LDS when they emit, one round later.  No workgroup barrier
// in the loop, and no wait on any other workgroup: nothing depends on dispatch order, residency or
// placement (several grids of this kernel may run at once -- gphf's slots do).  The waits inside a
// workgroup (rings below) are bounded; a timeout ends the scan with PFAC_E_INTERNAL instead of hanging.
//
// LDS header (unsigned words), rings of 8 rounds indexed by r & 7:
constexpr int RING = 8;
#ifndef PFAC_AHEAD
#define PFAC_AHEAD 2
#endif
constexpr int AHEAD = PFAC_AHEAD;          // rounds the batch ring runs ahead of the coordinator's own round
static_assert(AHEAD >= 2 && AHEAD <= RING - 3, "ring depth");
#ifndef PFAC_LOAD_DEPTH
#define PFAC_LOAD_DEPTH 1
#endif
constexpr int LOAD_DEPTH = PFAC_LOAD_DEPTH; // tiles a compute wave keeps in flight ahead of the one it scans (1 or 2)
static_assert(LOAD_DEPTH == 1 || (LOAD_DEPTH == 2 && AHEAD >= 3), "two tiles in flight need the ring three rounds ahead");
constexpr int H_BATCH = 0;                 // batch id of round r
constexpr int H_EPOCH = 8;                 // == r + 1 once H_BATCH (and a zeroed H_ARRIVED) are valid
constexpr int H_ARRIVED = 16;              // compute waves that have posted their count
constexpr int H_READY = 24;                // == r + 1 once H_WBASE of round r is valid
constexpr int H_CNT = 48;                  // 16 words per round: match count of each compute wave
constexpr int H_WBASE = H_CNT + RING * 16; // 32 words per round: {lo, hi} first record index of each compute wave
constexpr int H_OVF = H_WBASE + RING * 32;  // tiles of this workgroup with more matches than sparse_cap
constexpr int H_EXIT = H_OVF + 1;          // waves of this workgroup that have left the kernel
constexpr int H_OVF2 = H_OVF + 2;          // tiles of this workgroup with more matches than small_cap
constexpr int H_WORDS = H_OVF + 8;

// Error channel: flags are OR-ed into the control header in DEVICE memory (ordinary device atomics; the last
// workgroup to leave copies them to the host-mapped result words with plain stores).
struct ErrCh {
    unsigned *word;
    unsigned spin_max;
};
__device__ __forceinline__ void err_set(const ErrCh &e, unsigned code) {
    __hip_atomic_fetch_or(e.word, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ unsigned lds_load(const unsigned *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store(unsigned *p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// Spin (bounded) until *p == want.  Wave-uniform: every lane reads the same word.
__device__ __forceinline__ bool lds_wait_eq(const unsigned *p, unsigned want, const ErrCh &err, unsigned code) {
    unsigned spins = 0;
    while (lds_load(p) != want) {
        if (++spins >= err.spin_max) { err_set(err, code); return false; }
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return true;
}

// ---------------------------------------------------------------------------
// Record placement.  The record array is a HEAP, the tile index is what is ordered: tile t's records are the
// tile_index[t] >> 40 words (records) that start at tile_index[t] & (2^40 - 1), in (position, pattern length) order;
// tiles are placed wherever their workgroup's current chunk of the array has room.  A workgroup owns a CHUNK of the
// array at a time, taken from one global cursor with a single atomic (and one spare chunk is always on order, so
// the atomic's round trip is never waited for); tiles fill the chunk back to back, a batch that does not fit in the
// rest of the chunk continues in the next one from the first tile that does not fit, a batch larger than a chunk
// gets an exact allocation of its own.
//
// Why not one globally contiguous, sorted array: that needs every batch's first record index = the matches of ALL
// earlier batches, i.e. a prefix over batches that are being scanned by other workgroups at the same time.  Both
// forms of it were built and measured (decoupled look-back over 64-batch windows; per-generation totals with one
// hop per 256 batches): with two rounds of staging as slack, every workgroup still ends up waiting for the slowest
// one of its generation, round after round -- 12 % of the match-dense headline and 15-25 % of the sparse workloads
// (ablation build PFAC_ABL_NOLB vs the full kernel).  Consumers lose nothing: they walk the tile index in order
// (pfac_records_expand / pfac_records_d2h deliver one sorted pfac_record array, pfac_emit_packed prints from the
// heap directly), the gaps are below 1 % of the array, and the chunk atomics are a few per microsecond.
constexpr unsigned long long TIX_BASE_MASK = (1ull << 40) - 1;   // tile index word = first record | count << 40
constexpr int TIX_CNT_SHIFT = 40;

// ---------------------------------------------------------------------------
// One step of the perfect-hash lookup (master_kernel.cu:52-63): returns the
// next state or -1.  HT/val are interleaved as int2 {owner row, next state}.
// W8: PHF width 256 (the benchmark width) -> row == state, col == byte.
template <bool W8>
__device__ __forceinline__ int phf_step(const int *R, const int2 *T, int state, int ch, int wbit, int ht_size) {
    int row, idx;
    if (W8) {
        row = state;
        idx = R[state] + ch;
    } else {
        const int key = (state << 8) | ch;
        row = key >> wbit;
        idx = R[row] + (key & ((1 << wbit) - 1));
    }
    if ((unsigned)idx >= (unsigned)ht_size) return -1;
    const int2 e = T[idx];
    return e.x == row ? e.y : -1;
}

// NWALK independent walks per lane, stepped together.  Walk w starts at tile-local position pos[w];
// n[w] counts the final states reached; m[w][..] keeps the latest MREG of them as a shift register, m[w][0] the latest
// (= all of them, in reverse walk order, when n[w] <= MREG).  lim = first tile-local byte that may not be read.
//  * all lanes step in lock step (trip count = deepest walk in the wave); dead lanes are predicated with
//    selects instead of nested divergent branches -- far fewer exec-mask / scalar instructions per step;
//  * input bytes come four at a time from one aligned 8-byte LDS read, so a step's only dependent
//    accesses are R and T (or one dense-row lookup for the second byte);
//  * NWALK = 2 (tables gathered through L2): every table round trip has an independent twin in flight --
//    the memory-level parallelism of twice the occupancy without the LDS more waves would need.  With
//    the tables in LDS a second walk buys nothing (measured), so NWALK = 1 there.
//  * FUSED (tables gathered through L2, PHF width >= 256): each slot also carries r[] of the state it leads to,
//    so a step is ONE 16-byte gather instead of a 4-byte one (r) and a dependent 8-byte one (T) -- the
//    dictionary-on-text regime is bound by the L2's request rate (every lane of a gather is its own request), and
//    this halves the requests; only the first hashed step of a walk still looks r[] up.
//  * NWALK = 4 (the dense-match regime on L2 tables, e.g. a dictionary on text): a walk there is a chain of L2
//    round trips and the wave has nothing else to do, so four chains per lane run side by side; the first
//    MREG = 4 final states of a walk stay in registers (two otherwise) -- a position where more patterns start
//    costs a second, serial walk (walk_store).
//  * ROOT == 1 (one byte leads out of the root): every survivor's first state is the same one and the dense row is
//    row 0, so the root-row and row-index lookups -- two dependent LDS round trips per round -- disappear.
//  * deepf[w] == false (a SHALLOW-FINAL survivor, see the level-2 filter): the walk accounts for its depth-1 state
//    and stops there.
template <bool W8, int NWALK, bool FUSED, int MREG, int ROOT>
__device__ __forceinline__ void walkN(const unsigned char *tile, const int *s0, int root_state, const unsigned char *d1idx,
                                      const unsigned char *colmap, unsigned d1_stride, const int *D1, bool dense1, const int2 *D1R2, const int *S0R, const int *R, const int2 *T, const int4 *T4,
                                      const unsigned (&pos)[NWALK], const bool (&active)[NWALK], const bool (&deepf)[NWALK],
                                      unsigned lim, int wbit,
                                      int ht_size, int num_final, int rn_bias, bool skip_s0, unsigned (&n)[NWALK], unsigned (&m)[NWALK][MREG]) {
    static_assert(MREG == 2 || MREG == 4, "two or four final states per walk in registers");
    const unsigned *t32 = reinterpret_cast<const unsigned *>(tile);
    unsigned win[NWALK], left[NWALK], f[NWALK];                // left: bytes the walk may still read after its first
    int s[NWALK], rn[NWALK];                                   // rn: r[row of s] (FUSED)
    // (not with four walks per lane, the dense-match regime: there the tests cost more issue slots than the gathers they save
    // cost time -- the dictionary on text measured 3.5 % slower with them)
    constexpr bool USE_CM = FUSED && NWALK < 4;
    unsigned cm[NWALK];                                        // FUSED: child mask of s -- bit (b & 31) set iff s has an edge on some byte
                                                               // congruent to b mod 32 (all ones where it is not known): a walker whose next
                                                               // byte's bit is clear is dead WITHOUT the L2 round trip that would say so
    bool go[NWALK];
    unsigned k = 0;                                            // transitions made so far (the same for every walk: a scalar)
    const int sub = wbit - 8;                                  // FUSED needs wbit >= 8: row = state >> sub
    int4 e4[NWALK];                                            // the fused slots of the last gather
#pragma unroll
    for (int w = 0; w < NWALK; w++) {
        const unsigned lo = t32[pos[w] >> 2], hi = t32[(pos[w] >> 2) + 1];   // may run a few bytes past lim: never used
        win[w] = __builtin_amdgcn_alignbyte(hi, lo, pos[w] & 3u);          // bytes pos .. pos+3
    }
#pragma unroll
    for (int w = 0; w < NWALK; w++) {
        // (dense second byte and no 1-byte pattern: the depth-1 state itself is never looked at -- any live, non-final value will do)
        const int st = ROOT == 1 ? root_state : (skip_s0 ? num_final : s0[win[w] & 0xFFu]);
        f[w] = ROOT == 1 ? 0u : d1idx[win[w] & 0xFFu];        // dense row of that state (when dense1)
        s[w] = active[w] ? st : -1;
        n[w] = 0;
#pragma unroll
        for (int k = 0; k < MREG; k++) m[w][k] = 0;
        left[w] = (deepf[w] && lim > pos[w] + 1u) ? lim - pos[w] - 1u : 0u;
        go[w] = false;
        rn[w] = 0;
        cm[w] = 0xFFFFFFFFu;
    }
    // account for the states just reached; false when no lane of the wave can go on.
    // final <=> (unsigned)s < num_final, which also rejects the dead state -1.
    // masked (FUSED): byte `bi` of the window is the NEXT byte of the walk; a walker whose state has no edge on any byte
    // congruent to it mod 32 (child mask) is dead here and now -- no gather, and the loop can end a step earlier
    auto reached = [&](int bi, bool masked) -> bool {
        bool any = false;
#pragma unroll
        for (int w = 0; w < NWALK; w++) {
            const bool fin = (unsigned)s[w] < (unsigned)num_final;
            // the latest MREG final states as a shift register (m[0] = latest): one select each, no index compare;
            // the caller puts them back in walk order
#pragma unroll
            for (int j = MREG - 1; j > 0; j--) m[w][j] = fin ? m[w][j - 1] : m[w][j];
            m[w][0] = fin ? (unsigned)s[w] : m[w][0];
            n[w] += fin ? 1u : 0u;
            go[w] = s[w] >= 0 && k < left[w];
            if (USE_CM && masked) go[w] = go[w] && ((cm[w] >> ((win[w] >> (8 * bi)) & 31u)) & 1u) != 0u;
            any = any || go[w];
        }
        return __any(any);
    };
    // fused slot number i (biased: >= 0, see ScanArgs::rn_bias) by a 32-bit byte offset from the scalar base -- one
    // shift, no 64-bit address arithmetic per lane
    auto slot = [&](int i) -> int4 {
        return *reinterpret_cast<const int4 *>(reinterpret_cast<const unsigned char *>(T4) + ((unsigned)i << 4));
    };
    // one transition on byte number `bi` of the window (straight-line: dead lanes look up a harmless, valid slot)
    auto step = [&](int bi) {
        int row[NWALK], idx[NWALK];
        if (FUSED) {
            constexpr bool MASKED = NWALK == 4 || MASK_GATHERS;
#pragma unroll
            for (int w = 0; w < NWALK; w++) {
                const int ch = (int)((win[w] >> (8 * bi)) & 0xFFu);
                // (masked gathers: a dead lane loads nothing, so its index need not be a valid one)
                const int sg = MASKED ? s[w] : (go[w] ? s[w] : 0);
                const int rg = MASKED ? rn[w] : (go[w] ? rn[w] : 0);
                if (W8) {
                    row[w] = sg;
                    idx[w] = rg + ch;
                } else {
                    row[w] = sg >> sub;
                    idx[w] = rg + (((sg & ((1 << sub) - 1)) << 8) | ch);
                }
            }
            // (a live walker's index is inside the slot array: it is padded to max(r) + width entries, and
            // pfac_repack_kernel checked every displacement and state of the image)
#pragma unroll
            for (int w = 0; w < NWALK; w++) {
                // (set before every gather although a dead lane's slot is only ever selected under go[w]: measured 15 %
                // faster than carrying the stale value -- the gather then has no dependence on the register's old content)
                e4[w] = make_int4(-1, -1, 0, 0);
                if (MASKED) {
                    // dead lanes stay out of the gather: every lane of a gather costs the texture path an address
                    // cycle (64 per wave-instruction, against 16 for a coalesced 1 KiB load), and a round's later
                    // steps have few walkers left
                    if (go[w]) e4[w] = slot(idx[w]);
                } else {
                    e4[w] = slot(idx[w]);
                }
            }
#pragma unroll
            for (int w = 0; w < NWALK; w++) {
                s[w] = (go[w] && e4[w].x == row[w]) ? e4[w].y : -1;
                rn[w] = e4[w].z;
                cm[w] = (unsigned)e4[w].w;
            }
            k++;
            return;
        }
#pragma unroll
        for (int w = 0; w < NWALK; w++) {
            const int ch = (int)((win[w] >> (8 * bi)) & 0xFFu);
            const int sg = go[w] ? s[w] : 0;
            if (W8) {
                row[w] = sg;
                idx[w] = R[sg] + ch;
            } else {
                const int key = (sg << 8) | ch;
                row[w] = key >> wbit;
                idx[w] = R[row[w]] + (key & ((1 << wbit) - 1));
            }
        }
        int2 e[NWALK];
        unsigned ic[NWALK];
#pragma unroll
        for (int w = 0; w < NWALK; w++) {
            ic[w] = min((unsigned)idx[w], (unsigned)ht_size - 1u);     // (the table ends at its last used slot)
            e[w] = T[ic[w]];
        }
#pragma unroll
        for (int w = 0; w < NWALK; w++) {
            s[w] = (go[w] && ic[w] == (unsigned)idx[w] && e[w].x == row[w]) ? e[w].y : -1;
        }
        k++;
    };
    // FUSED: the first hashed step of a walk has no slot to take r[] from
    auto load_rn = [&]() {
        if (FUSED) {
#pragma unroll
            for (int w = 0; w < NWALK; w++) rn[w] = R[(go[w] ? s[w] : 0) >> sub] + rn_bias;
        }
    };
    if (!reached(1, false)) return;
    if (dense1) {
        // second byte: the depth-1 state's row is dense in LDS -- one lookup, no hash, no owner check.  Stage by stage for
        // all the walks of the lane (column, row entry, packed r[] / child mask): written walk by walk, the uniform test on
        // D1R2 put every walk into basic blocks of its own and their LDS round trips in SERIES -- six trips instead of three
        unsigned col[NWALK];
        int nx[NWALK];
#pragma unroll
        for (int w = 0; w < NWALK; w++) col[w] = colmap[(win[w] >> 8) & 0xFFu];
#pragma unroll
        for (int w = 0; w < NWALK; w++) nx[w] = D1[(go[w] ? f[w] : 0u) * d1_stride + col[w]];
        if (FUSED && D1R2) {
            // packed entry: the depth-2 state and where its r[] sits in LDS -- the walk's first hashed step needs no r[]
            // gather either
            bool ok[NWALK];
            int2 e2[NWALK];
#pragma unroll
            for (int w = 0; w < NWALK; w++) {
                ok[w] = go[w] && nx[w] >= 0;
                e2[w] = D1R2[ok[w] ? (nx[w] >> D1_STATE_BITS) : 0];        // {r[] of the depth-2 state, its child mask}
            }
#pragma unroll
            for (int w = 0; w < NWALK; w++) {
                s[w] = ok[w] ? (nx[w] & ((1 << D1_STATE_BITS) - 1)) : -1;
                rn[w] = e2[w].x;
                cm[w] = (unsigned)e2[w].y;
            }
        } else {
#pragma unroll
            for (int w = 0; w < NWALK; w++) s[w] = go[w] ? nx[w] : -1;
        }
        k++;
    } else {
        if (FUSED && S0R) {
            // too many depth-1 states for dense rows: their r[] at least sits in LDS (by root byte), so the second
            // byte costs one gather, not two
#pragma unroll
            for (int w = 0; w < NWALK; w++) rn[w] = S0R[win[w] & 0xFFu];
        } else {
            load_rn();
        }
        step(1);
    }
    if (!reached(2, true)) return;
    if (dense1 && !(FUSED && D1R2)) load_rn();
    step(2);
    if (!reached(3, true)) return;
    step(3);
    for (;;) {                                                 // deeper than 4 bytes: next aligned windows
        if (!reached(0, false)) return;
#pragma unroll
        for (int w = 0; w < NWALK; w++) {
            const unsigned p = pos[w] + 1u + k;                // next byte of the walk
            const unsigned lo = t32[p >> 2], hi = t32[(p >> 2) + 1];
            win[w] = __builtin_amdgcn_alignbyte(hi, lo, p & 3u);
        }
        if (USE_CM) {                                          // the child masks against the first byte of the new windows
            bool any = false;
#pragma unroll
            for (int w = 0; w < NWALK; w++) {
                go[w] = go[w] && ((cm[w] >> (win[w] & 31u)) & 1u) != 0u;
                any = any || go[w];
            }
            if (!__any(any)) return;
        }
        step(0);
        if (!reached(1, true)) return;
        step(1);
        if (!reached(2, true)) return;
        step(2);
        if (!reached(3, true)) return;
        step(3);
    }
}

// One record into the global array (the DIRECT paths), in the scan's record form.
__device__ __forceinline__ void put_record(const ScanArgs &a, unsigned long long ri, unsigned tpos, unsigned gpos, unsigned state) {
    if (ri >= a.out_cap) return;
    if (a.rec_bytes == 2) {
        static_cast<unsigned short *>(a.out)[ri] = (unsigned short)(tpos | (state << 12));
    } else if (a.rec_bytes == 4) {
        static_cast<unsigned *>(a.out)[ri] = tpos | (state << 12);
    } else {
        pfac_record rec;
        rec.pos = gpos;
        rec.state = state;
        static_cast<pfac_record *>(a.out)[ri] = rec;
    }
}

// Same walk for the rare offsets where more patterns start than the fast walk keeps final states in registers:
// every final state goes to the LDS staging buffer (packed) or straight to global memory.
template <bool W8, bool DIRECT>
__device__ __forceinline__ void walk_store(const ScanArgs &a, const unsigned char *tile, const int *s0, const int *R, const int2 *T,
                                           unsigned pos, unsigned lim, unsigned *stage, unsigned long long ri, unsigned gpos) {
    unsigned n = 0;
    int s = s0[tile[pos]];
    unsigned p = pos + 1;
    while (s >= 0) {
        if (s < a.num_final) {
            if (DIRECT) put_record(a, ri + n, pos, gpos, (unsigned)s);
            else if (ri + n < a.stage_cap) stage[ri + n] = pos | ((unsigned)s << 12);
            n++;
        }
        if (p >= lim) break;
        s = phf_step<W8>(R, T, s, tile[p], a.wbit, a.ht_size);
        p++;
    }
}

struct Dense1 {
    const unsigned char *colmap;
    const unsigned char *idx;
    const int *rows;
    bool on;
    const int2 *r2;     // packed rows (FUSED): {r[], child mask} of the depth-2 states, in LDS; null = plain rows
    const int *s0r;     // no dense rows (FUSED): r[] of the depth-1 state each root byte leads to, in LDS; else null
};

// One round: up to 64*NWALK FIFO entries [q0, q0+nact), lane L takes entries L, L+64, ... side by side; their records
// are appended, in queue (= position) order, at index `wrun` of the staging buffer (DIRECT == false) or of the global
// record array.  Entries without QDEEP are shallow-final survivors: one record, the depth-1 state of their byte; a
// round made of those alone needs neither a walk nor a prefix sum.  Returns the number of records.
template <bool W8, bool DIRECT, int NWALK, bool FUSED, int ROOT>
__device__ __forceinline__ unsigned roundN(const ScanArgs &a, const unsigned char *tile, const int *s0, const Dense1 &d1,
                                           const int *R, const int2 *T, const unsigned short *q, unsigned q0,
                                           unsigned nact, int lane, unsigned *stage, unsigned lim,
                                           unsigned long long tile_base, unsigned long long wrun) {
    static_assert(NWALK >= 1 && NWALK <= 4, "one to four walks per lane");
    constexpr int MREG = NWALK == 4 ? 4 : 2;
    bool active[NWALK], deepf[NWALK];
    unsigned pos[NWALK], n[NWALK], m[NWALK][MREG];
    bool anydeep = false;
#pragma unroll
    for (int w = 0; w < NWALK; w++) {
        active[w] = (unsigned)lane + WAVE * w < nact;
        const unsigned e = active[w] ? q[q0 + WAVE * w + lane] : 0u;
        pos[w] = e & 0xFFFu;
        deepf[w] = (e & QDEEP) != 0;
        anydeep = anydeep || deepf[w];
    }
#ifdef PFAC_ABL_NOWALK                         // ablation builds only: deep entries are treated as shallow (wrong records)
    anydeep = false;
#endif
    if (!__any(anydeep)) {
#pragma unroll
        for (int w = 0; w < NWALK; w++) {
            if (!active[w]) continue;
            const unsigned st = ROOT == 1 ? (unsigned)a.root_state : (unsigned)s0[tile[pos[w]]];
            const unsigned long long ri = wrun + WAVE * w + lane;
            if (DIRECT) put_record(a, ri, pos[w], (unsigned)tile_base + pos[w], st);
            else if (ri < a.stage_cap) stage[ri] = pos[w] | (st << 12);
        }
        return nact;
    }
    walkN<W8, NWALK, FUSED, MREG, ROOT>(tile, s0, a.root_state, d1.idx, d1.colmap, (unsigned)a.d1_stride, d1.rows, d1.on, d1.r2, d1.s0r, R, T, a.T4, pos, active, deepf, lim,
                                        a.wbit, a.ht_size, a.num_final, a.rn_bias, ROOT != 1 && d1.on && a.sec_filter != 0, n, m);
    // the walk kept its latest MREG final states as a shift register (m[0] = latest): back into walk order
#pragma unroll
    for (int w = 0; w < NWALK; w++) {
        const unsigned c = n[w];
        if (MREG == 4) {
            const unsigned a0 = m[w][0], a1 = m[w][1], a2 = m[w][2], a3 = m[w][3];
            m[w][0] = c >= 4u ? a3 : (c == 3u ? a2 : (c == 2u ? a1 : a0));
            m[w][1] = c >= 4u ? a2 : (c == 3u ? a1 : a0);
            m[w][2] = c >= 4u ? a1 : a0;
            m[w][3] = a0;
        } else {
            const unsigned a0 = m[w][0], a1 = m[w][1];
            m[w][0] = c >= 2u ? a1 : a0;
            m[w][1] = a0;
        }
    }
    // prefix sums of the counts, two walks per scan (16-bit fields; a walk reports < 1024 matches)
    unsigned ex[NWALK], total = 0;
#pragma unroll
    for (int w0 = 0; w0 < NWALK; w0 += 2) {
        const bool pair = w0 + 1 < NWALK;
        const unsigned packed = pair ? (n[w0] | (n[pair ? w0 + 1 : w0] << 16)) : n[w0];
        const unsigned inc = wave_incl_scan(packed);
        const unsigned last = bcast_last(inc);
        const unsigned t0 = pair ? (last & 0xFFFFu) : last;
        ex[w0] = total + (pair ? (inc & 0xFFFFu) : inc) - n[w0];
        total += t0;
        if (pair) {
            ex[w0 + 1] = total + (inc >> 16) - n[w0 + 1];
            total += last >> 16;
        }
    }
#pragma unroll
    for (int w = 0; w < NWALK; w++) {
        const bool regs = n[w] <= (unsigned)MREG;              // every record of this walk is in registers
        const unsigned gpos = (unsigned)tile_base + pos[w];
        if (DIRECT) {
            const unsigned long long ri = wrun + ex[w];
            if (regs && n[w] > 0) put_record(a, ri, pos[w], gpos, m[w][0]);
#pragma unroll
            for (int k = 1; k < MREG; k++)
                if (regs && n[w] > (unsigned)k) put_record(a, ri + k, pos[w], gpos, m[w][k]);
            if (!regs) walk_store<W8, true>(a, tile, s0, R, T, pos[w], lim, nullptr, ri, gpos);
        } else {
            const unsigned ri = (unsigned)wrun + ex[w];        // tile-local record index: 32 bits are plenty
            if (regs && n[w] > 0 && ri < a.stage_cap) stage[ri] = pos[w] | (m[w][0] << 12);
#pragma unroll
            for (int k = 1; k < MREG; k++)
                if (regs && n[w] > (unsigned)k && ri + k < a.stage_cap) stage[ri + k] = pos[w] | (m[w][k] << 12);
            if (!regs) walk_store<W8, false>(a, tile, s0, R, T, pos[w], lim, stage, ri, 0);
        }
    }
    return total;
}

// Compaction + walk over one wave tile.  keep[j] = the survivors of half-tile j that yield a record or need a walk
// (bit b = byte b of the lane's 32), deep[j] = those of them that need the walk.  Kept survivors are appended, in
// position order, to a FIFO in LDS; whenever 64 are pending a full round runs, so lanes stay busy even when only
// one offset in thirteen survives the root test.  Returns the tile's match count; with DIRECT the records are
// written at global index wrun onwards.
template <bool W8, bool DIRECT, int NWALK, bool FUSED, int ROOT>
__device__ __forceinline__ unsigned long long tile_pass(const ScanArgs &a, const unsigned char *tile, const int *s0,
                                                        const Dense1 &d1, const int *R, const int2 *T, unsigned short *q,
                                                        unsigned *stage, const unsigned (&keep)[MSUBS],
                                                        const unsigned (&deep)[MSUBS], int lane,
                                                        unsigned lim, unsigned long long tile_base,
                                                        unsigned long long wrun) {
    unsigned head = 0, tail = 0;               // pending survivors: q[head, tail), always fewer than one round between appends
    // per-lane survivor counts of the 2 half-tiles, prefix-summed in one packed DPP scan (16-bit fields: a
    // half-tile holds at most 2048 survivors)
    static_assert(MSUBS == 2, "the packed scan assumes 2 half-tiles");
    const unsigned c0 = __popc(keep[0]), c1 = __popc(keep[1]);
    const unsigned pa = wave_incl_scan(c0 | (c1 << 16));
    const unsigned incls[MSUBS] = {pa & 0xFFFFu, pa >> 16};
    const unsigned cnts[MSUBS] = {c0, c1};
    const unsigned la = bcast_last(pa);
    const unsigned totals[MSUBS] = {la & 0xFFFFu, la >> 16};
    if (ROOT == 1 && !__any((deep[0] | deep[1]) != 0u)) {
        // No deep survivor in the tile (the usual case for a sparse pattern set): every kept survivor is exactly one
        // record, the depth-1 state -- written straight to its slot, in position order; no FIFO, no rounds.
        const unsigned st = (unsigned)a.root_state;
#pragma unroll
        for (int j = 0; j < MSUBS; j++) {
            if (totals[j] == 0) continue;
            const unsigned lpos = j * MSUB + lane * MLANE;
            unsigned o = (j ? totals[0] : 0u) + incls[j] - cnts[j];
#ifdef PFAC_ABL_NOSTAGE                        // ablation builds only: counted, never staged (stale records)
            if (true) continue;
#endif
            if (!DIRECT && totals[0] + totals[1] <= a.stage_cap) {
                // everything fits (else the tile is walked again, DIRECT): no bound checks, two records per trip
                const unsigned w0 = lpos | (st << 12);
                for (unsigned m = keep[j]; m;) {
                    stage[o] = w0 + (unsigned)__ffs(m) - 1u;
                    m &= m - 1;
                    if (m) stage[o + 1] = w0 + (unsigned)__ffs(m) - 1u;
                    m &= m - 1;                         // (0 stays 0)
                    o += 2;
                }
                continue;
            }
            for (unsigned m = keep[j]; m; m &= m - 1) {
                const unsigned pos = lpos + (__ffs(m) - 1);
                if (DIRECT) put_record(a, wrun + o, pos, (unsigned)tile_base + pos, st);
                else if (o < a.stage_cap) stage[o] = pos | (st << 12);
                o++;
            }
        }
        return wrun + totals[0] + totals[1];
    }
    constexpr unsigned RW = WAVE * NWALK;      // survivors per round
#ifdef PFAC_TRACE_BUILD
    unsigned long long tp_round = 0, tp_n = 0, tp_t0 = __builtin_amdgcn_s_memrealtime();
#define TP_ROUND(call) do { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); call; tp_round += __builtin_amdgcn_s_memrealtime() - t_; tp_n++; } while (0)
#else
#define TP_ROUND(call) do { call; } while (0)
#endif
#pragma unroll
    for (int j = 0; j < MSUBS; j++) {
        const unsigned mask = keep[j];
        const unsigned dmask = deep[j];
        const unsigned cnt = cnts[j];
        const unsigned incl = incls[j];
        const unsigned S = totals[j];
        if (S == 0) continue;
        const unsigned lpos = j * MSUB + lane * MLANE;
        // A half-tile with more than 512 survivors is appended in four groups of 16 lanes (16 * 32 = 512 positions,
        // in position order), so the FIFO needs one round's left-overs + 512 slots only.
        const int ngrp = S > 512u ? 4 : 1;
        for (int gi = 0; gi < ngrp; gi++) {
            unsigned gbase = 0, gcnt = S;
            bool mine = true;
            if (ngrp != 1) {
                gbase = gi ? __builtin_amdgcn_readlane(incl, 16 * gi - 1) : 0u;
                gcnt = __builtin_amdgcn_readlane(incl, 16 * gi + 15) - gbase;
                mine = (lane >> 4) == gi;
            }
            if (tail + gcnt > (unsigned)QCAP) {    // no room behind the pending entries (fewer than one round of them): move them to the front
                const unsigned rem = tail - head;
                unsigned short v[NWALK];
#pragma unroll
                for (int w = 0; w < NWALK; w++) v[w] = (unsigned)lane + WAVE * w < rem ? q[head + WAVE * w + lane] : (unsigned short)0;
                wave_lds_sync();
#pragma unroll
                for (int w = 0; w < NWALK; w++) if ((unsigned)lane + WAVE * w < rem) q[WAVE * w + lane] = v[w];
                head = 0;
                tail = rem;
            }
            if (mine) {
                unsigned o = tail + (incl - cnt) - gbase;
                for (unsigned m = mask; m; m &= m - 1) {
                    const unsigned b = __ffs(m) - 1;
                    q[o++] = (unsigned short)((lpos + b) | (((dmask >> b) & 1u) << 15));
                }
            }
            tail += gcnt;
            wave_lds_sync();
            for (; head + RW <= tail; head += RW)
                TP_ROUND((wrun += roundN<W8, DIRECT, NWALK, FUSED, ROOT>(a, tile, s0, d1, R, T, q, head, RW, lane, stage, lim, tile_base, wrun)));
        }
    }
    // (the last, partial round runs where its entries lie: nothing is appended behind them any more; with two walks per
    // lane and at most 64 entries left, the one-walk form of the round does the same work in half the instructions -- the
    // sparse L2-table kernels are paced by their instruction stream, and their typical tile ends on such a round)
    if (tail > head) {
        if (NWALK == 2 && !DIRECT && tail - head <= (unsigned)WAVE)
            TP_ROUND((wrun += roundN<W8, DIRECT, 1, FUSED, ROOT>(a, tile, s0, d1, R, T, q, head, tail - head, lane, stage, lim, tile_base, wrun)));
        else
            TP_ROUND((wrun += roundN<W8, DIRECT, NWALK, FUSED, ROOT>(a, tile, s0, d1, R, T, q, head, tail - head, lane, stage, lim, tile_base, wrun)));
    }
#ifdef PFAC_TRACE_BUILD
    if (!DIRECT && a.dbg && blockIdx.x < 8 && lane == 0 && (threadIdx.x >> 6) == 0) {
        // compute wave 0 of the first 8 workgroups: accumulated over the launch (slot 31 of the block's first row)
        unsigned long long *acc = a.dbg + (size_t)blockIdx.x * 64 * 32 + 10;      // columns 10..14 of the block's row 0
        acc[0] += tp_round; acc[1] += tp_n; acc[2] += __builtin_amdgcn_s_memrealtime() - tp_t0; acc[3] += 1; acc[4] += tail - head;
    }
#endif
    return wrun;
}

// Staged words of one tile -> global memory, in order (packed format only: the staged word IS the record).  Four
// records (16 bytes) per lane per store, aligned to the record array; non-temporal stores -- the records are not read
// again by this launch, and written through they do not pile up dirty in L2 until the kernel ends (measured: the same at
// 1 GiB shards, +3.5 % at 4 GiB).
template <bool SPARSE>
__device__ __forceinline__ void copy_out(const ScanArgs &a, const unsigned *stage, unsigned cnt, unsigned long long base, int lane) {
    if (cnt == 0) return;
#ifdef PFAC_ABL_NOEMIT                         // ablation builds only: records never leave LDS
    return;
#endif
    if (a.rec_bytes == 2) {
        // automata with at most 16 final states: the record is the low half of the staged word; eight per 16-byte store
        unsigned short *out = static_cast<unsigned short *>(a.out);
        if (base + cnt > a.out_cap) {
            for (unsigned i = (unsigned)lane; i < cnt; i += WAVE)
                if (base + i < a.out_cap) out[base + i] = (unsigned short)stage[i];
            return;
        }
        unsigned head = (0u - (unsigned)base) & 7u;
        head = head < cnt ? head : cnt;
        if ((unsigned)lane < head) out[base + lane] = (unsigned short)stage[lane];
        unsigned i = head + 8u * (unsigned)lane;
        for (; i + 8u <= cnt; i += 8u * WAVE) {
            const u32x4 v = {(stage[i] & 0xFFFFu) | (stage[i + 1] << 16), (stage[i + 2] & 0xFFFFu) | (stage[i + 3] << 16),
                             (stage[i + 4] & 0xFFFFu) | (stage[i + 5] << 16), (stage[i + 6] & 0xFFFFu) | (stage[i + 7] << 16)};
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(out + base + i));
        }
        // the last 1..7 records: one lane each (not a loop in the one lane that stopped there: seven dependent LDS trips)
        const unsigned tail = head + ((cnt - head) & ~7u);
        if (tail + (unsigned)lane < cnt) out[base + tail + lane] = (unsigned short)stage[tail + lane];
        return;
    }
    if (SPARSE && cnt <= 2u * WAVE && base + cnt <= a.out_cap) {
        // a sparse tile (the L2-table workloads: tens of records): one record per lane per store, two independent trips
        // -- the aligned 16-byte form below pays three dependent LDS round trips (head, body, tail) for a few hundred bytes
        unsigned *out = static_cast<unsigned *>(a.out);
        const unsigned v0 = (unsigned)lane < cnt ? stage[lane] : 0u, v1 = (unsigned)lane + WAVE < cnt ? stage[lane + WAVE] : 0u;
        if ((unsigned)lane < cnt) __builtin_nontemporal_store(v0, out + base + lane);
        if ((unsigned)lane + WAVE < cnt) __builtin_nontemporal_store(v1, out + base + lane + WAVE);
        return;
    }
    unsigned *out = static_cast<unsigned *>(a.out);
    if (base + cnt > a.out_cap) {              // the record array ends inside this tile: word by word, checked
        for (unsigned i = (unsigned)lane; i < cnt; i += WAVE)
            if (base + i < a.out_cap) out[base + i] = stage[i];
        return;
    }
    unsigned head = (0u - (unsigned)base) & 3u;                // records up to the next 16-byte boundary of the array
    head = head < cnt ? head : cnt;
    if ((unsigned)lane < head) out[base + lane] = stage[lane];
    unsigned i = head + 4u * (unsigned)lane;
    for (; i + 4u <= cnt; i += 4u * WAVE) {
        const u32x4 v = {stage[i], stage[i + 1], stage[i + 2], stage[i + 3]};
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(out + base + i));
    }
    const unsigned tail = head + ((cnt - head) & ~3u);         // the last 1..3 records: one lane each
    if (tail + (unsigned)lane < cnt) out[base + tail + lane] = stage[tail + lane];
}

// ---------------------------------------------------------------------------
// Dense mode, second form (a dictionary on text: most offsets start a walk, every third goes beyond its second byte).
// tile_pass runs its walkers in rounds of 256, in lock step to the depth of the round's deepest walker -- on text a
// fifth of the lane-steps of such a round belong to a live walker -- because a round's records have to come out in
// (position, length) order, and its 8 KiB staging buffer per wave leaves room for 9 waves per CU.  Here the order is
// made afterwards, and the records wait in device memory, not in LDS (15 waves per CU):
//   * front end, D2_FB x 64 consecutive positions per trip, every lane busy: root table (final state of the first byte |
//     word offset of its dense row), column of the second byte -> dense row entry -> packed {r[], child mask}: three
//     dependent LDS trips; the final states of depth 1 and 2 are logged at once; the positions whose depth-2 state has an
//     edge on a byte congruent to the third one (child mask) go into a ring of pending walkers {position, state, fused
//     slot of the first gather, records so far};
//   * PFAC_D2_NS walker slots per lane; a slot whose walk has ended takes the next ring entry, so the gathers of a step
//     belong to live walkers only; a walker dies WITHOUT the gather that would say so when the child mask of its state
//     has no bit for the next byte; every final state is logged as {position, k-th record of the position, state};
//   * the log is the wave's own scratch in device memory, written 64 words at a time from an LDS staging area (a store of
//     a handful of lanes costs the texture path as much as a full one);
//   * once the tile is done the log is read back (into registers, where it fits): the records of every position are
//     counted (4-bit fields in LDS), the counts prefix-summed (per 8 positions + a nibble sum inside the word), and every
//     log entry is stored straight to its place in the heap: base + prefix[position] + k.
// Log full or more than 15 patterns starting at one offset: the tile is done again by tile_pass (returns ~0u), counted
// and then written directly (no staging buffer in this layout).  Needs: fused tables, packed dense rows (every depth-1
// state has a dense row, every depth-2 state an entry in d1.r2), final states below 2^16, 4-byte records.
#ifndef PFAC_D2_FB
#define PFAC_D2_FB 2
#endif
#ifndef PFAC_D2_NS
#define PFAC_D2_NS 2
#endif
constexpr int D2_FB = PFAC_D2_FB;              // front-end batches per trip (their LDS round trips overlap)
static_assert(D2_RING >= (D2_FB + 1) * WAVE, "ring: one trip's pushes on top of a batch of left-overs");
template <bool W8>
__device__ __forceinline__ unsigned dense2_tile(const ScanArgs &a, const unsigned char *tile, const unsigned *t0, const Dense1 &d1,
                                                unsigned char *aux, unsigned *logg, int lane, unsigned lim, unsigned own_end) {
    constexpr int NS = PFAC_D2_NS;             // walker slots per lane
    uint2 *ring = reinterpret_cast<uint2 *>(aux);
    const unsigned *t32 = reinterpret_cast<const unsigned *>(tile);
    const int sub = a.wbit - 8;
    const unsigned nfin = (unsigned)a.num_final, logcap = a.d2log_cap;
    // the log through a buffer descriptor: a scalar base, one shift per store, and a store past the end is dropped
    const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(logg, 0, (int)(logcap * 4u), 0x00020000);
    unsigned pos[NS], pn[NS], jn[NS];          // position; index of the byte the next gather consumes; records of the position so far
    int s[NS], idx[NS];                        // state; fused slot the next gather reads (r[row of state] + column of that byte)
    bool alive[NS];
#pragma unroll
    for (int w = 0; w < NS; w++) { pos[w] = pn[w] = jn[w] = 0u; s[w] = -1; idx[w] = 0; alive[w] = false; }
    unsigned P0 = 0, head = 0, fcount = 0, lc = 0;             // wave-uniform: next front-end position, ring, log fill
    unsigned jmax = 0;
    const unsigned no_root = 0xFFFFu | ((unsigned)(a.d1_rows * a.d1_stride) << 16);
    auto lane_rank = [&](unsigned long long b) -> unsigned {
        return __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
    };
    // Log words are collected in LDS (the bytes of the position counts, which only the scatter phase uses) and leave 64 at a
    // time: a store of a handful of lanes costs the texture path as much as a full one, and a tile logs ~180 handfuls
    unsigned *stg = reinterpret_cast<unsigned *>(aux + D2_CNT_OFF);
    constexpr unsigned STG = (unsigned)(WTILE / 2 / 4);        // words (a power of two)
    static_assert(STG >= (unsigned)(WAVE + 2 * D2_FB * WAVE) && STG >= (unsigned)(WAVE + PFAC_D2_NS * WAVE), "log staging: < 64 words pending + what a trip / a step adds");
    unsigned fl = 0;                                           // log words flushed so far (a multiple of 64 until the end)
    auto log_put = [&](bool f, unsigned word) {
        const unsigned long long b = __builtin_amdgcn_ballot_w64(f);
        if (b) {
            if (f) stg[(lc + lane_rank(b)) & (STG - 1u)] = word;
            lc += (unsigned)__popcll(b);
        }
    };
    auto log_flush = [&](bool all) {
        if (lc - fl >= (unsigned)WAVE || (all && lc != fl)) {
            wave_lds_sync();
            while (lc - fl >= (unsigned)WAVE) {
                __builtin_amdgcn_raw_buffer_store_b32(stg[(fl + (unsigned)lane) & (STG - 1u)], lrs, (int)((fl + (unsigned)lane) << 2), 0, 0);
                fl += WAVE;
            }
            if (all && fl + (unsigned)lane < lc)
                __builtin_amdgcn_raw_buffer_store_b32(stg[(fl + (unsigned)lane) & (STG - 1u)], lrs, (int)((fl + (unsigned)lane) << 2), 0, 0);
        }
    };
    auto slot = [&](int i) -> int4 {
        return *reinterpret_cast<const int4 *>(reinterpret_cast<const unsigned char *>(a.T4) + ((unsigned)i << 4));
    };
    auto slot_of = [&](int rnv, unsigned st, unsigned byte) -> int {       // fused slot of (state, byte), r[row of state] given
        return W8 ? rnv + (int)byte : rnv + (int)(((st & ((1u << sub) - 1u)) << 8) | byte);
    };
    // One trip of the front end: D2_FB x 64 positions from P0 on.  EDGE: the input's last tile (positions past n_owned start
    // nothing, bytes past the input are not read) -- everywhere else a position's first three bytes are inside the tile + halo
    // and those tests fall away.  A byte that starts no pattern has the no-edge row, a (first, second) byte pair that is no
    // prefix the no-child entry of d1.r2: "no second state" needs no test of its own.
    const bool edge = own_end < (unsigned)WTILE || lim < (unsigned)WTILE + 3u;
    const unsigned n2 = (unsigned)a.d1_n2;
    auto front = [&](auto edge_c) {
        constexpr bool EDGE = decltype(edge_c)::value;
        unsigned p[D2_FB], win[D2_FB], rt[D2_FB], col[D2_FB];
        int nx[D2_FB];
        int2 e2[D2_FB];
#pragma unroll
        for (int u = 0; u < D2_FB; u++) {
            p[u] = P0 + (unsigned)(u * WAVE + lane);
            const unsigned lo = t32[p[u] >> 2], hi = t32[(p[u] >> 2) + 1];
            win[u] = __builtin_amdgcn_alignbyte(hi, lo, p[u] & 3u);              // bytes p .. p+3
        }
#pragma unroll
        for (int u = 0; u < D2_FB; u++) {
            rt[u] = t0[win[u] & 0xFFu];
            col[u] = d1.colmap[(win[u] >> 8) & 0xFFu];
        }
#pragma unroll
        for (int u = 0; u < D2_FB; u++) {
            if (EDGE && p[u] >= own_end) rt[u] = no_root;
            nx[u] = d1.rows[(rt[u] >> 16) + col[u]];                             // depth-2 state | index of its r2 entry << 20, or -1
        }
#pragma unroll
        for (int u = 0; u < D2_FB; u++) {
            if (EDGE && p[u] + 1u >= lim) nx[u] = -1;
            const unsigned ri = (unsigned)(nx[u] >> D1_STATE_BITS);             // (-1: beyond every index)
            e2[u] = d1.r2[ri < n2 ? ri : n2];                                    // {r[] of the depth-2 state, its child mask}; entry n2: {0, 0}
        }
#pragma unroll
        for (int u = 0; u < D2_FB; u++) {
            const unsigned s1 = rt[u] & 0xFFFFu, s2 = (unsigned)nx[u] & ((1u << D1_STATE_BITS) - 1u);
            const bool fin1 = s1 != 0xFFFFu, fin2 = s2 < nfin;                   // (no second state: s2 = 2^20 - 1, above every final state)
            const unsigned b2 = (win[u] >> 16) & 0xFFu;
#ifdef PFAC_ABL_D2NOWALK                       // ablation builds only: nothing goes beyond its second byte (records missing)
            const bool more = false && b2;
#else
            const bool more = (((unsigned)e2[u].y >> (b2 & 31u)) & 1u) != 0u && (!EDGE || p[u] + 2u < lim);
#endif
            log_put(fin1, p[u] | (s1 << 16));
            log_put(fin2, p[u] | (fin1 ? 1u << 12 : 0u) | ((unsigned)nx[u] << 16));
            const unsigned long long mb = __builtin_amdgcn_ballot_w64(more);
            if (mb) {
                const unsigned n12 = (fin1 ? 1u : 0u) + (fin2 ? 1u : 0u);
                if (more) ring[(head + fcount + lane_rank(mb)) & (unsigned)(D2_RING - 1)] =
                    make_uint2(p[u] | ((unsigned)nx[u] << 12), (unsigned)slot_of(e2[u].x, s2, b2) | (n12 << 28));
                fcount += (unsigned)__popcll(mb);
            }
        }
        P0 += D2_FB * WAVE;
    };
#ifdef PFAC_TRACE_BUILD                        // diagnostic build only: where a tile's time goes (compute wave 0 of the first 8 workgroups)
    unsigned long long tq_front = 0, tq_iter = 0, tq_n = 0;
    const unsigned long long tq_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (;;) {
        unsigned long long dm[NS];
        unsigned nd = 0;
#pragma unroll
        for (int w = 0; w < NS; w++) { dm[w] = __builtin_amdgcn_ballot_w64(!alive[w]); nd += (unsigned)__popcll(dm[w]); }
#ifdef PFAC_TRACE_BUILD
        const unsigned long long tq_a = __builtin_amdgcn_s_memrealtime();
#endif
        // ---- front end: as many trips as the free slots ask for
        while (fcount < nd && P0 < own_end && fcount <= (unsigned)(D2_RING - D2_FB * WAVE)) {
            if (edge) front(std::true_type{}); else front(std::false_type{});
            log_flush(false);
        }
#ifdef PFAC_TRACE_BUILD
        tq_front += __builtin_amdgcn_s_memrealtime() - tq_a;
        tq_n++;
#endif
        if (nd == (unsigned)(NS * WAVE) && fcount == 0u) break;        // no walker left, none pending, every position seen
        wave_lds_sync();
        // ---- free slots take the pending walkers
#pragma unroll
        for (int w = 0; w < NS; w++) {
            if (fcount && dm[w]) {
                const unsigned free_n = (unsigned)__popcll(dm[w]);
                const unsigned take = free_n < fcount ? free_n : fcount;
                const unsigned rk = lane_rank(dm[w]);
                if (!alive[w] && rk < take) {
                    const uint2 e = ring[(head + rk) & (unsigned)(D2_RING - 1)];
                    pos[w] = e.x & 0xFFFu;
                    s[w] = (int)(e.x >> 12);
                    idx[w] = (int)(e.y & 0x0FFFFFFFu);
                    jn[w] = e.y >> 28;
                    pn[w] = pos[w] + 2u;
                    alive[w] = true;
                }
                head += take;
                fcount -= take;
            }
        }
        // ---- one transition of every live walker
        int4 e4[NS];
        unsigned nb[NS];
#pragma unroll
        for (int w = 0; w < NS; w++) {
            e4[w] = make_int4(-1, -1, 0, 0);
            if (alive[w]) e4[w] = slot(idx[w]);
        }
#pragma unroll
        for (int w = 0; w < NS; w++) nb[w] = tile[alive[w] ? pn[w] + 1u : 0u];     // the byte after (at most one past lim: not used then)
#pragma unroll
        for (int w = 0; w < NS; w++) {
            const bool ok = alive[w] && e4[w].x == (W8 ? s[w] : (s[w] >> sub));
            const int sn = e4[w].y;
            const bool fin = ok && (unsigned)sn < nfin;
            log_put(fin, pos[w] | ((jn[w] & 15u) << 12) | ((unsigned)sn << 16));
            jn[w] += fin ? 1u : 0u;
            jmax |= jn[w];                     // (bit 4 or above: a 16th record at one position)
            pn[w] += 1u;
            alive[w] = ok && pn[w] < lim && (((unsigned)e4[w].w >> (nb[w] & 31u)) & 1u) != 0u;
            s[w] = sn;
            idx[w] = slot_of(e4[w].z, (unsigned)sn, nb[w]);
        }
        log_flush(false);
    }
    log_flush(true);
#ifdef PFAC_TRACE_BUILD
    if (a.dbg && blockIdx.x < 8 && lane == 0 && (threadIdx.x >> 6) == 0) {
        unsigned long long *acc = a.dbg + (size_t)blockIdx.x * 64 * 32 + 10;      // columns 10..14 of the block's row 0
        const unsigned long long tq_all = __builtin_amdgcn_s_memrealtime() - tq_t0;
        acc[0] += tq_front; acc[1] += tq_n; acc[2] += tq_all; acc[3] += 1; acc[4] += lc;
    }
    (void)tq_iter;
#endif
    if (__any(jmax > 15u) || lc > logcap) return ~0u;
    return lc;
}

// The tile's log -> its place in the record heap (4-byte records), in (position, length) order: the records of every
// position are counted (4-bit fields), the counts prefix-summed, and every log entry goes to base + prefix[position] + k.
__device__ __forceinline__ void dense2_scatter(const ScanArgs &a, unsigned char *aux, const unsigned *logg, unsigned cnt,
                                               unsigned long long base, int lane) {
    unsigned *cntw = reinterpret_cast<unsigned *>(aux + D2_CNT_OFF);
    unsigned *pref2 = reinterpret_cast<unsigned *>(aux + D2_PREF_OFF);            // u16 pairs
    const unsigned short *pref = reinterpret_cast<const unsigned short *>(aux + D2_PREF_OFF);
    auto nibsum = [](unsigned x) -> unsigned { return (((x & 0x0F0F0F0Fu) + ((x >> 4) & 0x0F0F0F0Fu)) * 0x01010101u) >> 24; };
    constexpr int UN = 4;                      // log words per lane per group
    constexpr int ER = 32;                     // a tile of up to ER x 64 records keeps its log words in registers between the two passes
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    reinterpret_cast<u32x4 *>(cntw)[lane] = zero4;
    reinterpret_cast<u32x4 *>(cntw)[lane + WAVE] = zero4;
    // (the log was written by other lanes of THIS wave: the workgroup-scope fence waits for their stores, and a CU's L1 is
    // coherent with the stores of its own waves -- an agent-scope release would write the whole L2 back, per tile)
    wave_lds_sync();
    // the tile's piece of the heap through a buffer descriptor: a scalar base, and what lies past the end of the record
    // array is dropped by the bounds check
    const unsigned long long room = base < a.out_cap ? a.out_cap - base : 0ull;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(static_cast<unsigned *>(a.out) + base, 0,
                                                                         (int)((room < (unsigned long long)cnt ? (unsigned)room : cnt) * 4u), 0x00020000);
    auto count_one = [&](unsigned e) { atomicAdd(&cntw[(e & 0xFFFu) >> 3], 1u << ((e & 7u) * 4u)); };
    auto prefix = [&]() {
        wave_lds_sync();
        const u32x4 x0 = reinterpret_cast<const u32x4 *>(cntw)[2 * lane], x1 = reinterpret_cast<const u32x4 *>(cntw)[2 * lane + 1];
        unsigned sm[8];
#pragma unroll
        for (int k = 0; k < 4; k++) { sm[k] = nibsum(x0[k]); sm[4 + k] = nibsum(x1[k]); }
        unsigned tot = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) tot += sm[k];
        unsigned run = wave_incl_scan(tot) - tot;
        u32x4 pw;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned lo = run; run += sm[2 * k];
            const unsigned hi = run; run += sm[2 * k + 1];
            pw[k] = lo | (hi << 16);
        }
        reinterpret_cast<u32x4 *>(pref2)[lane] = pw;
        wave_lds_sync();
    };
    auto place_one = [&](unsigned e) {
        const unsigned p = e & 0xFFFu, j = (e >> 12) & 15u, st = e >> 16;
        const unsigned x = cntw[p >> 3];
        const unsigned dest = (unsigned)pref[p >> 3] + nibsum(x & ((1u << ((p & 7u) * 4u)) - 1u)) + j;
        __builtin_amdgcn_raw_buffer_store_b32(p | (st << 12), ors, (int)(dest << 2), 0, 0);
    };
    if (cnt <= (unsigned)(ER * WAVE)) {
        const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(logg), 0, (int)(cnt * 4u), 0x00020000);
        // the usual tile: every load of the log goes out before the first word is looked at -- the head of the log has left
        // the L2 by now (one memory latency per tile, not one per group of loads) -- and the words stay in registers for
        // the second pass
        unsigned e[ER];
#pragma unroll
        for (int g = 0; g < ER / UN; g++) {
            if ((unsigned)(g * UN * WAVE) < cnt) {
#pragma unroll
                for (int u = 0; u < UN; u++)      // (one lane offset, the group in the scalar offset; past the end of the log: 0)
                    e[g * UN + u] = __builtin_amdgcn_raw_buffer_load_b32(lrs, lane * 4, (g * UN + u) * WAVE * 4, 0);
            }
        }
#pragma unroll
        for (int g = 0; g < ER / UN; g++) {
            if ((unsigned)(g * UN * WAVE) < cnt) {
#pragma unroll
                for (int u = 0; u < UN; u++)
                    if ((unsigned)((g * UN + u) * WAVE + lane) < cnt) count_one(e[g * UN + u]);
            }
        }
        prefix();
#ifdef PFAC_ABL_D2NOSCATTER                    // ablation builds only: the records never reach the heap
        cnt = 0;
#endif
#pragma unroll
        for (int g = 0; g < ER / UN; g++) {
            if ((unsigned)(g * UN * WAVE) < cnt) {
#pragma unroll
                for (int u = 0; u < UN; u++)
                    if ((unsigned)((g * UN + u) * WAVE + lane) < cnt) place_one(e[g * UN + u]);
            }
        }
        return;
    }
    for (unsigned i0 = 0; i0 < cnt; i0 += UN * WAVE) {
        unsigned e[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const unsigned i = i0 + (unsigned)(u * WAVE + lane);
            e[u] = i < cnt ? logg[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < UN; u++)
            if (i0 + (unsigned)(u * WAVE + lane) < cnt) count_one(e[u]);
    }
    prefix();
#ifdef PFAC_ABL_D2NOSCATTER
    cnt = 0;
#endif
    for (unsigned i0 = 0; i0 < cnt; i0 += UN * WAVE) {
        unsigned e[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const unsigned i = i0 + (unsigned)(u * WAVE + lane);
            e[u] = i < cnt ? logg[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < UN; u++)
            if (i0 + (unsigned)(u * WAVE + lane) < cnt) place_one(e[u]);
    }
}

// Root test: 16-bit mask of the lane's 16 bytes that have an edge out of the root.
//   ROOT == 1: exactly one such byte value -> exact SWAR compare, flags gathered with v_dot4
//   ROOT == 0: one LDS lookup per byte in pre-shifted flag tables (byte k of a dword looks into table k).  The same lookup
//              answers a second question for free: bits 16..31 of the result = which of the 16 bytes can be the
//              SECOND byte of a pattern at all (the cheap half of the level-2 filter).
template <int ROOT>
__device__ __forceinline__ unsigned root_mask(const u32x4 w, const unsigned char *ftab, unsigned root_x4) {
    if (ROOT == 1) {
        unsigned nm[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const unsigned x = w[i] ^ root_x4;                            // zero byte <=> match
            nm[i] = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;  // 0x80 where the byte is NOT a match
        }
        unsigned lo = __builtin_amdgcn_udot4(nm[0], 0x08040201u, 0u, false);
        lo = __builtin_amdgcn_udot4(nm[1], 0x80402010u, lo, false);       // = 128 * (not-match bits 0..7)
        unsigned hi = __builtin_amdgcn_udot4(nm[2], 0x08040201u, 0u, false);
        hi = __builtin_amdgcn_udot4(nm[3], 0x80402010u, hi, false);
        return ~((lo >> 7) | (hi << 1)) & 0xFFFFu;
    } else {
        // four byte-wide tables, one per byte of a dword: table k holds  root flag << k | second-byte flag << (k + 4).
        // 256 bytes = 64 LDS words = one word per bank, so a wave's 64 random lookups never conflict (the 16-bit
        // tables, two words per bank, spent half their cycles on conflicts)
        const unsigned char *t8 = ftab;
        unsigned p = 0;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const unsigned d = w[g];
            const unsigned v = (unsigned)t8[d & 0xFFu] | (unsigned)t8[256 + ((d >> 8) & 0xFFu)] |
                               (unsigned)t8[512 + ((d >> 16) & 0xFFu)] | (unsigned)t8[768 + (d >> 24)];
            p |= v << (8 * g);
        }
        // p: byte g = dword g's flags, low nibble root, high nibble second-byte -> two 16-bit masks
        unsigned x = p & 0x0F0F0F0Fu, y = (p >> 4) & 0x0F0F0F0Fu;
        x = (x | (x >> 4)) & 0x00FF00FFu; y = (y | (y >> 4)) & 0x00FF00FFu;
        x = (x | (x >> 8)) & 0xFFFFu; y = (y | (y >> 8)) & 0xFFFFu;
        return x | (y << 16);
    }
}

// Bytes equal to x (replicated x4) among the lane's 32: bit b <=> byte b.
__device__ __forceinline__ unsigned eq_mask32(const u32x4 lo16, const u32x4 hi16, unsigned x4) {
    return root_mask<1>(lo16, nullptr, x4) | (root_mask<1>(hi16, nullptr, x4) << 16);
}

#ifdef PFAC_TRACE_BUILD
#define PFAC_STAMP(cond, slot) do { if (cond) tr[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PFAC_STAMP(cond, slot) do { } while (0)
#endif

// ---------------------------------------------------------------------------
// The scan kernel.  Workgroups share the read-only tables staged in LDS once; after that there is
// no workgroup barrier: compute waves pipeline  [loads of round r+1 in flight | scan round r |
// emit round r-2]  and meet the coordinator only through the LDS rings above.
#ifndef PFAC_SPARSE_FUSED_NW
#define PFAC_SPARSE_FUSED_NW 2
#endif
constexpr int MAX_WAVES_NW4 = 10;          // four walks per lane, staged in LDS: dense mode's buffers leave room for 9-10 waves per workgroup

template <bool TLDS, bool W8, int ROOT, bool FUSED, int NW, int NB>
__device__ __forceinline__ void scan_body(const ScanArgs &a, unsigned char *smem, const ErrCh &err) {
    unsigned *hdr = reinterpret_cast<unsigned *>(smem + SH_HDR);
    int *s0 = reinterpret_cast<int *>(smem + SH_S0);
    unsigned char *ftab = smem + SH_FTAB;
    unsigned char *finl = smem + SH_FIN;
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keep it in an SGPR
    const int nc = (int)(blockDim.x >> 6) - 1;             // compute waves; wave nc coordinates

    // ---- the control header of the slot's NEXT scan (ticket counters, flags, heap cursor of its other buffer) is zeroed here,
    // spread over the whole grid: the next launch on the stream starts after this one has ended
    for (unsigned i = blockIdx.x * blockDim.x + tid; i < a.zero_vec; i += gridDim.x * blockDim.x)
        a.zero_next[i] = make_uint4(0u, 0u, 0u, 0u);
    // ---- once per workgroup: rings cleared, root row, root flag tables, (small) PHF tables -> LDS
    for (int i = tid; i < H_WORDS; i += blockDim.x) hdr[i] = 0;
    for (int i = tid; i < 256; i += blockDim.x) {
        const int v = a.s0[i];
        const bool sec2 = a.sec2[i] != 0;
        s0[i] = v;
        finl[i] = (unsigned)v < (unsigned)a.num_final ? (unsigned char)1 : (unsigned char)0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            ftab[k * 256 + i] = (unsigned char)((v >= 0 ? 1u << k : 0u) | (sec2 ? 1u << (k + 4) : 0u));
    }
    unsigned char *d1idx_l = smem + SH_D1IDX;
    int *d1_l = reinterpret_cast<int *>(smem + SH_D1);
    unsigned char *colmap_l = smem + SH_COLMAP;
    if (a.d1_rows > 0) {
        for (int i = tid; i < 256; i += blockDim.x) { d1idx_l[i] = a.d1idx[i]; colmap_l[i] = a.d1_colmap[i]; }
        // (only the columns of bytes some pattern has second; the last column of a row: no edge)
        for (int i = tid; i < a.d1_rows * a.d1_stride; i += blockDim.x) {
            const int row = i / a.d1_stride, c = i - row * a.d1_stride;
            d1_l[i] = c < a.d1_ncols ? a.d1[row * 256 + a.d1_colbyte[c]] : -1;
        }
    } else {
        for (int i = tid; i < 256; i += blockDim.x) { d1idx_l[i] = 0; colmap_l[i] = 0; }
    }
    if (a.d1_rows > 0)                         // (one more row, without edges: where dense2_tile sends the bytes that start no pattern)
        for (int i = tid; i < a.d1_stride; i += blockDim.x) d1_l[a.d1_rows * a.d1_stride + i] = -1;
    // root table of dense2_tile: final state of the byte's depth-1 state (0xFFFF: not final) | word offset of its dense row << 16
    const unsigned *t0_l = reinterpret_cast<const unsigned *>(smem + a.sh_t0);
    if (FUSED && NW == 4 && a.sh_t0)
        for (int i = tid; i < 256; i += blockDim.x) {
            const int v = a.s0[i];
            reinterpret_cast<unsigned *>(smem + a.sh_t0)[i] = v >= 0 ? (((unsigned)v < (unsigned)a.num_final ? (unsigned)v : 0xFFFFu) | ((unsigned)(a.d1idx[i] * a.d1_stride) << 16))
                                                                     : (0xFFFFu | ((unsigned)(a.d1_rows * a.d1_stride) << 16));
        }
    int2 *d1r2_l = reinterpret_cast<int2 *>(smem + SH_D1 + a.d1_lds_bytes);
    if (FUSED && a.d1_n2 > 0)
        for (int i = tid; i <= a.d1_n2; i += blockDim.x) {     // (one more entry, {0, no child}: where dense2_tile sends "no second state")
            const int2 e = i < a.d1_n2 ? a.d1r2[i] : make_int2(-a.rn_bias, 0);
            d1r2_l[i] = make_int2(e.x + a.rn_bias, e.y);
        }
    // FUSED without dense rows: r[] of the depth-1 states by root byte, in the (unused) dense-row region
    int *s0r_l = reinterpret_cast<int *>(smem + SH_D1);
    const bool have_s0r = FUSED && a.d1_rows == 0;
    if (have_s0r)
        for (int i = tid; i < 256; i += blockDim.x) {
            const int v = a.s0[i];
            s0r_l[i] = v >= 0 ? a.r[v >> (a.wbit - 8)] + a.rn_bias : 0;
        }
    const Dense1 d1 = {colmap_l, d1idx_l, d1_l, a.d1_rows > 0, (FUSED && a.d1_n2 > 0) ? d1r2_l : nullptr, have_s0r ? s0r_l : nullptr};
    const int sh_tab = SH_D1 + a.d1_lds_bytes;       // (the packed rows' r[] and LDS tables never coexist)
    const int *R = a.r;
    const int2 *T = a.T;
    if (TLDS) {
        int *lr = reinterpret_cast<int *>(smem + sh_tab);
        int2 *lt = reinterpret_cast<int2 *>(smem + sh_tab + ((a.r_words * 4 + 15) & ~15));
        for (int i = tid; i < a.r_words; i += blockDim.x) lr[i] = a.r[i];
        for (int i = tid; i < a.t_entries; i += blockDim.x) lt[i] = a.T[i];
        R = lr;
        T = lt;
    }
    // level-2 filter, lookup form: the 2-byte-prefix bitmap (all 256 rows, or the root byte's row when ROOT == 1)
    const unsigned char *bm2l = smem + a.sh_bm2;
    if (a.l2f_mode == 2) {
        const unsigned *src = reinterpret_cast<const unsigned *>(a.bm2 + (ROOT == 1 ? (a.root_byte & 0xFFu) * 32u : 0u));
        unsigned *dst = reinterpret_cast<unsigned *>(smem + a.sh_bm2);
        for (int i = tid; i < a.bm2_rows * 8; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();                           // the only workgroup barrier
#ifdef PFAC_TRACE_BUILD
    if (a.dbg && blockIdx.x < 8 && threadIdx.x == 0) a.dbg[((size_t)blockIdx.x * 64 + 1) * 32 + 15] = __builtin_amdgcn_s_memrealtime();
#endif

#ifdef PFAC_ABL_NOCOORD                        // ablation builds only: static tiles, no coordinator, counts dropped
    if (wave == nc) return;
#endif
    if (wave == nc) {
        // ================= coordinator =================
        __builtin_amdgcn_s_setprio(3);         // tiny, latency-critical instruction stream
        // batch tickets: the atomic is ISSUED one iteration before its result is needed, so its ~1.3 us
        // round trip never blocks the coordinator
#ifdef PFAC_ABL_STATIC                         // ablation builds only (tools/abn.sh): batches dealt round-robin, no atomic
        unsigned abl_seq = 0;
        auto ticket = [&]() -> unsigned { return (abl_seq++) * gridDim.x + blockIdx.x; };
#else
        auto ticket = [&]() -> unsigned {
            unsigned g = 0;
            if (lane == 0) g = atomicAdd(&a.ctl[(blockIdx.x % a.ticket_ways) * 64u], 1u) * a.ticket_ways + blockIdx.x % a.ticket_ways;
            return g;                          // valid in lane 0 (not waited for here)
        };
#endif
        auto publish_batch = [&](unsigned r, unsigned g_lane0) -> unsigned {   // ring entry of round r; returns batch id
            if (lane == 0) {
                lds_store(&hdr[H_ARRIVED + (r & 7)], 0u);
                lds_store(&hdr[H_BATCH + (r & 7)], g_lane0);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                lds_store(&hdr[H_EPOCH + (r & 7)], r + 1);
            }
            return __builtin_amdgcn_readfirstlane(g_lane0);
        };
        unsigned g[AHEAD];                     // batch ids of rounds r .. r+AHEAD-1 (published)
#pragma unroll
        for (int k = 0; k < AHEAD; k++) g[k] = publish_batch((unsigned)k, ticket());
        unsigned t_pending = ticket();         // for round AHEAD, published at the top of iteration 0
        // the heap: this workgroup's current chunk [ch_base, ch_base + ch_size), ch_used records of it taken, and the
        // chunk on order (the atomic was issued when the current one came into use; its value sits in lane 0)
        unsigned long long *cursor = reinterpret_cast<unsigned long long *>(a.ctl + CTL_CURSOR);
        auto order_chunk = [&]() -> unsigned long long {
            unsigned long long v = 0;
            if (lane == 0 && a.chunk) v = __hip_atomic_fetch_add(cursor, (unsigned long long)a.chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return v;                          // valid in lane 0 (not waited for here)
        };
        unsigned long long ch_base = 0, spare = order_chunk(), local_total = 0;
        unsigned ch_size = 0, ch_used = 0;
        for (unsigned r = 0;; r++) {
            const unsigned g_cur = g[0];
            const unsigned long long first = (unsigned long long)g_cur * (unsigned)nc;
            if (first >= a.n_tiles) break;     // batch ids only grow: nothing left for this workgroup
            const unsigned g_new = publish_batch(r + AHEAD, t_pending);   // AHEAD rounds ahead of this iteration
            t_pending = ticket();                           // for round r+AHEAD+1
#ifdef PFAC_TRACE_BUILD
            const bool trace = a.dbg && blockIdx.x < 8 && r < 64 && lane == 0;
            unsigned long long *tr = a.dbg + ((size_t)blockIdx.x * 64 + (r & 63)) * 32;
#endif
            PFAC_STAMP(trace, 0);
            // ---- round r: wait for the counts of its tiles
            const unsigned long long left = a.n_tiles - first;
            const unsigned n_valid = left < (unsigned long long)nc ? (unsigned)left : (unsigned)nc;
            if (!lds_wait_eq(&hdr[H_ARRIVED + (r & 7)], n_valid, err, 4u)) break;
            PFAC_STAMP(trace, 1);
            // (bit 31 of a posted count: that tile is emitted at once and has taken its own space from the heap cursor)
            const unsigned c_raw = (unsigned)lane < n_valid ? hdr[H_CNT + (r & 7) * 16 + lane] : 0u;
            const unsigned c_all = c_raw & 0x7FFFFFFFu;
            const bool self_placed = (c_raw >> 31) != 0u;
            const unsigned c = self_placed ? 0u : c_all;
            const unsigned incl = wave_incl_scan(c);        // 15 tile counts of < 2^22 each
            const unsigned tot = bcast_last(incl);
            const unsigned excl = incl - c;
            const unsigned tot_all = __any(self_placed) ? bcast_last(wave_incl_scan(c_all)) : tot;
            // ---- place the tiles (lane c: the tile of compute wave c)
            unsigned long long wb;
            if (tot == 0) {
                wb = ch_base + ch_used;                     // nothing to place (any valid index will do)
            } else if (tot > a.chunk) {
                // larger than a chunk (dense matches, or a small record array): an allocation of exactly this size
                unsigned long long v = 0;
                if (lane == 0) v = __hip_atomic_fetch_add(cursor, (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
                wb = (((unsigned long long)hi << 32) | lo) + excl;
            } else if (ch_used + tot <= ch_size) {
                wb = ch_base + ch_used + excl;
                ch_used += tot;
            } else {
                // the batch continues in the spare chunk from the first tile that does not fit into this one
                const unsigned long long fits = __ballot(ch_used + incl <= ch_size);    // a prefix of the lanes
                const unsigned k = (unsigned)__popcll(fits);                            // first lane that moves (0..nc-1)
                const unsigned off = __builtin_amdgcn_readlane(excl, k);
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)spare), hi = __builtin_amdgcn_readfirstlane((unsigned)(spare >> 32));
                const unsigned long long nb = ((unsigned long long)hi << 32) | lo;
                wb = (unsigned)lane < k ? ch_base + ch_used + excl : nb + (excl - off);
                ch_base = nb;
                ch_size = a.chunk;
                ch_used = tot - off;
                spare = order_chunk();
            }
            local_total += tot_all;
            const bool mute = (a.fault & 1u) && blockIdx.x == 1 && r == 1;   // test knob: the bases of this round never come
            if (lane < nc) {
                hdr[H_WBASE + (r & 7) * 32 + lane * 2] = (unsigned)wb;
                hdr[H_WBASE + (r & 7) * 32 + lane * 2 + 1] = (unsigned)(wb >> 32);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0 && !mute) lds_store(&hdr[H_READY + (r & 7)], r + 1);
            if ((unsigned)lane < n_valid && !self_placed) a.tile_index[first + (unsigned)lane] = wb | ((unsigned long long)c << TIX_CNT_SHIFT);
#ifdef PFAC_TRACE_BUILD
            if (trace) { tr[2] = __builtin_amdgcn_s_memrealtime(); tr[3] = g_cur; }
#endif
#pragma unroll
            for (int k = 0; k + 1 < AHEAD; k++) g[k] = g[k + 1];
            g[AHEAD - 1] = g_new;
        }
        // every count of this workgroup has been posted by now: matches, and how many tiles were denser than the sparse
        // staging capacity (the host switches staging mode on it)
        if (lane == 0) {
            if (local_total) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(a.ctl + CTL_TOTAL), local_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned ovf = lds_load(&hdr[H_OVF]);
            if (ovf) __hip_atomic_fetch_add(&a.ctl[CTL_OVF], ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned ovf2 = lds_load(&hdr[H_OVF2]);
            if (ovf2) __hip_atomic_fetch_add(&a.ctl[CTL_OVF2], ovf2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }

    // ================= compute waves =================
    unsigned char *tile = smem + a.shared_bytes + wave * a.pw_bytes;
    unsigned short *q = reinterpret_cast<unsigned short *>(tile + WTILE + a.halo);
    const bool d2 = NW == 4 && FUSED && a.dense2 != 0;         // dense mode, second form (dense2_tile); its LDS carve differs
    unsigned char *aux = tile + WTILE + a.halo;
    unsigned *stage0 = reinterpret_cast<unsigned *>(aux + QCAP * 2);        // (d2: never written, stage_cap is 0)
    unsigned *d2log = d2 ? a.d2log + ((size_t)blockIdx.x * (unsigned)nc + (unsigned)wave) * a.d2log_cap : nullptr;
    const bool root_final = ROOT == 1 && (unsigned)a.root_state < (unsigned)a.num_final;

    // prefetch registers: the wave's 4 KiB + halo, LOAD_DEPTH tiles ahead of the one being scanned (set A / set B)
    u32x4 wA[SUBS], wB[SUBS];
    u32x4 hA = {0u, 0u, 0u, 0u}, hB = {0u, 0u, 0u, 0u};
    auto issue_loads = [&](unsigned long long tt, u32x4 (&w)[SUBS], u32x4 &hw) {
        const unsigned long long tb = tt * WTILE;
        const unsigned long long remain = a.n_avail - tb;
        const unsigned lm = remain < (unsigned long long)(WTILE + a.halo) ? (unsigned)remain : (unsigned)(WTILE + a.halo);
        // the descriptor covers whole 16-B units only: every dword of a load is fully inside or reads as 0
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned char *>(a.in + tb), 0, (int)(lm & ~15u), 0x00020000);
#pragma unroll
        for (int j = 0; j < SUBS; j++) w[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, j * SUB + lane * 16, 0, LOAD_AUX);
        if (lane * 16 < a.halo) hw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, WTILE + lane * 16, 0, 0);
    };
    // this wave's tile of round rr; false: there is none (the input is used up) or the ring timed out.
    // Tile numbers are wave-uniform: as scalars they make the buffer descriptor of the tile loads a scalar too
    // (a descriptor in VGPRs costs a readfirstlane "waterfall" loop around every load)
    auto probe = [&](unsigned rr, unsigned long long &tt) -> bool {
#ifdef PFAC_ABL_NOCOORD
        tt = ((unsigned long long)rr * gridDim.x + blockIdx.x) * (unsigned)nc + (unsigned)wave;
        return tt < a.n_tiles;
#endif
        if (!lds_wait_eq(&hdr[H_EPOCH + (rr & 7)], rr + 1, err, 16u)) return false;
        tt = (unsigned long long)__builtin_amdgcn_readfirstlane(hdr[H_BATCH + (rr & 7)]) * (unsigned)nc + (unsigned)wave;
        return tt < a.n_tiles;
    };
    // first record index of this wave's tile of round rr (computed by the coordinator)
    auto record_base = [&](unsigned rr, unsigned long long &base) -> bool {
        if (!lds_wait_eq(&hdr[H_READY + (rr & 7)], rr + 1, err, 8u)) return false;
        // (every lane reads the same words: tell the compiler, so the record addresses get a scalar base)
        const unsigned blo = __builtin_amdgcn_readfirstlane(hdr[H_WBASE + (rr & 7) * 32 + wave * 2]);
        const unsigned bhi = __builtin_amdgcn_readfirstlane(hdr[H_WBASE + (rr & 7) * 32 + wave * 2 + 1]);
        base = ((unsigned long long)bhi << 32) | blo;
        return true;
    };

    unsigned r = 0;
    unsigned long long t = 0, t_n1 = 0;        // tiles of rounds r and (LOAD_DEPTH == 2) r+1
    bool have_n1 = false;
    if (!probe(0, t)) return;
    issue_loads(t, wA, hA);
    if (LOAD_DEPTH == 2) {
        have_n1 = probe(1, t_n1);
        if (have_n1) issue_loads(t_n1, wB, hB);
    }

    // pend_have[k]: the tile scanned k+1 rounds ago still sits in its staging buffer (pend_cnt[k] records)
    // NB == 3: the three-buffer kernels (a.nbuf == 3); NB == 2: two buffers, or one (dense mode)
    constexpr int LAG_MAX = NB - 1;
    const unsigned nbuf = NB == 3 ? 3u : a.nbuf;
    const unsigned lag = NB == 3 ? 2u : a.nbuf - 1u;   // rounds between a tile's scan and its emission (0: dense mode)
    bool pend_have[LAG_MAX];
    unsigned pend_cnt[LAG_MAX], buf = 0;       // buf: staging buffer of the current round, (r % nbuf)
#pragma unroll
    for (int k = 0; k < LAG_MAX; k++) { pend_have[k] = false; pend_cnt[k] = 0; }

    // One round: w / hw hold this round's tile; they are refilled with the tile LOAD_DEPTH rounds ahead as soon as
    // their bytes sit in LDS.  Returns false after the wave's last tile.
    auto round_body = [&](u32x4 (&w)[SUBS], u32x4 &hw) -> bool {
        const unsigned long long tile_base = t * WTILE;
        const unsigned long long remain = a.n_avail - tile_base;           // > 0
        const unsigned lim = remain < (unsigned long long)(WTILE + a.halo) ? (unsigned)remain : (unsigned)(WTILE + a.halo);

#ifdef PFAC_TRACE_BUILD
        const bool trace = a.dbg && blockIdx.x < 8 && r < 64 && lane == 0 && wave == 0;
        unsigned long long *tr = a.dbg + ((size_t)blockIdx.x * 64 + (r & 63)) * 32;
#endif
        PFAC_STAMP(trace, 4);
        // ---- registers -> LDS (tile + halo), then start the loads of the tile LOAD_DEPTH rounds ahead right away
#ifndef PFAC_ABL_NOLDSCOPY                     // ablation builds only: the tile never reaches LDS (wrong masks)
#pragma unroll
        for (int j = 0; j < SUBS; j++) *reinterpret_cast<u32x4 *>(tile + j * SUB + lane * 16) = w[j];
        if (lane * 16 < a.halo) *reinterpret_cast<u32x4 *>(tile + WTILE + lane * 16) = hw;
#else
        asm volatile("" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(hw));
#endif
        if (lim & 15u) {                       // ragged end of the input (last tile only): patch the tail bytes
            wave_lds_sync();
            if (lane < (int)(lim & 15u)) tile[(lim & ~15u) + lane] = a.in[tile_base + (lim & ~15u) + lane];
        }
        wave_lds_sync();
        PFAC_STAMP(trace, 5);
        unsigned long long t_far = 0;
        const bool more_far = (LOAD_DEPTH == 1 || have_n1) && probe(r + LOAD_DEPTH, t_far);
        if (more_far) issue_loads(t_far, w, hw);
        PFAC_STAMP(trace && r > 0, 10);
        // ---- emit the tile of `lag` rounds ago: its bases came when that round's last count was in (a tile without
        // records has nothing to wait for: the coordinator stores the tile index).  Its buffer is the one after the
        // current round's, cyclically.
        auto emit_pending = [&](bool have, unsigned n_rec) {
            if (have && n_rec != 0) {
                unsigned long long base = 0;
                const bool okb = record_base(r - lag, base);
                PFAC_STAMP(trace, 9);
                if (okb) copy_out<!TLDS>(a, stage0 + (buf + 1u == nbuf ? 0u : buf + 1u) * a.stage_cap, n_rec, base, lane);
            }
        };
        if (NB == 3) emit_pending(pend_have[LAG_MAX - 1], pend_cnt[LAG_MAX - 1]);      // stores right behind the loads
        PFAC_STAMP(trace && r > 0, 11);

        // ---- root test -> 32-bit survivor mask per lane per half-tile; level-2 filter -> which of them are kept
        // (yield a record or need a walk) and which of those are deep (need the walk)
        unsigned keep[MSUBS] = {0u, 0u}, deep[MSUBS] = {0u, 0u};
        unsigned *stage = stage0 + buf * a.stage_cap;
        // dense mode on fused tables with packed dense rows: no classification, no rounds (dense2_tile); a tile it gives
        // up on (log full, > 15 patterns at one offset) goes the classic way below
        unsigned d2cnt = ~0u;
        if (NW == 4 && FUSED) {
            if (d2) {
                const unsigned long long own = a.n_owned - tile_base;
                d2cnt = dense2_tile<W8>(a, tile, t0_l, d1, aux, d2log, lane, lim, own < (unsigned long long)WTILE ? (unsigned)own : (unsigned)WTILE);
            }
        }
        const bool d2done = d2cnt != ~0u;
        PFAC_STAMP(trace && r > 0 && d2, 12);
        if (!d2done) {
#pragma unroll
        for (int j = 0; j < MSUBS; j++) {
            const unsigned off = j * MSUB + lane * MLANE;
            const u32x4 lo16 = *reinterpret_cast<const u32x4 *>(tile + off);
            const u32x4 hi16 = *reinterpret_cast<const u32x4 *>(tile + off + 16);
#ifdef PFAC_ABL_NOROOT                         // ablation builds only: no root test (nothing survives)
            const unsigned rlo = (lo16[0] == 0x12345678u), rhi = (hi16[0] == 0x12345678u);
#else
            const unsigned rlo = root_mask<ROOT>(lo16, ftab, a.root_byte), rhi = root_mask<ROOT>(hi16, ftab, a.root_byte);
#endif
            const unsigned raw = (rlo & 0xFFFFu) | (rhi << 16);
#ifdef PFAC_TRACE_BUILD
            asm volatile("" :: "v"(raw));
            PFAC_STAMP(trace && r > 0, j == 0 ? 12 : 14);
#endif
            unsigned m1 = raw;
            if (tile_base + WTILE > a.n_owned) {   // last tile only: offsets at or past n_owned start no walk
                const unsigned long long g = tile_base + off;
                if (g + MLANE > a.n_owned) m1 = g >= a.n_owned ? 0u : (m1 & ((1u << (unsigned)(a.n_owned - g)) - 1u));
            }
            if (a.l2f_mode == 0) {
                keep[j] = m1;
                deep[j] = m1;
            } else if (ROOT == 1 && a.l2f_mode == 1) {
                // bit-parallel: deep <=> the NEXT byte is a child byte of the depth-1 state (byte 32 = the first byte of
                // the next lane's run, or of the halo)
                unsigned nextc = 0;
                if (a.n_child > 0) {
                    const unsigned nx = *reinterpret_cast<const unsigned *>(tile + off + 32);
                    unsigned ec = a.child0 == a.root_byte ? raw : eq_mask32(lo16, hi16, a.child0);
                    unsigned e32 = ((nx ^ a.child0) & 0xFFu) == 0u ? 1u : 0u;
                    if (a.n_child > 1) {
                        ec |= a.child1 == a.root_byte ? raw : eq_mask32(lo16, hi16, a.child1);
                        e32 |= ((nx ^ a.child1) & 0xFFu) == 0u ? 1u : 0u;
                    }
                    nextc = (ec >> 1) | (e32 << 31);
                }
                deep[j] = m1 & nextc;
                keep[j] = root_final ? m1 : deep[j];
            } else {
                // one lookup per survivor in the 2-byte-prefix bitmap (and, for a multi-edge root, in the
                // depth-1-state-is-final table).  Per-lane loop, up to L2F_UNROLL survivors per trip: their lookups are
                // independent chains of two LDS round trips, so a trip costs one chain, not four.
                unsigned dm = 0, fm = 0;
                const unsigned *t32 = reinterpret_cast<const unsigned *>(tile);
                unsigned cand = m1;
                if (ROOT != 1 && a.sec_filter) {
                    // multi-edge root without 1-byte patterns: only survivors whose NEXT byte can be a second byte at
                    // all are looked up (flags of bytes 1..31 from the root test's own lookups, byte 32 = one more)
                    const unsigned nx = *reinterpret_cast<const unsigned *>(tile + off + 32);
                    const unsigned s32 = ((unsigned)ftab[nx & 0xFFu] >> 4) & 1u;
                    cand = m1 & ((((rlo >> 16) | (rhi & 0xFFFF0000u)) >> 1) | (s32 << 31));
                }
#ifdef PFAC_ABL_NOCLASS                        // ablation builds only: no survivor is looked up (no records)
                cand = 0;
#endif
                if (ROOT != 1 && a.l2f_mode == 3) {
                    // dense pair matrix (most flagged first bytes are followed by most flagged second bytes in the
                    // trie): the per-survivor bitmap lookup -- a serial, per-lane chain of LDS round trips -- would drop
                    // few of these, so they all go to the walk, whose dense depth-1 row IS the pair lookup, 64 lanes wide
                    dm = cand;
                    cand = 0;
                }
                for (unsigned mm = cand; mm;) {
                    unsigned b[L2F_UNROLL], win[L2F_UNROLL], v[L2F_UNROLL], fin[L2F_UNROLL];
                    bool on[L2F_UNROLL];
#pragma unroll
                    for (int u = 0; u < L2F_UNROLL; u++) {
                        on[u] = mm != 0u;
                        b[u] = on[u] ? (unsigned)__ffs(mm) - 1u : 0u;
                        mm &= mm - 1u;                              // (0 stays 0)
                        const unsigned p = off + b[u];
                        const unsigned lo = t32[p >> 2], hi = t32[(p >> 2) + 1];
                        win[u] = __builtin_amdgcn_alignbyte(hi, lo, p & 3u);   // bytes p .. p+3
                    }
#pragma unroll
                    for (int u = 0; u < L2F_UNROLL; u++) {
                        const unsigned b0 = win[u] & 0xFFu, b1 = (win[u] >> 8) & 0xFFu;
                        v[u] = bm2l[(ROOT == 1 ? 0u : b0 * 32u) + (b1 >> 3)];
                        fin[u] = ROOT == 1 ? 0u : (unsigned)finl[b0];
                    }
#pragma unroll
                    for (int u = 0; u < L2F_UNROLL; u++) {
                        const unsigned b1 = (win[u] >> 8) & 0xFFu;
                        dm |= on[u] ? ((v[u] >> (b1 & 7u)) & 1u) << b[u] : 0u;
                        fm |= on[u] ? fin[u] << b[u] : 0u;
                    }
                }
                deep[j] = dm;
                keep[j] = dm | (ROOT == 1 ? (root_final ? m1 : 0u) : fm);
            }
#ifdef PFAC_TRACE_BUILD
            asm volatile("" :: "v"(keep[j]), "v"(deep[j]));
            PFAC_STAMP(trace && r > 0 && j == 0, 13);
#endif
        }
        }

#ifdef PFAC_ABL_NOKEEP                         // ablation builds only: survivors classified, then dropped (no records)
        asm volatile("" :: "v"(keep[0]), "v"(keep[1]), "v"(deep[0]), "v"(deep[1]));
        keep[0] = keep[1] = deep[0] = deep[1] = 0;
#endif
        PFAC_STAMP(trace, 6);
        // ---- compact + walk once; records staged in LDS buffer `buf`; post the count
        // (a tile nothing survives in -- most tiles of a sparse pattern set -- goes straight to posting its zero)
        const unsigned long long cnt = d2done ? (unsigned long long)d2cnt : !__any((keep[0] | keep[1]) != 0u) ? 0ull :
            tile_pass<W8, false, NW, FUSED, ROOT>(a, tile, s0, d1, R, T, q, stage, keep, deep, lane, lim, tile_base, 0);
        PFAC_STAMP(trace, 7);
#ifdef PFAC_TRACE_BUILD
        if (a.dbg && blockIdx.x < 8 && r < 64 && lane == 0) tr[16 + wave] = __builtin_amdgcn_s_memrealtime();
#endif
        const bool overflow = cnt > a.stage_cap;
        const bool now = overflow || nbuf == 1;          // this tile is emitted right away (needs its base at once)
        unsigned arrival = 0;
#ifdef PFAC_ABL_NOCOORD
        if (false)
#endif
        if (lane == 0) {
            hdr[H_CNT + (r & 7) * 16 + wave] = (unsigned)cnt | ((now && cnt != 0) ? 0x80000000u : 0u);
            if (cnt > a.small_cap) {
                atomicAdd(&hdr[H_OVF2], 1u);
                if (cnt > a.sparse_cap) atomicAdd(&hdr[H_OVF], 1u);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            arrival = atomicAdd(&hdr[H_ARRIVED + (r & 7)], 1u);
        }
        // The SIMD arbitrates VALU issue by priority, then age, so the youngest wave of each SIMD falls
        // behind and everybody ends up waiting for it.  Waves that arrived in the later half of this round
        // run the next one at raised priority: the laggards catch up, the round time approaches the mean.
        if (__builtin_amdgcn_readfirstlane(arrival) * 2u >= (unsigned)nc) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);

        if (now && cnt != 0) {
            // A tile that leaves at once waits for nobody: it takes exactly its records from the heap cursor itself (one
            // device atomic per tile -- these are the dense tiles, tens of microseconds each) and writes its own index
            // word; the coordinator only adds its count to the total.
            unsigned long long v = 0;
            if (lane == 0)
                v = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(a.ctl + CTL_CURSOR), (unsigned long long)cnt,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long base = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) << 32) |
                                            __builtin_amdgcn_readfirstlane((unsigned)v);
            if (lane == 0) a.tile_index[t] = base | ((unsigned long long)cnt << TIX_CNT_SHIFT);
            PFAC_STAMP(trace && r > 0 && d2, 13);
            if (NW == 4 && FUSED && d2done) {
                dense2_scatter(a, aux, d2log, (unsigned)cnt, base, lane);
                PFAC_STAMP(trace && r > 0, 14);
            }
            else if (overflow)
                // staging overflowed (or the automaton is too large for packed staging): walk the tile again,
                // while its bytes are still in LDS, writing straight to global memory
                tile_pass<W8, true, NW, FUSED, ROOT>(a, tile, s0, d1, R, T, q, stage, keep, deep, lane, lim, tile_base, base);
            else
                copy_out<!TLDS>(a, stage, (unsigned)cnt, base, lane);   // dense mode: staged, emitted at once
        }
        if (NB == 2) emit_pending(pend_have[0], pend_cnt[0]);
        PFAC_STAMP(trace, 8);
        if (NB == 3) { pend_have[LAG_MAX - 1] = pend_have[0]; pend_cnt[LAG_MAX - 1] = pend_cnt[0]; }
        pend_have[0] = !now; pend_cnt[0] = (unsigned)cnt;
        buf = buf + 1u >= nbuf ? 0u : buf + 1u;
        if (LOAD_DEPTH == 1) {
            if (!more_far) return false;
            t = t_far;
        } else {
            if (!have_n1) return false;
            t = t_n1;
            t_n1 = t_far;
            have_n1 = more_far;
        }
        r++;
        return true;
    };
    for (;;) {
        if (!round_body(wA, hA)) break;
        if (LOAD_DEPTH == 2 && !round_body(wB, hB)) break;
    }
    // drain: the tiles of the last `lag` rounds (pend_have[k]: round r - k, staged in buffer (r - k) % nbuf)
#pragma unroll
    for (int k = LAG_MAX - 1; k >= 0; k--) {
        if ((unsigned)k < lag && pend_have[k] && pend_cnt[k] != 0) {
            unsigned long long base = 0;
            if (record_base(r - (unsigned)k, base))
                copy_out<!TLDS>(a, stage0 + ((buf + nbuf - 1u - (unsigned)k) % nbuf) * a.stage_cap, pend_cnt[k], base, lane);
        }
    }
}

template <bool TLDS, bool W8, int ROOT, bool FUSED, int NW, int NB = 2>
__global__ __launch_bounds__(WAVE * MAX_WAVES_PER_BLOCK) void pfac_scan_kernel(ScanArgs a) {
    static_assert(NB == 2 || (NB == 3 && NW <= 3), "three staging buffers: the sparse-mode kernels (dense mode has one)");
    static_assert(!(TLDS && FUSED), "the fused table is for tables gathered through L2");
    static_assert(NW == (TLDS ? 1 : 2) || (FUSED && (NW == 3 || NW == 4)), "walks per lane: 1 (LDS tables), 2 (L2 tables), 3 (L2, fused), 4 (L2, fused, dense matches)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const ErrCh err = {&a.ctl[CTL_ERR], a.spin_max};
#ifdef PFAC_TRACE_BUILD                        // column 15 of a traced workgroup's rows 0 / 1 / 2: entry, tables staged, last wave out
    if (a.dbg && blockIdx.x < 8 && threadIdx.x == 0) a.dbg[((size_t)blockIdx.x * 64 + 0) * 32 + 15] = __builtin_amdgcn_s_memrealtime();
#endif
    scan_body<TLDS, W8, ROOT, FUSED, NW, NB>(a, smem, err);
    // ---- leaving: the last wave of the workgroup counts the workgroup out; the last workgroup of the grid copies
    // the error flags and the dense-tile count from the control header (device memory) to the host-visible result
    // words with plain stores.  Every wave comes through here, whichever way it left the loop.
    unsigned *hdr = reinterpret_cast<unsigned *>(smem + SH_HDR);
    if ((threadIdx.x & (WAVE - 1)) == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's flag atomics have been performed
        const unsigned left = atomicAdd(&hdr[H_EXIT], 1u);
        if (left + 1 == (blockDim.x >> 6)) {
#ifdef PFAC_TRACE_BUILD
            if (a.dbg && blockIdx.x < 8) a.dbg[((size_t)blockIdx.x * 64 + 2) * 32 + 15] = __builtin_amdgcn_s_memrealtime();
#endif
            const unsigned done = __hip_atomic_fetch_add(&a.ctl[CTL_DONE], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (done + 1 == gridDim.x) {
                const unsigned long long tot = __hip_atomic_load(reinterpret_cast<unsigned long long *>(a.ctl + CTL_TOTAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long cur = __hip_atomic_load(reinterpret_cast<unsigned long long *>(a.ctl + CTL_CURSOR), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a.res[0] = (unsigned)tot; a.res[1] = (unsigned)(tot >> 32);
                a.res[2] = __hip_atomic_load(&a.ctl[CTL_ERR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a.res[3] = __hip_atomic_load(&a.ctl[CTL_OVF], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a.res[4] = (unsigned)cur; a.res[5] = (unsigned)(cur >> 32);
                a.res[6] = __hip_atomic_load(&a.ctl[CTL_OVF2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// small service kernels

// blob (int32 image, pfac.h) -> device layout {s0[256] | r[max_row] | pad | T[ht_size] int2 | idmap}
// (*bad is set when the image would send a walk outside the tables: a row displacement outside the hash table, a root
// or next state that is no state -- the fused walk indexes with these values unchecked; its slot array is padded to
// max(r) + width entries, see pfac_fuse_kernel)
__global__ void pfac_repack_kernel(const int *blob, int *s0, int *r, int2 *T, int *idmap, int max_row, int ht_size,
                                   int num_final, int state_num, int width, int *bad) {
    const int *b_s0 = blob + PFAC_BLOB_HEADER_WORDS;
    const int *b_r = b_s0 + 256;
    const int *b_HT = b_r + max_row;
    const int *b_val = b_HT + ht_size;
    const int *b_id = b_val + ht_size;
    const int stride = gridDim.x * blockDim.x;
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = true;
    for (int i = i0; i < 256; i += stride) { s0[i] = b_s0[i]; ok = ok && b_s0[i] >= -1 && b_s0[i] < state_num; }
    for (int i = i0; i < max_row; i += stride) { r[i] = b_r[i]; ok = ok && b_r[i] > -width && b_r[i] < ht_size; }
    for (int i = i0; i < ht_size; i += stride) {
        T[i] = make_int2(b_HT[i], b_val[i]);
        ok = ok && (b_HT[i] < 0 || (b_val[i] >= -1 && b_val[i] < state_num));     // (an unowned slot's value is never used)
    }
    for (int i = i0; i < num_final; i += stride) idmap[i] = b_id[i];
    if (!ok) *bad = 1;
}

// Child mask of a state (the fused walk's prefilter): bit (c & 31) set iff the state has an edge on byte c.
__device__ __forceinline__ unsigned child_mask32(const int *r, const int2 *T, int wbit, int ht_size, int max_row, int state) {
    unsigned m = 0;
    if (state < 0) return 0u;
    for (int c = 0; c < 256; c++) {
        const int key = (state << 8) | c;
        const int row = key >> wbit;
        if (row >= max_row) continue;
        const int idx = r[row] + (key & ((1 << wbit) - 1));
        if ((unsigned)idx < (unsigned)ht_size) {
            const int2 e = T[idx];
            if (e.x == row && e.y >= 0) m |= 1u << (c & 31);
        }
    }
    return m;
}

// Fused slots for tables gathered through L2 (PHF width >= 256): T4[i] = {owner row, next state, r[row of next], 0}.
// (slots [lo, hi) with lo <= min(r), hi >= max(r) + width: the table image holds [0, ht_size) -- displacements may be
// negative and the image ends at its last used slot -- but the fused walk indexes unchecked; the slots outside the
// image are empty ones.  T4 points at slot 0.)
__global__ void pfac_fuse_kernel(const int2 *T, const int *r, int wbit, int ht_size, int max_row, int4 *T4, int lo, int hi) {
    const int stride = gridDim.x * blockDim.x;
    for (int i = lo + blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += stride) {
        const int2 e = (i >= 0 && i < ht_size) ? T[i] : make_int2(-1, -1);
        int rn = -1;                           // (.z: r[row of the next state], biased like every fused index: slot - lo)
        bool have = false;
        if (e.y >= 0) {
            const int row = e.y >> (wbit - 8);
            if (row < max_row) { rn = r[row]; have = true; }
        }
        // .w: the child mask of the next state -- what lets the walk drop a walker without the gather that would
        // (only owned slots are ever selected, so an unowned slot's mask is never used)
        T4[i] = make_int4(e.x, e.y, have ? rn - lo : 0, e.x >= 0 ? (int)child_mask32(r, T, wbit, ht_size, max_row, e.y) : 0);
    }
}

// Dense rows of the depth-1 states: row f, column c = lookup(state d1state[f], byte c) through the PHF.
__global__ void pfac_build_d1_kernel(const int *d1state, const int *r, const int2 *T, int wbit, int ht_size, int *d1) {
    const int st = d1state[blockIdx.x];
    const int c = threadIdx.x;
    const int key = (st << 8) | c;
    const int row = key >> wbit;
    const int idx = r[row] + (key & ((1 << wbit) - 1));
    int nx = -1;
    if ((unsigned)idx < (unsigned)ht_size) {
        const int2 e = T[idx];
        if (e.x == row) nx = e.y;
    }
    d1[blockIdx.x * 256 + c] = nx;
}

// FUSED: pack the dense rows in place -- entry = state | k << 20 with r2[k] = {r[row of that state], its child mask};
// *counter hands out k.
__global__ void pfac_pack_d1_kernel(int *d1, int n_entries, const int *r, const int2 *T, int wbit, int ht_size, int max_row, int2 *r2, int *counter) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_entries) return;
    const int nx = d1[i];
    if (nx < 0) return;
    const int k = atomicAdd(counter, 1);
    const int row = nx >> (wbit - 8);
    r2[k] = make_int2(row < max_row ? r[row] : -1, (int)child_mask32(r, T, wbit, ht_size, max_row, nx));
    d1[i] = nx | (k << D1_STATE_BITS);
}

// 2-byte-prefix bitmap: bit (b0 << 8 | b1) set iff the root has an edge on b0 and that state one on b1 (the level-2
// filter of the scan).  One block per b0, one thread per b1; a wave's ballot is 8 bytes of the row.
__global__ void pfac_build_bm2_kernel(const int *s0, const int *r, const int2 *T, int wbit, int ht_size, unsigned long long *bm2) {
    const int st = s0[blockIdx.x];
    const int c = threadIdx.x;
    int nx = -1;
    if (st >= 0) {
        const int key = (st << 8) | c;
        const int row = key >> wbit;
        const int idx = r[row] + (key & ((1 << wbit) - 1));
        if ((unsigned)idx < (unsigned)ht_size) {
            const int2 e = T[idx];
            if (e.x == row) nx = e.y;
        }
    }
    const unsigned long long bits = __ballot(nx >= 0);
    if ((c & (WAVE - 1)) == 0) bm2[blockIdx.x * 4 + (c >> 6)] = bits;
}

__device__ __forceinline__ unsigned long long match_hash(unsigned long long pos, unsigned id) {
    unsigned long long x = (pos + 1) * 0x9E3779B97F4A7C15ull ^ ((unsigned long long)id * 0xC2B2AE3D27D4EB4Full);
    x ^= x >> 29;
    return x * 0xBF58476D1CE4E5B9ull;
}

// Records are read through the tile index (the heap has gaps): tile t holds tix[t] >> 40 records from index
// tix[t] & TIX_BASE_MASK on; records past the capacity of the array were never written.
template <int BYTES>
__device__ __forceinline__ void heap_record(const void *rec, unsigned long long i, unsigned long long tile, unsigned &pos, unsigned &state) {
    if (BYTES == 2) {
        const unsigned w = static_cast<const unsigned short *>(rec)[i];
        pos = (unsigned)(tile * WTILE) + (w & 0xFFFu);
        state = w >> 12;
    } else if (BYTES == 4) {
        const unsigned w = static_cast<const unsigned *>(rec)[i];
        pos = (unsigned)(tile * WTILE) + (w & 0xFFFu);
        state = w >> 12;
    } else {
        const pfac_record r = static_cast<const pfac_record *>(rec)[i];
        pos = r.pos;
        state = r.state;
    }
}

// Order-independent checksum of all records: one wave per tile at a time.
template <int BYTES>
__global__ void pfac_checksum_kernel(const void *rec, const unsigned long long *tix, unsigned long long n_tiles,
                                     unsigned long long cap, unsigned long long base, const int *idmap, unsigned long long *out) {
    unsigned long long sum = 0;
    const int lane = threadIdx.x & (WAVE - 1);
    const unsigned long long wstride = (unsigned long long)gridDim.x * (blockDim.x >> 6);
    for (unsigned long long t = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); t < n_tiles; t += wstride) {
        const unsigned long long e = tix[t], lo = e & TIX_BASE_MASK;
        const unsigned cnt = (unsigned)(e >> TIX_CNT_SHIFT);
        for (unsigned i = (unsigned)lane; i < cnt && lo + i < cap; i += WAVE) {
            unsigned pos, st;
            heap_record<BYTES>(rec, lo + i, t, pos, st);
            sum += match_hash(base + pos, (unsigned)idmap[st]);
        }
    }
    sum = wave_sum64(sum);
    if (lane == 0) atomicAdd(out, sum);
}

// Heap -> one sorted pfac_record array, three small kernels: records per group of 64 tiles, exclusive scan of the
// group sums (one block), copy.  Only consumers that want the flat sorted array pay for this.
constexpr int XGROUP = 64;
__global__ void pfac_tix_group_sum_kernel(const unsigned long long *tix, unsigned long long n_tiles, unsigned long long *gsum, unsigned n_groups) {
    const int lane = threadIdx.x & (WAVE - 1);
    const unsigned g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g >= n_groups) return;
    const unsigned long long t = (unsigned long long)g * XGROUP + lane;
    const unsigned c = t < n_tiles ? (unsigned)(tix[t] >> TIX_CNT_SHIFT) : 0u;
    const unsigned long long sum = wave_sum64(c);
    if (lane == 0) gsum[g] = sum;
}
__global__ void pfac_scan_groups_kernel(unsigned long long *gsum, unsigned n_groups) {      // ONE block of 1024 threads
    __shared__ unsigned long long part[1024];
    const unsigned per = (n_groups + 1023u) / 1024u;
    const unsigned lo = threadIdx.x * per, hi = lo + per < n_groups ? lo + per : n_groups;
    unsigned long long sum = 0;
    for (unsigned i = lo; i < hi; i++) sum += gsum[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long acc = 0;
        for (int i = 0; i < 1024; i++) { const unsigned long long v = part[i]; part[i] = acc; acc += v; }
    }
    __syncthreads();
    unsigned long long acc = part[threadIdx.x];
    for (unsigned i = lo; i < hi; i++) { const unsigned long long v = gsum[i]; gsum[i] = acc; acc += v; }
    if (hi == n_groups && lo < hi) gsum[n_groups] = acc;       // (the thread that owns the last group: the grand total)
}
// records [first, first + n) of the sorted sequence -> out[0, n)
template <int BYTES>
__global__ void pfac_expand_kernel(const void *rec, const unsigned long long *tix, unsigned long long n_tiles,
                                   const unsigned long long *gpre, unsigned n_groups, unsigned long long cap,
                                   unsigned long long first, unsigned long long n, pfac_record *out) {
    const int lane = threadIdx.x & (WAVE - 1);
    const unsigned g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g >= n_groups) return;
    const unsigned long long t = (unsigned long long)g * XGROUP + lane;
    const unsigned long long e = t < n_tiles ? tix[t] : 0ull;
    const unsigned c = (unsigned)(e >> TIX_CNT_SHIFT);
    const unsigned long long lo = e & TIX_BASE_MASK;
    const unsigned long long off = gpre[g] + (wave_incl_scan(c) - c);     // sorted index of the tile's first record (< 2^32 per group)
    const unsigned long long end = first + n;
    for (int j = 0; j < XGROUP; j++) {
        const unsigned tc = __shfl(c, j, WAVE);
        if (tc == 0) continue;
        const unsigned long long tlo = __shfl(lo, j, WAVE), toff = __shfl(off, j, WAVE);
        if (toff >= end || toff + tc <= first) continue;
        for (unsigned i = (unsigned)lane; i < tc; i += WAVE) {
            const unsigned long long k = toff + i;
            if (k < first || k >= end || tlo + i >= cap) continue;
            pfac_record o;
            heap_record<BYTES>(rec, tlo + i, (unsigned long long)g * XGROUP + j, o.pos, o.state);
            out[k - first] = o;
        }
    }
}

// ---------------------------------------------------------------------------
// GPU-side text emitter (replaces the fprintf loop of main.cc:335-350 on the device): the compact records of a scan ->
// the lines  "At position %4d, match pattern %d\n"  in output order, in one device buffer.  Three kernels: bytes per
// group of 64 tiles (line length depends on the digit counts), exclusive scan of the group sums (pfac_scan_groups_kernel),
// format.  The host only copies the finished text and write()s it.
__device__ __forceinline__ unsigned ndigits32(unsigned v) {
    return 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) + (v >= 10000000u) +
           (v >= 100000000u) + (v >= 1000000000u);
}
__device__ __forceinline__ unsigned ndigits64(unsigned long long v) {       // v < 10^18
    return v < 1000000000ull ? ndigits32((unsigned)v) : 9u + ndigits32((unsigned)(v / 1000000000ull));
}
__device__ __forceinline__ unsigned text_line_len(unsigned long long pos, unsigned id) {
    const unsigned dp = ndigits64(pos);
    return 12u + (dp < 4u ? 4u : dp) + 16u + ndigits32(id) + 1u;
}
constexpr int TEXT_LINE_MAX = 12 + 18 + 16 + 10 + 1;          // positions below 10^18, ids below 2^32
constexpr int TEXT_IMG = 16 + WAVE * TEXT_LINE_MAX + 15;       // LDS image of one chunk of 64 lines, congruent mod 16 with its place in the text

// bytes of text per group of XGROUP tiles (lane j of a wave: tile g * 64 + j; its records by all lanes in turn)
template <int BYTES>
__global__ void pfac_text_size_kernel(const void *rec, const unsigned long long *tix, unsigned long long n_tiles, unsigned long long cap,
                                      unsigned long long base, const int *idmap, unsigned long long *gsum, unsigned n_groups) {
    const int lane = threadIdx.x & (WAVE - 1);
    const unsigned g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g >= n_groups) return;
    const unsigned long long t = (unsigned long long)g * XGROUP + lane;
    const unsigned long long e = t < n_tiles ? tix[t] : 0ull;
    const unsigned c = (unsigned)(e >> TIX_CNT_SHIFT);
    const unsigned long long lo = e & TIX_BASE_MASK;
    unsigned long long total = 0;
    for (int j = 0; j < XGROUP; j++) {
        const unsigned tc = __shfl(c, j, WAVE);
        if (tc == 0) continue;
        const unsigned long long tlo = __shfl(lo, j, WAVE);
        unsigned long long sum = 0;
        for (unsigned i = (unsigned)lane; i < tc && tlo + i < cap; i += WAVE) {
            unsigned pos, st;
            heap_record<BYTES>(rec, tlo + i, (unsigned long long)g * XGROUP + j, pos, st);
            sum += text_line_len(base + pos, (unsigned)idmap[st]);
        }
        total += sum;                                          // (per lane; summed over the wave once, below)
    }
    total = wave_sum64(total);
    if (lane == 0) gsum[g] = total;
}

// decimal digits of v (nd of them, most significant first) to p[0..nd)
__device__ __forceinline__ void put_digits32(unsigned char *p, unsigned v, unsigned nd) {
    for (unsigned k = nd; k-- > 0;) { const unsigned q = v / 10u; p[k] = (unsigned char)('0' + (v - q * 10u)); v = q; }
}

template <int BYTES>
__global__ __launch_bounds__(256) void pfac_text_format_kernel(const void *rec, const unsigned long long *tix, unsigned long long n_tiles,
                                                               unsigned long long cap, unsigned long long base, const int *idmap,
                                                               const unsigned long long *gpre, unsigned n_groups, unsigned char *text) {
    __shared__ __attribute__((aligned(16))) unsigned char img_all[4][(TEXT_IMG + 15) / 16 * 16];
    const int lane = threadIdx.x & (WAVE - 1);
    unsigned char *img = img_all[threadIdx.x >> 6];
    const unsigned g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g >= n_groups) return;
    const unsigned long long t = (unsigned long long)g * XGROUP + lane;
    const unsigned long long e = t < n_tiles ? tix[t] : 0ull;
    const unsigned c = (unsigned)(e >> TIX_CNT_SHIFT);
    const unsigned long long lo = e & TIX_BASE_MASK;
    unsigned long long at = gpre[g];                           // text offset of the next line (wave-uniform)
    for (int j = 0; j < XGROUP; j++) {
        const unsigned tc = __shfl(c, j, WAVE);
        if (tc == 0) continue;
        const unsigned long long tlo = __shfl(lo, j, WAVE);
        for (unsigned c0 = 0; c0 < tc; c0 += WAVE) {           // a chunk of up to 64 lines
            const unsigned i = c0 + (unsigned)lane;
            const bool have = i < tc && tlo + i < cap;
            unsigned pos = 0, st = 0, id = 0, len = 0, dp = 0, di = 0;
            unsigned long long gp = 0;
            if (have) {
                heap_record<BYTES>(rec, tlo + i, (unsigned long long)g * XGROUP + j, pos, st);
                id = (unsigned)idmap[st];
                gp = base + pos;
                dp = ndigits64(gp);
                di = ndigits32(id);
                len = 12u + (dp < 4u ? 4u : dp) + 16u + di + 1u;
            }
            const unsigned incl = wave_incl_scan(len);
            const unsigned total = bcast_last(incl);
            const unsigned a0 = (unsigned)(at & 15ull);        // the image starts at the 16-byte block the chunk's first byte lies in
            if (have) {
                unsigned char *p = img + a0 + (incl - len);
                const char *h = "At position ";
#pragma unroll
                for (int k = 0; k < 12; k++) p[k] = (unsigned char)h[k];
                p += 12;
                for (unsigned k = dp; k < 4u; k++) *p++ = ' ';                               // %4d
                if (gp < 1000000000ull) put_digits32(p, (unsigned)gp, dp);
                else {
                    const unsigned hi = (unsigned)(gp / 1000000000ull), lo9 = (unsigned)(gp - (unsigned long long)hi * 1000000000ull);
                    put_digits32(p, hi, dp - 9u);
                    put_digits32(p + (dp - 9u), lo9, 9u);
                }
                p += dp;
                const char *m = ", match pattern ";
#pragma unroll
                for (int k = 0; k < 16; k++) p[k] = (unsigned char)m[k];
                p += 16;
                put_digits32(p, id, di);
                p[di] = (unsigned char)'\n';
            }
            wave_lds_sync();
            // image bytes [a0, a0 + total) -> text[at, at + total): whole 16-byte blocks with one store per lane, the ragged
            // first and last block byte by byte (their other bytes belong to the neighbouring chunks)
            unsigned char *dst = text + (at - a0);            // 16-byte aligned (the text buffer is)
            const unsigned end = a0 + total;
            const unsigned b_lo = a0 ? 16u : 0u, b_hi = end & ~15u;   // full blocks cover [b_lo, b_hi)
            if (b_hi > b_lo) {
                for (unsigned o = b_lo + 16u * (unsigned)lane; o < b_hi; o += 16u * WAVE)
                    __builtin_nontemporal_store(*reinterpret_cast<const u32x4 *>(img + o), reinterpret_cast<u32x4 *>(dst + o));
                if (a0 && (unsigned)lane < 16u - a0) dst[a0 + lane] = img[a0 + lane];
                if ((unsigned)lane < end - b_hi) dst[b_hi + lane] = img[b_hi + lane];
            } else {
                // no whole block: at most 31 bytes (e.g. one short line): byte by byte
                if (a0 + (unsigned)lane < end) dst[a0 + lane] = img[a0 + lane];
            }
            wave_lds_sync();
            at += total;
        }
    }
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// byte i = byte (i & 7) (little endian) of splitmix64(seed + (i >> 3)); dst 8-B aligned, n rounded up by caller
__global__ void pfac_fill_random_kernel(unsigned long long *dst, unsigned long long n_words, unsigned long long seed) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride)
        dst[i] = splitmix64(seed + i);
}

// byte i = pat[(phase + i) % period]
__global__ void pfac_fill_tiled_kernel(unsigned char *dst, unsigned long long n, const unsigned char *pat,
                                       unsigned period, unsigned long long phase) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x * 16;
    for (unsigned long long i = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += stride) {
        unsigned k = (unsigned)((phase + i) % period);
        if (i + 16 <= n) {
            union { unsigned char b[16]; uint4 v; } u;
#pragma unroll
            for (int j = 0; j < 16; j++) { u.b[j] = pat[k]; k = k + 1 == period ? 0 : k + 1; }
            *reinterpret_cast<uint4 *>(dst + i) = u.v;
        } else {
            for (unsigned long long j = i; j < n; j++) { dst[j] = pat[k]; k = k + 1 == period ? 0 : k + 1; }
        }
    }
}

// ---------------------------------------------------------------------------
// runtime

thread_local std::string g_err;

struct Slot {
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    unsigned char *d_input = nullptr;
    uint64_t input_cap = 0;
    void *d_records = nullptr;            // record_cap x 8 bytes: holds either record format
    uint64_t record_cap = 0;
    unsigned long long *d_tile_index = nullptr;   // per tile of the last scan: first record | count << 40
    unsigned *d_d2log = nullptr;                  // dense mode, second form: the record logs of the grid's compute waves
    size_t d2log_words = 0;
    uint64_t tile_cap = 0;
    unsigned long long *d_gsum = nullptr; // scratch of the expand / text paths: record (byte) prefix per group of 64 tiles (+ the total)
    uint64_t gsum_cap = 0;
    unsigned char *d_text = nullptr;      // pfac_emit_text_device: the formatted lines of the slot's last scan
    uint64_t text_cap = 0, text_bytes = 0;
    pfac_record *d_wide = nullptr;        // scratch of pfac_records_d2h: packed records expanded on the device
    uint64_t wide_cap = 0;
    int last_rec_bytes = 4;               // record form of the slot's last scan (2, 4 or 8 bytes)
    const void *last_records = nullptr;   // ... and where it wrote
    unsigned *d_ctl = nullptr;            // TWO control headers (ticket counters, flags, heap cursor), used alternately:
    unsigned *d_ctlbuf[2] = {nullptr, nullptr};   // a scan zeroes the other one for the scan after it
    bool clean[2] = {false, false};       // header known to be zero
    int flip = 0;                         // header of the next scan
    unsigned *h_ctl = nullptr;            // pinned, device-visible: [0..1] matches, [2] err, [3] dense tiles, [4..5] heap
                                          // records used (written by the kernel), [8..9] checksum
    unsigned *d_res = nullptr;            // device-side address of h_ctl
    unsigned long long *d_sum = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_h2d = nullptr;          // behind the slot's last pfac_slot_h2d: the host buffer may be reused once it has fired
    bool h2d_issued = false;
    uint64_t last_cap = 0, last_tiles = 0, last_total = 0, last_used = 0;
    bool scanned = false, pending = false, last_dense = false;
    unsigned long long *d_dbg = nullptr;  // PFAC_TRACE_BUILD + PFAC_TRACE
};

}  // namespace

struct StageLayout {                      // per-wave LDS of one staging mode
    int nbuf = 2, pw_bytes = 0, waves_per_block = 0, lds_bytes = 0;
    unsigned stage_cap = 0;               // records per buffer (0: final states do not fit the packed word)
};

struct pfac_ctx {
    int device = 0;
    int n_cu = 0;
    std::vector<Slot> slots;
    hipStream_t copy_stream = nullptr;    // every pfac_slot_h2d goes through this ONE stream, in call order: copies queued back to
                                          // back on one DMA queue run at the link's rate (55 GB/s), copies that alternate with
                                          // kernels on the slots' own streams leave gaps (45 GB/s measured); the slot's stream
                                          // waits for its copy through an event
    // table
    int *d_tab = nullptr;
    size_t tab_bytes = 0;
    int *d_s0 = nullptr, *d_r = nullptr, *d_idmap = nullptr;
    int2 *d_T = nullptr;
    int4 *d_T4 = nullptr;                 // fused slots (variant 1, width >= 256), else null: slot 0 of ...
    int4 *d_T4_alloc = nullptr;           // ... this allocation, which starts at slot min(0, min r) = -rn_bias
    int rn_bias = 0;
    int width_bit = 0, num_final = 0, max_pat_len = 0, max_row = 0, ht_size = 0, state_num = 0;
    bool have_table = false;
    int variant = 1;
    const void *kernel = nullptr;
    const void *kernel_d = nullptr;       // the kernel dense mode launches (four walks per lane on fused L2 tables)
    const void *kernel3 = nullptr;        // the three-staging-buffer twin of `kernel` (tables in LDS only), else null
    int shared_bytes = 0, halo = 0, root_mode = 0;
    unsigned root_byte = 0;
    int root_state = -1;
    StageLayout lay[2];                   // sparse staging: [0] two buffers (emission lags one round), [1] three (two rounds)
    bool lag2_ok = false, lag2 = false, lag_forced = false;   // three-buffer layout usable / in use / pinned (PFAC_LAG)
    const StageLayout &sparse() const { return lay[lag2 ? 1 : 0]; }
    // the dense-mode twin of {pw_bytes, waves_per_block, lds_bytes, stage_cap}: one big staging buffer per wave
    int pw_bytes_d = 0, waves_per_block_d = 0, lds_bytes_d = 0;
    unsigned stage_cap_d = 0;
    bool dense = false;                   // current staging mode (adapts to the match density seen by the last scan)
    bool dense2 = false;                  // dense mode runs in its second form (dense2_tile): fused tables, packed dense rows, 4-byte records
    unsigned d2log_cap = 0;               // ... words of record log per compute wave (test knob PFAC_D2_LOGCAP: a smaller one)
    int dense_forced = -1;                // PFAC_DENSE=0/1 pins the mode
    int *d_d1 = nullptr;                  // dense depth-1 rows + (after them) the 256-byte row index
    int d1_rows = 0, d1_stride = 0, d1_ncols = 0, d1_lds_bytes = 0;
    int d1_n2 = 0;                        // > 0: dense rows are packed (fused tables), r[] of the depth-2 states follows them
    int grid_blocks = 0;
    int rec_bytes = 4;                    // record form: 2 (<= 16 final states), 4 (<= 2^20), 8 bytes (pfac_record)
    // level-2 filter (ScanArgs::l2f_mode)
    unsigned char *d_bm2 = nullptr;       // 2-byte-prefix bitmap, 256 rows of 32 bytes
    int l2f_mode = 0, n_child = 0, bm2_rows = 0, sh_bm2 = 0, sec_filter = 0, sh_t0 = 0;
    unsigned child0 = 0, child1 = 0;
    // tuning / test knobs, read from the environment ONCE, when a table is installed
    unsigned spin_max = SPIN_MAX, fault = 0, ticket_ways_knob = 0;
    std::string trace_file;
    std::string err;
    std::mutex mu;
};

namespace {

int fail(pfac_ctx *ctx, int code, const std::string &msg) {
    g_err = msg;
    if (ctx) ctx->err = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(ctx, PFAC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

// Every entry point works on its context's device and leaves the calling thread's current device as it found it
// (a caller that drives several GPUs from one thread -- torch does -- must not have it changed under its feet).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
#define USE_DEVICE(ctx)                                                                                    \
    DeviceGuard device_guard_((ctx)->device);                                                              \
    if (device_guard_.err != hipSuccess)                                                                   \
        return fail(ctx, PFAC_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(device_guard_.err))

int check_slot(pfac_ctx *ctx, int slot) {
    if (!ctx) return fail(nullptr, PFAC_E_ARG, "null context");
    if (slot < 0 || slot >= (int)ctx->slots.size()) return fail(ctx, PFAC_E_ARG, "bad slot index");
    return PFAC_OK;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

constexpr size_t CTL_REGION = (CTL_WORDS * 4 + 255) / 256 * 256;

int ensure_ctl(pfac_ctx *ctx, Slot &s) {
    if (s.d_ctl) return PFAC_OK;
    HIP_TRY(ctx, hipMalloc((void **)&s.d_ctl, 2 * CTL_REGION));
    s.d_ctlbuf[0] = s.d_ctl;
    s.d_ctlbuf[1] = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(s.d_ctl) + CTL_REGION);
    s.clean[0] = s.clean[1] = false;
    s.flip = 0;
    return PFAC_OK;
}

int ensure_tiles(pfac_ctx *ctx, Slot &s, uint64_t n_entries) {
    if (s.d_tile_index && s.tile_cap >= n_entries) return PFAC_OK;
    if (s.d_tile_index) {
        HIP_TRY(ctx, hipStreamSynchronize(s.stream));
        HIP_TRY(ctx, hipFree(s.d_tile_index));
        s.d_tile_index = nullptr;
    }
    const uint64_t cap = n_entries < 4096 ? 4096 : n_entries + n_entries / 4;
    HIP_TRY(ctx, hipMalloc((void **)&s.d_tile_index, cap * 8));
    s.tile_cap = cap;
    return PFAC_OK;
}

// Tuning and test knobs (PFAC_FORCE_L2, PFAC_NWB, PFAC_FAULT, ...) are honoured ONLY in a process that opts in with
// PFAC_ENABLE_KNOBS=1 (the tests and the tools do): a production process cannot have its scans altered -- or a fault injected
// -- by a stray environment variable.
const char *knob(const char *name) {
    static const bool enabled = [] { const char *v = getenv("PFAC_ENABLE_KNOBS"); return v && *v && *v != '0'; }();
    return enabled ? getenv(name) : nullptr;
}
int env_int(const char *name, int dflt) {
    const char *v = knob(name);
    return v && *v ? atoi(v) : dflt;
}

static int max_lds(const pfac_ctx *ctx) {
    int v = ctx->lds_bytes_d;
    for (const StageLayout &L : ctx->lay) v = L.lds_bytes > v ? L.lds_bytes : v;
    return v;
}

// layout of pfac_ctx::d_d1: rows (d1_rows x 256 int32) | 256-byte row index | the depth-1 states | (packed rows) {r[], child
// mask} of the depth-2 states (8-byte aligned) | counter | column map | column bytes
size_t d1_off_r2(int d1_rows) { return align_up((size_t)d1_rows * 1024 + 256 + (size_t)d1_rows * 4, 8); }
size_t d1_off_col(int d1_rows) { return d1_off_r2(d1_rows) + (size_t)D1_N2_MAX * 8 + 16; }

int configure_kernel(pfac_ctx *ctx, const int32_t *s0_host) {
    // table bytes if staged in LDS: r (16-B rounded) + T
    const size_t tbytes = align_up((size_t)ctx->max_row * 4, 16) + (size_t)ctx->ht_size * 8;
    ctx->variant = tbytes <= (size_t)LDS_TABLE_MAX ? 0 : 1;
    if (knob("PFAC_FORCE_L2")) ctx->variant = 1;             // tuning knob: tables via L2 even if they fit LDS
    int halo = ctx->max_pat_len > 1 ? ctx->max_pat_len - 1 : 0;
    ctx->halo = (halo + 15) & ~15;
    // dense rows for the depth-1 states (children of the root), when there are few enough of them
    int fan = 0, rb = 0;
    int d1state[256];
    unsigned char d1idx[256];
    for (int i = 0; i < 256; i++) {
        d1idx[i] = 0;
        if (s0_host[i] >= 0) { d1state[fan] = s0_host[i]; d1idx[i] = (unsigned char)(fan < 255 ? fan : 255); fan++; rb = i; }
    }
    // (rows are built in device memory in full, 256 columns; LDS takes the columns that have an edge in some row)
    ctx->d1_rows = (fan >= 1 && fan <= 255 && !knob("PFAC_NO_D1")) ? fan : 0;
    ctx->d1_stride = 0; ctx->d1_lds_bytes = 0;
    // tables via L2 and PHF width >= 256: fused slots, one gather per step
    const bool fused = ctx->variant == 1 && ctx->width_bit >= 8 && !knob("PFAC_NO_FUSE");
    ctx->d1_n2 = 0;
    if (ctx->d_d1) { HIP_TRY(ctx, hipFree(ctx->d_d1)); ctx->d_d1 = nullptr; }
    if (ctx->d1_rows) {
        const size_t off_r2 = d1_off_r2(ctx->d1_rows);
        const size_t off_col = d1_off_col(ctx->d1_rows);
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_d1, off_col + 512));
        unsigned char *b = reinterpret_cast<unsigned char *>(ctx->d_d1);
        int *d_state = reinterpret_cast<int *>(b + (size_t)ctx->d1_rows * 1024 + 256);
        HIP_TRY(ctx, hipMemcpy(b + (size_t)ctx->d1_rows * 1024, d1idx, 256, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(d_state, d1state, (size_t)ctx->d1_rows * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(pfac_build_d1_kernel, dim3(ctx->d1_rows), dim3(256), 0, 0, d_state, ctx->d_r, ctx->d_T,
                           ctx->width_bit, ctx->ht_size, ctx->d_d1);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipDeviceSynchronize());
        // which bytes are the second byte of some pattern: the LDS rows keep those columns only (host copy of the
        // rows: at most 255 KiB); too many rows x columns for LDS: no dense level
        std::vector<int> rows((size_t)ctx->d1_rows * 256);
        HIP_TRY(ctx, hipMemcpy(rows.data(), ctx->d_d1, rows.size() * 4, hipMemcpyDeviceToHost));
        unsigned char colmap[256], colbyte[256];
        int ncols = 0;
        for (int c = 0; c < 256; c++) {
            bool used = false;
            for (int f = 0; f < ctx->d1_rows && !used; f++) used = rows[(size_t)f * 256 + c] >= 0;
            if (used) colbyte[ncols++] = (unsigned char)c;
        }
        const int stride = ncols < 256 ? ncols + 1 : 256;      // one more column, "no edge", unless every byte has its own
        for (int c = 0; c < 256; c++) colmap[c] = (unsigned char)(ncols & 255);
        for (int k = 0; k < ncols; k++) colmap[colbyte[k]] = (unsigned char)k;
        if ((size_t)ctx->d1_rows * stride * 4 > (size_t)D1_LDS_MAX) {
            HIP_TRY(ctx, hipFree(ctx->d_d1));
            ctx->d_d1 = nullptr;
            ctx->d1_rows = 0;
        } else {
            ctx->d1_stride = stride;
            ctx->d1_ncols = ncols;
            ctx->d1_lds_bytes = (int)align_up((size_t)(ctx->d1_rows + 1) * ctx->d1_stride * 4, 16);   // (+ one row without edges)
            HIP_TRY(ctx, hipMemcpy(b + off_col, colmap, 256, hipMemcpyHostToDevice));
            HIP_TRY(ctx, hipMemcpy(b + off_col + 256, colbyte, 256, hipMemcpyHostToDevice));
        }
        if (ctx->d1_rows && fused && ctx->state_num <= (1 << D1_STATE_BITS) && !knob("PFAC_NO_D1PACK")) {
            // how many depth-2 states are there?
            int n2 = 0;
            for (int v : rows) n2 += v >= 0;
            if (n2 >= 1 && n2 <= D1_N2_MAX) {
                int2 *r2 = reinterpret_cast<int2 *>(b + off_r2);
                int *counter = reinterpret_cast<int *>(r2 + D1_N2_MAX);
                HIP_TRY(ctx, hipMemset(counter, 0, 4));
                hipLaunchKernelGGL(pfac_pack_d1_kernel, dim3((unsigned)ctx->d1_rows), dim3(256), 0, 0, ctx->d_d1,
                                   ctx->d1_rows * 256, ctx->d_r, ctx->d_T, ctx->width_bit, ctx->ht_size, ctx->max_row, r2, counter);
                HIP_TRY(ctx, hipGetLastError());
                HIP_TRY(ctx, hipDeviceSynchronize());
                ctx->d1_n2 = n2;
            }
        }
    }
    ctx->shared_bytes = SH_D1 + ctx->d1_lds_bytes + (ctx->variant == 0 ? (int)align_up(tbytes, 16) : (int)align_up((size_t)(ctx->d1_n2 ? ctx->d1_n2 + 1 : 0) * 8, 16));
    if (fused && ctx->d1_rows == 0) ctx->shared_bytes += 1024;   // r[] of the depth-1 states by root byte
    ctx->sh_t0 = 0;
    if (ctx->d1_n2 > 0) { ctx->sh_t0 = ctx->shared_bytes; ctx->shared_bytes += 1024; }   // dense mode, second form: its root table
    // ---- level-2 filter: the 2-byte-prefix bitmap is built on the device from the uploaded tables; a single-edge
    // root with at most two grandchildren gets the bit-parallel form (their bytes), everything else the lookup form
    if (!ctx->d_bm2) HIP_TRY(ctx, hipMalloc((void **)&ctx->d_bm2, 256 * 32 + 256));   // bitmap + the second-byte flags
    hipLaunchKernelGGL(pfac_build_bm2_kernel, dim3(256), dim3(256), 0, 0, ctx->d_s0, ctx->d_r, ctx->d_T, ctx->width_bit,
                       ctx->ht_size, reinterpret_cast<unsigned long long *>(ctx->d_bm2));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipDeviceSynchronize());
    ctx->l2f_mode = 2;
    ctx->bm2_rows = 256;
    ctx->n_child = 0;
    ctx->child0 = ctx->child1 = 0;
    std::vector<unsigned char> bm2_host(256 * 32 + 256);
    HIP_TRY(ctx, hipMemcpy(bm2_host.data(), ctx->d_bm2, 256 * 32, hipMemcpyDeviceToHost));
    for (int c = 0; c < 256; c++) {                         // column OR: can byte c be a pattern's second byte?
        unsigned char any = 0;
        for (int b0 = 0; b0 < 256; b0++) any |= (unsigned char)(bm2_host[(size_t)b0 * 32 + (c >> 3)] >> (c & 7) & 1);
        bm2_host[256 * 32 + c] = any;
    }
    HIP_TRY(ctx, hipMemcpy(ctx->d_bm2 + 256 * 32, bm2_host.data() + 256 * 32, 256, hipMemcpyHostToDevice));
    bool any_fin1 = false;                                  // a 1-byte pattern: its survivors are kept whatever follows
    for (int i = 0; i < 256; i++) any_fin1 = any_fin1 || (s0_host[i] >= 0 && s0_host[i] < ctx->num_final);
    ctx->sec_filter = (fan != 1 && !any_fin1 && !knob("PFAC_NO_SECF")) ? 1 : 0;
    if (fan == 1) {
        const unsigned char *row = bm2_host.data() + (size_t)rb * 32;
        int nch = 0, ch[2] = {0, 0};
        for (int c = 0; c < 256; c++)
            if (row[c >> 3] >> (c & 7) & 1) { if (nch < 2) ch[nch] = c; nch++; }
        ctx->bm2_rows = 1;
        if (nch <= 2) {
            ctx->l2f_mode = 1;
            ctx->n_child = nch;
            ctx->child0 = (unsigned)ch[0] * 0x01010101u;
            ctx->child1 = (unsigned)ch[nch > 1 ? 1 : 0] * 0x01010101u;
        }
    }
    // multi-edge root without 1-byte patterns: when most (first byte, second byte) combinations of the flagged bytes
    // ARE 2-byte prefixes, the bitmap lookup would drop few of the survivors the second-byte flags let through -- they
    // all go to the walk instead (mode 3), whose dense depth-1 row is the same lookup done 64 lanes wide
    if (fan != 1 && ctx->sec_filter) {
        long pairs = 0, nsec = 0;
        for (int i = 0; i < 256 * 32; i++) pairs += __builtin_popcount(bm2_host[(size_t)i]);
        for (int c = 0; c < 256; c++) nsec += bm2_host[256 * 32 + c];
        if (nsec > 0 && pairs * 100 >= (long)fan * nsec * PFAC_PAIR_DENSITY_PCT) ctx->l2f_mode = 3;
    }
    const int l2f_env = env_int("PFAC_L2F", -1);          // 0: filter off; 2: lookup form even where the SWAR / flag-only form applies; 3: flags only
    if (l2f_env == 0) ctx->l2f_mode = 0;
    else if (l2f_env == 2) ctx->l2f_mode = 2;
    else if (l2f_env == 3 && fan != 1 && ctx->sec_filter) ctx->l2f_mode = 3;
    ctx->sh_bm2 = ctx->shared_bytes;
    if (ctx->l2f_mode == 2) ctx->shared_bytes += ctx->bm2_rows * 32;
    // knobs (tuning and tests), read once per table install
    ctx->spin_max = (unsigned)env_int("PFAC_SPIN_MAX", (int)SPIN_MAX);
    if (ctx->spin_max < 64) ctx->spin_max = 64;
    ctx->fault = (unsigned)env_int("PFAC_FAULT", 0);
    ctx->ticket_ways_knob = (unsigned)env_int("PFAC_TICKET_WAYS", 0);
    ctx->trace_file = knob("PFAC_TRACE") ? knob("PFAC_TRACE") : "";
    // The two-buffer layout fixes the number of waves; LDS that no further wave fits into goes to the staging buffers
    // (up to 1024 records per 4 KiB tile before it has to be walked a second time).  The three-buffer layout (emission
    // at the top of the round, see NBUF_MAX) is used when it keeps that many waves with room for >= CAPW3_MIN records.
    const int nwb_knob = env_int("PFAC_NWB", 0);
    for (int nb = 2; nb <= NBUF_MAX; nb++) {
        StageLayout &L = ctx->lay[nb - 2];
        L.nbuf = nb;
        L.pw_bytes = (int)align_up((size_t)PW_FIXED_1BUF + (size_t)nb * CAPW * 4 + ctx->halo, 16);
        int nwb = (LDS_TOTAL - ctx->shared_bytes) / L.pw_bytes + 1;     // compute waves + the coordinator (no LDS region)
        if (nb == 3) {                                                  // as many waves as with two buffers, smaller buffers if need be
            nwb = ctx->lay[0].waves_per_block;
            L.pw_bytes = (int)align_up((size_t)PW_FIXED_1BUF + ctx->halo, 16);
        }
        if (nwb > MAX_WAVES_PER_BLOCK) nwb = MAX_WAVES_PER_BLOCK;
        if (nwb_knob > 0 && nwb_knob < nwb) nwb = nwb_knob;
        if (nwb < 2) return fail(ctx, PFAC_E_INTERNAL, "LDS budget cannot hold one compute wave");
        L.waves_per_block = nwb;
        const int base_cap = nb == 3 ? 0 : CAPW;
        const int spare = (LDS_TOTAL - ctx->shared_bytes) / (nwb - 1) - L.pw_bytes;         // bytes per compute wave
        const int extra = spare > 0 ? (spare / (nb * 4)) & ~15 : 0;                         // records per staging buffer
        unsigned cap = (unsigned)(base_cap + ((nwb_knob && nb == 2) ? 0 : extra));
        if (cap > 1024u) cap = 1024u;
        L.pw_bytes += (int)(cap - base_cap) * nb * 4;
        L.stage_cap = cap;
        L.lds_bytes = ctx->shared_bytes + (nwb - 1) * L.pw_bytes;
        // one workgroup per CU: ask for more than half of the LDS so two never share a CU while another idles
        if (L.lds_bytes < LDS_TOTAL / 2 + 256) L.lds_bytes = LDS_TOTAL / 2 + 256;
    }
    const int lag_knob = env_int("PFAC_LAG", 0);          // 1 / 2: pin the emission lag (tests, A/B runs)
    ctx->lag2_ok = ctx->lay[1].stage_cap >= (unsigned)CAPW3_MIN && lag_knob != 1;
    ctx->lag2 = ctx->lag2_ok;
    ctx->lag_forced = lag_knob == 1 || lag_knob == 2;
    ctx->grid_blocks = ctx->n_cu;
    // root fan-out 1 -> exact SWAR root test (ROOT = 1), else LDS flag tables (ROOT = 0)
    ctx->root_mode = fan == 1 ? 1 : 0;
    ctx->root_byte = (unsigned)rb * 0x01010101u;
    ctx->root_state = s0_host[rb];
    // records are as wide as the automaton needs: 12 position bits + the final state
    ctx->rec_bytes = ctx->num_final <= 16 ? 2 : (ctx->num_final <= (1 << PACK_STATE_BITS) ? 4 : 8);
    const int rb_knob = env_int("PFAC_REC_BYTES", knob("PFAC_WIDE") ? 8 : 0);     // test knob: a WIDER form than needed
    if ((rb_knob == 4 || rb_knob == 8) && rb_knob > ctx->rec_bytes) ctx->rec_bytes = rb_knob;
    if (ctx->rec_bytes == 8) {                              // nothing is staged: every tile is written as it is walked
        ctx->lay[0].stage_cap = ctx->lay[1].stage_cap = 0u;
        ctx->lag2_ok = ctx->lag2 = false;
    }
    // dense-mode layout
    ctx->dense2 = fused && !knob("PFAC_NO_NW4") && !knob("PFAC_NO_DENSE2") && ctx->d1_rows > 0 && ctx->d1_n2 > 0 &&
                  ctx->num_final <= 65535 && ctx->rec_bytes == 4;
    ctx->d2log_cap = D2_LOG_CAP;
    const int d2cap_knob = env_int("PFAC_D2_LOGCAP", 0);     // test knob: tiles with more records than this take the fallback pass
    if (d2cap_knob >= 16 && (unsigned)d2cap_knob < D2_LOG_CAP) ctx->d2log_cap = (unsigned)d2cap_knob;
    ctx->pw_bytes_d = (int)align_up((size_t)(ctx->dense2 ? PW_FIXED_DENSE2 : PW_FIXED_DENSE) + ctx->halo, 16);
    int nwd = (LDS_TOTAL - ctx->shared_bytes) / ctx->pw_bytes_d + 1;
    if (nwd > MAX_WAVES_PER_BLOCK) nwd = MAX_WAVES_PER_BLOCK;
    ctx->waves_per_block_d = nwd;
    ctx->lds_bytes_d = ctx->shared_bytes + (nwd - 1) * ctx->pw_bytes_d;
    if (ctx->lds_bytes_d < LDS_TOTAL / 2 + 256) ctx->lds_bytes_d = LDS_TOTAL / 2 + 256;
    ctx->stage_cap_d = (nwd >= 4 && ctx->lay[0].stage_cap) ? (unsigned)CAPW_DENSE : 0u;   // 0: dense mode unavailable
    ctx->dense_forced = knob("PFAC_DENSE") ? atoi(knob("PFAC_DENSE")) : -1;
    ctx->dense = ctx->dense_forced == 1 && ctx->stage_cap_d;
    const bool w8 = ctx->width_bit == 8;
    if (ctx->d_T4_alloc) { HIP_TRY(ctx, hipFree(ctx->d_T4_alloc)); ctx->d_T4_alloc = ctx->d_T4 = nullptr; }
    ctx->rn_bias = 0;
    if (fused) {
        std::vector<int> r_host((size_t)ctx->max_row);
        HIP_TRY(ctx, hipMemcpy(r_host.data(), ctx->d_r, r_host.size() * 4, hipMemcpyDeviceToHost));
        long long lo = 0, hi = ctx->ht_size;
        for (int v : r_host) {
            lo = v < lo ? v : lo;
            hi = (long long)v + (1 << ctx->width_bit) > hi ? (long long)v + (1 << ctx->width_bit) : hi;
        }
        if (hi - lo >= (1ll << 28)) return fail(ctx, PFAC_E_ARG, "table image: more than 2^28 hash slots");
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_T4_alloc, (size_t)(hi - lo) * sizeof(int4)));
        ctx->d_T4 = ctx->d_T4_alloc - lo;
        ctx->rn_bias = (int)-lo;
        hipLaunchKernelGGL(pfac_fuse_kernel, dim3(256), dim3(256), 0, 0, ctx->d_T, ctx->d_r, ctx->width_bit, ctx->ht_size,
                           ctx->max_row, ctx->d_T4, (int)lo, (int)hi);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipDeviceSynchronize());
    }
    constexpr int FNW = PFAC_SPARSE_FUSED_NW;   // walks per lane of the sparse-mode kernels on fused L2 tables
    const void *k[4][2][2] = {
        {{(const void *)pfac_scan_kernel<false, false, 0, false, 2>, (const void *)pfac_scan_kernel<false, false, 1, false, 2>},
         {(const void *)pfac_scan_kernel<false, true, 0, false, 2>, (const void *)pfac_scan_kernel<false, true, 1, false, 2>}},
        {{(const void *)pfac_scan_kernel<true, false, 0, false, 1>, (const void *)pfac_scan_kernel<true, false, 1, false, 1>},
         {(const void *)pfac_scan_kernel<true, true, 0, false, 1>, (const void *)pfac_scan_kernel<true, true, 1, false, 1>}},
        {{(const void *)pfac_scan_kernel<false, false, 0, true, FNW>, (const void *)pfac_scan_kernel<false, false, 1, true, FNW>},
         {(const void *)pfac_scan_kernel<false, true, 0, true, FNW>, (const void *)pfac_scan_kernel<false, true, 1, true, FNW>}},
        {{(const void *)pfac_scan_kernel<false, false, 0, true, 4>, (const void *)pfac_scan_kernel<false, false, 1, true, 4>},
         {(const void *)pfac_scan_kernel<false, true, 0, true, 4>, (const void *)pfac_scan_kernel<false, true, 1, true, 4>}}};
    // (three walks per lane for the sparse fused kernels -- one round per tile of the 75 840-pattern set on random bytes
    // instead of 1.45 -- measured 3 % slower than two: 112 VGPRs and the longer round cost more than the second round)
    ctx->kernel = k[ctx->variant == 0 ? 1 : (fused ? 2 : 0)][w8 ? 1 : 0][ctx->root_mode];
    // dense mode on fused L2 tables: four walks per lane (needs <= MAX_WAVES_NW4 waves per workgroup)
    ctx->kernel_d = ctx->kernel;
    {
        const void *k3[3][2][2] = {
            {{(const void *)pfac_scan_kernel<false, false, 0, false, 2, 3>, (const void *)pfac_scan_kernel<false, false, 1, false, 2, 3>},
             {(const void *)pfac_scan_kernel<false, true, 0, false, 2, 3>, (const void *)pfac_scan_kernel<false, true, 1, false, 2, 3>}},
            {{(const void *)pfac_scan_kernel<true, false, 0, false, 1, 3>, (const void *)pfac_scan_kernel<true, false, 1, false, 1, 3>},
             {(const void *)pfac_scan_kernel<true, true, 0, false, 1, 3>, (const void *)pfac_scan_kernel<true, true, 1, false, 1, 3>}},
            {{(const void *)pfac_scan_kernel<false, false, 0, true, FNW, 3>, (const void *)pfac_scan_kernel<false, false, 1, true, FNW, 3>},
             {(const void *)pfac_scan_kernel<false, true, 0, true, FNW, 3>, (const void *)pfac_scan_kernel<false, true, 1, true, FNW, 3>}}};
        ctx->kernel3 = k3[ctx->variant == 0 ? 1 : (fused ? 2 : 0)][w8 ? 1 : 0][ctx->root_mode];
        HIP_TRY(ctx, hipFuncSetAttribute(ctx->kernel3, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds(ctx)));
    }
    if (fused && !knob("PFAC_NO_NW4")) {
        ctx->kernel_d = k[3][w8 ? 1 : 0][ctx->root_mode];
        if (!ctx->dense2 && ctx->waves_per_block_d > MAX_WAVES_NW4) {
            ctx->waves_per_block_d = MAX_WAVES_NW4;
            ctx->lds_bytes_d = ctx->shared_bytes + (MAX_WAVES_NW4 - 1) * ctx->pw_bytes_d;
            if (ctx->lds_bytes_d < LDS_TOTAL / 2 + 256) ctx->lds_bytes_d = LDS_TOTAL / 2 + 256;
        }
        HIP_TRY(ctx, hipFuncSetAttribute(ctx->kernel_d, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds(ctx)));
    }
    HIP_TRY(ctx, hipFuncSetAttribute(ctx->kernel, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds(ctx)));
    if (knob("PFAC_VERBOSE"))
        fprintf(stderr, "pfac: variant %d fused %d shared LDS %d B; sparse: %d waves x %d B; dense%s: %d waves x %d B; dense rows %d x %d, %d depth-2 states, level-2 filter mode %d\n",
                ctx->variant, (int)fused, ctx->shared_bytes, ctx->lay[0].waves_per_block, ctx->lay[0].pw_bytes, ctx->dense2 ? " (second form)" : "",
                ctx->waves_per_block_d, ctx->pw_bytes_d, ctx->d1_rows, ctx->d1_stride, ctx->d1_n2, ctx->l2f_mode);
    return PFAC_OK;
}

// d_blob: device-resident blob image; hdr: its first 16 words on the host
int install_table(pfac_ctx *ctx, const int *d_blob, const int32_t *hdr, size_t n_words, hipStream_t stream) {
    if (hdr[0] != PFAC_BLOB_MAGIC || hdr[1] != PFAC_BLOB_VERSION) return fail(ctx, PFAC_E_ARG, "not a PFAC table image");
    const int width = hdr[2], wbit = hdr[3], num_final = hdr[5], state_num = hdr[6], max_pat_len = hdr[7];
    const int max_row = hdr[8], ht_size = hdr[9];
    if (width < 1 || width > 4096 || (1 << wbit) != width || num_final < 0 || state_num < num_final + 2 ||
        max_row < 1 || ht_size < 1 || max_pat_len < 0 || max_pat_len > HALO_MAX - 1 ||
        (size_t)PFAC_BLOB_HEADER_WORDS + 256 + (size_t)max_row + 2 * (size_t)ht_size + (size_t)num_final > n_words)
        return fail(ctx, PFAC_E_ARG, "inconsistent PFAC table image header");
    if ((int64_t)max_row < ((int64_t)state_num * 256 >> wbit) + 1)
        return fail(ctx, PFAC_E_ARG, "table image: r[] shorter than state_num*256/width+1");
    const size_t off_r = 256 * 4;
    const size_t off_T = align_up(off_r + (size_t)max_row * 4, 16);
    const size_t off_id = off_T + (size_t)ht_size * 8;
    const size_t total = align_up(off_id + (size_t)num_final * 4, 16) + 16;
    if (ctx->d_tab) { HIP_TRY(ctx, hipFree(ctx->d_tab)); ctx->d_tab = nullptr; }
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_tab, total));
    ctx->tab_bytes = total;
    unsigned char *base = reinterpret_cast<unsigned char *>(ctx->d_tab);
    ctx->d_s0 = reinterpret_cast<int *>(base);
    ctx->d_r = reinterpret_cast<int *>(base + off_r);
    ctx->d_T = reinterpret_cast<int2 *>(base + off_T);
    ctx->d_idmap = reinterpret_cast<int *>(base + off_id);
    int *d_bad = reinterpret_cast<int *>(base + total - 16);
    HIP_TRY(ctx, hipMemsetAsync(d_bad, 0, 4, stream));
    hipLaunchKernelGGL(pfac_repack_kernel, dim3(256), dim3(256), 0, stream, d_blob, ctx->d_s0, ctx->d_r, ctx->d_T,
                       ctx->d_idmap, max_row, ht_size, num_final, state_num, width, d_bad);
    HIP_TRY(ctx, hipGetLastError());
    int bad = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    if (bad) {
        ctx->have_table = false;
        return fail(ctx, PFAC_E_ARG, "table image: a displacement or a state points outside the tables");
    }
    ctx->width_bit = wbit; ctx->num_final = num_final; ctx->max_pat_len = max_pat_len;
    ctx->max_row = max_row; ctx->ht_size = ht_size; ctx->state_num = state_num;
    ctx->have_table = true;
    int32_t s0_host[256];
    HIP_TRY(ctx, hipMemcpy(s0_host, ctx->d_s0, sizeof s0_host, hipMemcpyDeviceToHost));
    return configure_kernel(ctx, s0_host);
}

}  // namespace

// ---------------------------------------------------------------------------
extern "C" {

const char *pfac_last_error(const pfac_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int pfac_device_count(int *n) {
    if (!n) return fail(nullptr, PFAC_E_ARG, "null argument");
    *n = 0;
    hipError_t e = hipGetDeviceCount(n);
    if (e != hipSuccess) { *n = 0; return fail(nullptr, PFAC_E_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    return PFAC_OK;
}

int pfac_ctx_create(int device, int n_streams, pfac_ctx **out) {
    if (!out || n_streams < 1 || n_streams > 64) return fail(nullptr, PFAC_E_ARG, "bad argument to pfac_ctx_create");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return fail(nullptr, PFAC_E_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(nullptr, PFAC_E_NO_DEVICE, "device index out of range");
    pfac_ctx *ctx = new pfac_ctx();
    ctx->device = device;
    USE_DEVICE(ctx);
    hipDeviceProp_t prop;
    HIP_TRY(ctx, hipGetDeviceProperties(&prop, device));
    ctx->n_cu = prop.multiProcessorCount;
    ctx->slots.resize(n_streams);
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (auto &s : ctx->slots) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&s.own_stream, hipStreamNonBlocking));
        s.stream = s.own_stream;
        HIP_TRY(ctx, hipHostMalloc((void **)&s.h_ctl, 64, hipHostMallocMapped));
        memset(s.h_ctl, 0, 64);
        HIP_TRY(ctx, hipHostGetDevicePointer((void **)&s.d_res, s.h_ctl, 0));
        HIP_TRY(ctx, hipMalloc((void **)&s.d_sum, 16));
        HIP_TRY(ctx, hipEventCreate(&s.ev0));
        HIP_TRY(ctx, hipEventCreate(&s.ev1));
        HIP_TRY(ctx, hipEventCreateWithFlags(&s.ev_h2d, hipEventDisableTiming));
    }
    *out = ctx;
    return PFAC_OK;
}

void pfac_ctx_destroy(pfac_ctx *ctx) {
    if (!ctx) return;
    DeviceGuard device_guard_(ctx->device);
    for (auto &s : ctx->slots) {
        if (s.own_stream) (void)hipStreamSynchronize(s.own_stream);
        if (s.d_input) (void)hipFree(s.d_input);
        if (s.d_records) (void)hipFree(s.d_records);
        if (s.d_ctl) (void)hipFree(s.d_ctl);
        if (s.d_tile_index) (void)hipFree(s.d_tile_index);
        if (s.d_d2log) (void)hipFree(s.d_d2log);
        if (s.d_gsum) (void)hipFree(s.d_gsum);
        if (s.d_text) (void)hipFree(s.d_text);
        if (s.d_wide) (void)hipFree(s.d_wide);
        if (s.d_dbg) (void)hipFree(s.d_dbg);
        if (s.d_sum) (void)hipFree(s.d_sum);
        if (s.h_ctl) (void)hipHostFree(s.h_ctl);
        if (s.ev0) (void)hipEventDestroy(s.ev0);
        if (s.ev1) (void)hipEventDestroy(s.ev1);
        if (s.ev_h2d) (void)hipEventDestroy(s.ev_h2d);
        if (s.own_stream) (void)hipStreamDestroy(s.own_stream);
    }
    if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
    if (ctx->d_tab) (void)hipFree(ctx->d_tab);
    if (ctx->d_d1) (void)hipFree(ctx->d_d1);
    if (ctx->d_T4_alloc) (void)hipFree(ctx->d_T4_alloc);
    if (ctx->d_bm2) (void)hipFree(ctx->d_bm2);
    delete ctx;
}

int pfac_table_upload(pfac_ctx *ctx, const int32_t *blob, size_t n_words) {
    if (!ctx || !blob || n_words < PFAC_BLOB_HEADER_WORDS) return fail(ctx, PFAC_E_ARG, "bad argument to pfac_table_upload");
    std::lock_guard<std::mutex> lk(ctx->mu);
    USE_DEVICE(ctx);
    int *d_blob = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&d_blob, n_words * 4));
    hipError_t e = hipMemcpy(d_blob, blob, n_words * 4, hipMemcpyHostToDevice);   // master_kernel.cu:365-383
    int rc = e == hipSuccess ? install_table(ctx, d_blob, blob, n_words, ctx->slots[0].stream)
                             : fail(ctx, PFAC_E_HIP, std::string("hipMemcpy(table): ") + hipGetErrorString(e));
    (void)hipFree(d_blob);
    return rc;
}

int pfac_table_upload_device(pfac_ctx *ctx, const void *d_blob, size_t n_words, void *stream_handle) {
    if (!ctx || !d_blob || n_words < PFAC_BLOB_HEADER_WORDS) return fail(ctx, PFAC_E_ARG, "bad argument to pfac_table_upload_device");
    std::lock_guard<std::mutex> lk(ctx->mu);
    USE_DEVICE(ctx);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_handle);
    int32_t hdr[PFAC_BLOB_HEADER_WORDS];
    HIP_TRY(ctx, hipMemcpyAsync(hdr, d_blob, sizeof hdr, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return install_table(ctx, reinterpret_cast<const int *>(d_blob), hdr, n_words, st);
}

int pfac_host_alloc(void **p, size_t n_bytes) {
    if (!p) return fail(nullptr, PFAC_E_ARG, "null argument");
    hipError_t e = hipHostMalloc(p, n_bytes ? n_bytes : 1, hipHostMallocPortable);
    if (e != hipSuccess) { *p = nullptr; return fail(nullptr, PFAC_E_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
    return PFAC_OK;
}
void pfac_host_free(void *p) { if (p) (void)hipHostFree(p); }

int pfac_host_register(void *p, size_t n_bytes) {
    if (!p || !n_bytes) return fail(nullptr, PFAC_E_ARG, "bad argument to pfac_host_register");
    hipError_t e = hipHostRegister(p, n_bytes, hipHostRegisterPortable);     // (usable from every device's context)
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, PFAC_E_HIP, std::string("hipHostRegister: ") + hipGetErrorString(e)); }
    return PFAC_OK;
}
int pfac_host_unregister(void *p) {
    if (!p) return fail(nullptr, PFAC_E_ARG, "null argument");
    hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, PFAC_E_HIP, std::string("hipHostUnregister: ") + hipGetErrorString(e)); }
    return PFAC_OK;
}

int pfac_slot_reserve(pfac_ctx *ctx, int slot, uint64_t input_bytes, uint64_t record_capacity) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    USE_DEVICE(ctx);
    Slot &s = ctx->slots[slot];
    if (input_bytes > s.input_cap) {
        if (s.d_input) { HIP_TRY(ctx, hipFree(s.d_input)); s.d_input = nullptr; s.input_cap = 0; }
        const uint64_t cap = align_up(input_bytes, WTILE) + HALO_MAX + 256;
        HIP_TRY(ctx, hipMalloc((void **)&s.d_input, cap));
        s.input_cap = cap;
    }
    if (record_capacity > s.record_cap) {
        if (s.d_records) { HIP_TRY(ctx, hipFree(s.d_records)); s.d_records = nullptr; s.record_cap = 0; }
        HIP_TRY(ctx, hipMalloc((void **)&s.d_records, record_capacity * sizeof(pfac_record)));
        s.record_cap = record_capacity;
    }
    return PFAC_OK;
}

void *pfac_slot_input(pfac_ctx *ctx, int slot) { return check_slot(ctx, slot) ? nullptr : ctx->slots[slot].d_input; }
void *pfac_slot_records(pfac_ctx *ctx, int slot) { return check_slot(ctx, slot) ? nullptr : ctx->slots[slot].d_records; }
void *pfac_slot_stream(pfac_ctx *ctx, int slot) { return check_slot(ctx, slot) ? nullptr : (void *)ctx->slots[slot].stream; }

int pfac_slot_set_stream(pfac_ctx *ctx, int slot, void *stream_handle) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    Slot &s = ctx->slots[slot];
    hipStream_t ns = stream_handle ? reinterpret_cast<hipStream_t>(stream_handle) : s.own_stream;
    if (ns != s.stream && s.pending) HIP_TRY(ctx, hipStreamSynchronize(s.stream));   // a scan in flight zeroes the next scan's control words
    s.stream = ns;
    return PFAC_OK;
}

int pfac_slot_h2d(pfac_ctx *ctx, int slot, const void *host, uint64_t n_bytes, uint64_t dst_offset) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    Slot &s = ctx->slots[slot];
    if (!host || dst_offset + n_bytes > s.input_cap) return fail(ctx, PFAC_E_ARG, "pfac_slot_h2d: range exceeds the reserved input buffer");
    USE_DEVICE(ctx);
    // on the context's copy stream, behind whatever the slot's stream still does with the buffer (its last scan reads it);
    // the slot's stream then waits for the copy: same ordering as a copy on the slot's stream, without the gaps
    if (s.scanned) HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, s.ev1, 0));
    HIP_TRY(ctx, hipMemcpyAsync(s.d_input + dst_offset, host, n_bytes, hipMemcpyHostToDevice, ctx->copy_stream));
    HIP_TRY(ctx, hipEventRecord(s.ev_h2d, ctx->copy_stream));
    HIP_TRY(ctx, hipStreamWaitEvent(s.stream, s.ev_h2d, 0));
    s.h2d_issued = true;
    return PFAC_OK;
}

int pfac_slot_h2d_done(pfac_ctx *ctx, int slot) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    USE_DEVICE(ctx);
    const hipError_t e = hipEventQuery(ctx->slots[slot].ev_h2d);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
    return fail(ctx, PFAC_E_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
}

int pfac_slot_h2d_wait(pfac_ctx *ctx, int slot) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    USE_DEVICE(ctx);
    HIP_TRY(ctx, hipEventSynchronize(ctx->slots[slot].ev_h2d));   // (an event never recorded counts as complete)
    return PFAC_OK;
}

int pfac_scan_async(pfac_ctx *ctx, int slot, const void *d_input, uint64_t n_owned, uint64_t n_avail,
                    void *d_records, uint64_t capacity) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (!ctx->have_table) return fail(ctx, PFAC_E_STATE, "pfac_scan_async before a table upload");
    Slot &s = ctx->slots[slot];
    const unsigned char *in = d_input ? static_cast<const unsigned char *>(d_input) : s.d_input;
    if (!d_records) { d_records = s.d_records; capacity = s.record_cap; }
    if (!in) return fail(ctx, PFAC_E_ARG, "pfac_scan_async: no input buffer");
    if (((uintptr_t)in & 15) != 0) return fail(ctx, PFAC_E_ARG, "pfac_scan_async: input pointer must be 16-byte aligned");
    if (n_owned > n_avail || n_owned > (1ull << 32)) return fail(ctx, PFAC_E_ARG, "pfac_scan_async: need n_owned <= n_avail and n_owned <= 2^32");
    if (!d_input && n_avail > s.input_cap) return fail(ctx, PFAC_E_ARG, "pfac_scan_async: n_avail exceeds the reserved input buffer");
    if (((uintptr_t)d_records & 15) != 0) return fail(ctx, PFAC_E_ARG, "pfac_scan_async: record buffer must be 16-byte aligned");
    if (!d_records && capacity) return fail(ctx, PFAC_E_ARG, "pfac_scan_async: no record buffer");
    USE_DEVICE(ctx);
    const uint64_t n_tiles = (n_owned + WTILE - 1) / WTILE;
    if (s.pending) HIP_TRY(ctx, hipStreamSynchronize(s.stream));   // the previous scan of this slot still owns h_ctl
    s.last_cap = capacity;
    s.scanned = true;
    s.pending = true;
    for (int i = 0; i < 7; i++) s.h_ctl[i] = 0;        // result words (the kernel writes them through the host mapping)
    // staging mode of this launch (see pfac_scan_finish for the adaptation)
    const bool dense = ctx->dense && ctx->stage_cap_d;
    const StageLayout &L = ctx->sparse();
    const int wpb = dense ? ctx->waves_per_block_d : L.waves_per_block;
    const int lds_bytes = dense ? ctx->lds_bytes_d : L.lds_bytes;
    s.last_dense = dense;
    s.last_tiles = n_tiles;
    s.last_rec_bytes = ctx->rec_bytes;
    s.last_records = d_records;
    // The control header (ticket counters, flags, heap cursor) must start at zero.  The slot has two: every scan
    // zeroes, in its own prologue, the other one for the scan after it, so back-to-back scans need no memset.
    rc = ensure_ctl(ctx, s);
    if (rc) return rc;
    rc = ensure_tiles(ctx, s, n_tiles + 1);
    if (rc) return rc;
    const uint64_t n_batches = (n_tiles + wpb - 2) / (wpb - 1);
    unsigned *const cur = s.d_ctlbuf[s.flip], *const nxt = s.d_ctlbuf[1 - s.flip];
    if (n_tiles > 0 && !s.clean[s.flip]) HIP_TRY(ctx, hipMemsetAsync(cur, 0, CTL_REGION, s.stream));
    if (n_tiles == 0) HIP_TRY(ctx, hipEventRecord(s.ev0, s.stream));
    if (n_tiles > 0) {
        ScanArgs a;
        a.in = in; a.n_owned = n_owned; a.n_avail = n_avail;
        a.out = d_records; a.out_cap = capacity;
        a.tile_index = s.d_tile_index;
        a.rec_bytes = (unsigned)ctx->rec_bytes;
        a.l2f_mode = ctx->l2f_mode; a.child0 = ctx->child0; a.child1 = ctx->child1; a.n_child = ctx->n_child;
        a.bm2 = ctx->d_bm2; a.bm2_rows = ctx->bm2_rows; a.sh_bm2 = ctx->sh_bm2; a.sh_t0 = ctx->sh_t0;
        a.sec2 = ctx->d_bm2 + 256 * 32; a.sec_filter = ctx->sec_filter;
        a.spin_max = ctx->spin_max; a.fault = ctx->fault;
        a.s0 = ctx->d_s0; a.r = ctx->d_r; a.T = ctx->d_T; a.T4 = ctx->d_T4_alloc; a.rn_bias = ctx->rn_bias;
        a.r_words = ctx->max_row; a.t_entries = ctx->ht_size;
        a.ht_size = ctx->ht_size; a.wbit = ctx->width_bit; a.num_final = ctx->num_final;
        a.halo = ctx->halo;
        a.shared_bytes = ctx->shared_bytes; a.pw_bytes = dense ? ctx->pw_bytes_d : L.pw_bytes;
        a.d1 = ctx->d_d1; a.d1_rows = ctx->d1_rows; a.d1_stride = ctx->d1_stride; a.d1_ncols = ctx->d1_ncols; a.d1_lds_bytes = ctx->d1_lds_bytes;
        a.d1_colmap = ctx->d_d1 ? reinterpret_cast<const unsigned char *>(ctx->d_d1) + d1_off_col(ctx->d1_rows) : nullptr;
        a.d1_colbyte = a.d1_colmap ? a.d1_colmap + 256 : nullptr;
        a.d1_n2 = ctx->d1_n2;
        a.d1r2 = ctx->d1_n2 ? reinterpret_cast<const int2 *>(reinterpret_cast<const unsigned char *>(ctx->d_d1) + d1_off_r2(ctx->d1_rows))
                            : nullptr;
        a.d1idx = ctx->d_d1 ? reinterpret_cast<const unsigned char *>(ctx->d_d1) + (size_t)ctx->d1_rows * 1024 : nullptr;
        a.root_byte = ctx->root_byte;
        a.root_state = ctx->root_state;
        a.stage_cap = dense ? ctx->stage_cap_d : L.stage_cap;
        a.nbuf = dense ? 1u : (unsigned)L.nbuf;
        a.dense2 = dense && ctx->dense2 ? 1 : 0;
        a.d2log = nullptr; a.d2log_cap = ctx->d2log_cap;
        if (a.dense2) {
            a.stage_cap = 0;                   // (its LDS carve has no staging buffer: a tile it gives up on is counted, then written directly)
            const size_t words = (size_t)ctx->grid_blocks * (size_t)(wpb - 1) * ctx->d2log_cap;
            if (s.d2log_words < words) {
                if (s.d_d2log) { HIP_TRY(ctx, hipStreamSynchronize(s.stream)); HIP_TRY(ctx, hipFree(s.d_d2log)); s.d_d2log = nullptr; s.d2log_words = 0; }
                HIP_TRY(ctx, hipMalloc((void **)&s.d_d2log, words * 4));
                s.d2log_words = words;
            }
            a.d2log = s.d_d2log;
        }
        a.sparse_cap = ctx->lay[0].stage_cap;
        a.small_cap = ctx->lag2_ok ? ctx->lay[1].stage_cap : ctx->lay[0].stage_cap;
        a.n_tiles = (unsigned)n_tiles;
        a.ctl = cur;
        a.zero_next = reinterpret_cast<uint4 *>(nxt);
        a.zero_vec = (unsigned)(CTL_REGION / 16);
        a.res = s.d_res;
        a.dbg = nullptr;
#ifdef PFAC_TRACE_BUILD
        if (!ctx->trace_file.empty()) {
            if (!s.d_dbg) HIP_TRY(ctx, hipMalloc((void **)&s.d_dbg, 8 * 64 * 32 * 8));
            HIP_TRY(ctx, hipMemsetAsync(s.d_dbg, 0, 8 * 64 * 32 * 8, s.stream));
            a.dbg = s.d_dbg;
        }
#endif
        const uint64_t want = n_batches;
        uint64_t grid = (uint64_t)ctx->grid_blocks < want ? (uint64_t)ctx->grid_blocks : want;
        a.ticket_ways = grid < TICKET_WAYS ? (unsigned)grid : TICKET_WAYS;
        if (ctx->ticket_ways_knob >= 1 && ctx->ticket_ways_knob < a.ticket_ways) a.ticket_ways = ctx->ticket_ways_knob;
        // heap chunk: 1/32 of an even share of the record array per workgroup -- the current and the spare chunk of
        // every workgroup can stay unfilled at the end, i.e. at most 1/16 of the capacity; record arrays too small for
        // chunks of 1024 records get exact allocations (one atomic per batch)
        uint64_t chunk = (capacity / (32 * grid)) & ~3ull;
        if (chunk > (1u << 22)) chunk = 1u << 22;
        a.chunk = (chunk >= 1024 && !dense) ? (unsigned)chunk : 0u;      // (dense mode: every tile takes its own space)
        void *kargs[] = {&a};
        // the slot's two events ride on the dispatch itself (start / stop of THIS kernel): no barrier packets of their own
        // in front of and behind every scan
        HIP_TRY(ctx, hipExtLaunchKernel(dense ? ctx->kernel_d : (ctx->lag2 ? ctx->kernel3 : ctx->kernel), dim3((unsigned)grid),
                                        dim3(WAVE * wpb), kargs, (size_t)lds_bytes, s.stream, s.ev0, s.ev1, 0));
        s.clean[s.flip] = false;               // used by this scan
        s.clean[1 - s.flip] = true;            // zeroed by this scan
        s.flip = 1 - s.flip;
    }
    if (n_tiles == 0) HIP_TRY(ctx, hipEventRecord(s.ev1, s.stream));
    return PFAC_OK;
}

int pfac_scan_finish(pfac_ctx *ctx, int slot, uint64_t *n_matches) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    Slot &s = ctx->slots[slot];
    if (!s.scanned) return fail(ctx, PFAC_E_STATE, "pfac_scan_finish without a scan");
    USE_DEVICE(ctx);
    // wait for THIS scan's end event, not for the stream: another slot may share the stream (launch pipelining), and
    // its scan -- enqueued after this one -- should keep the GPU busy while the host reads this result
    HIP_TRY(ctx, hipEventSynchronize(s.ev1));
    s.pending = false;
    const uint64_t total = ((uint64_t)s.h_ctl[1] << 32) | s.h_ctl[0];
    s.last_total = total;
    s.last_used = ((uint64_t)s.h_ctl[5] << 32) | s.h_ctl[4];
    if (n_matches) *n_matches = total;
#ifdef PFAC_TRACE_BUILD
    if (s.d_dbg && !ctx->trace_file.empty()) {
        std::vector<unsigned long long> h(8 * 64 * 32);
        if (hipMemcpy(h.data(), s.d_dbg, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE *f = fopen(ctx->trace_file.c_str(), "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
        }
    }
#endif
    if (s.h_ctl[2] != 0) {
        s.clean[0] = s.clean[1] = false;       // whatever state the control headers are in: zero them before the next scan
        return fail(ctx, PFAC_E_INTERNAL, "scan kernel reported a timeout (flags " + std::to_string(s.h_ctl[2]) +
                                          ": 4 arrivals, 8 record base, 16 batch ring)");
    }
    // Staging mode for the NEXT scans of this context: when more than a quarter of the tiles held more matches
    // than the two-buffer staging area takes, go dense (one big buffer, emitted at once, no second walk); go back
    // when fewer than 1/16 do.  PFAC_DENSE=0/1 pins the mode.  Likewise between the three- and the two-buffer layout,
    // on the tiles above the (smaller) three-buffer capacity: 1/16 of the tiles walked twice cost what the earlier
    // emission gains.  PFAC_LAG=1/2 pins that.
    if (ctx->dense_forced < 0 && ctx->stage_cap_d && s.last_tiles >= 64) {
        const uint64_t ovf = s.h_ctl[3];
        if (!ctx->dense && ovf * 4 > s.last_tiles) ctx->dense = true;
        else if (ctx->dense && ovf * 16 < s.last_tiles) ctx->dense = false;
    }
    if (ctx->lag2_ok && !ctx->lag_forced && s.last_tiles >= 64) {
        const uint64_t ovf2 = s.h_ctl[6];
        if (ctx->lag2 && ovf2 * 16 > s.last_tiles) ctx->lag2 = false;
        else if (!ctx->lag2 && ovf2 * 64 < s.last_tiles) ctx->lag2 = true;
    }
    if (s.last_used > s.last_cap)
        return fail(ctx, PFAC_E_OVERFLOW, "record array too small: " + std::to_string(total) + " matches need " +
                                              std::to_string(s.last_used) + " records of capacity (pfac_scan_capacity_hint)");
    return PFAC_OK;
}

int pfac_scan_elapsed_ms(pfac_ctx *ctx, int slot, float *ms) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (!ms) return fail(ctx, PFAC_E_ARG, "null argument");
    Slot &s = ctx->slots[slot];
    if (!s.scanned) return fail(ctx, PFAC_E_STATE, "no scan to time");
    HIP_TRY(ctx, hipEventSynchronize(s.ev1));
    HIP_TRY(ctx, hipEventElapsedTime(ms, s.ev0, s.ev1));
    return PFAC_OK;
}

// Records [first, first+n) of the slot's last scan, in (position, pattern length) order -> pfac_record at d_out
// (device), on the slot's stream: prefix over the tile index, then a copy out of the heap.
static int ensure_gsum(pfac_ctx *ctx, Slot &s, unsigned n_groups) {       // n_groups prefixes + the grand total behind them
    if ((uint64_t)n_groups + 1 <= s.gsum_cap) return PFAC_OK;
    if (s.d_gsum) { HIP_TRY(ctx, hipStreamSynchronize(s.stream)); HIP_TRY(ctx, hipFree(s.d_gsum)); s.d_gsum = nullptr; }
    const uint64_t cap = n_groups < 4096 ? 4097 : (uint64_t)n_groups + n_groups / 4 + 1;
    HIP_TRY(ctx, hipMalloc((void **)&s.d_gsum, cap * 8));
    s.gsum_cap = cap;
    return PFAC_OK;
}

static int expand_records(pfac_ctx *ctx, Slot &s, const void *src, uint64_t first, uint64_t n, pfac_record *d_out) {
    // records that do not exist, or that the last scan could not write, are never delivered as if they did: the copy
    // kernel skips them and the caller would read whatever its buffer held before
    if (s.pending) return fail(ctx, PFAC_E_STATE, "records requested before pfac_scan_finish");
    if (first + n > s.last_total) return fail(ctx, PFAC_E_ARG, "records [first, first + n) exceed the scan's match count");
    if (s.last_used > s.last_cap) return fail(ctx, PFAC_E_OVERFLOW, "the slot's last scan overflowed its record heap: scan again with a larger one");
    if (n == 0 || s.last_tiles == 0) return PFAC_OK;
    const unsigned n_groups = (unsigned)((s.last_tiles + XGROUP - 1) / XGROUP);
    int rc = ensure_gsum(ctx, s, n_groups);
    if (rc) return rc;
    const unsigned gblocks = (n_groups + 3) / 4;           // four waves (groups) per 256-thread block
    hipLaunchKernelGGL(pfac_tix_group_sum_kernel, dim3(gblocks), dim3(256), 0, s.stream, s.d_tile_index,
                       (unsigned long long)s.last_tiles, s.d_gsum, n_groups);
    hipLaunchKernelGGL(pfac_scan_groups_kernel, dim3(1), dim3(1024), 0, s.stream, s.d_gsum, n_groups);
    auto ek = s.last_rec_bytes == 2 ? pfac_expand_kernel<2> : (s.last_rec_bytes == 4 ? pfac_expand_kernel<4> : pfac_expand_kernel<8>);
    hipLaunchKernelGGL(ek, dim3(gblocks), dim3(256), 0, s.stream, src, s.d_tile_index, (unsigned long long)s.last_tiles,
                       s.d_gsum, n_groups, (unsigned long long)s.last_cap, (unsigned long long)first, (unsigned long long)n, d_out);
    HIP_TRY(ctx, hipGetLastError());
    return PFAC_OK;
}

int pfac_scan_format(pfac_ctx *ctx, int slot, int *record_bytes, uint64_t *n_tiles, uint64_t *used) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    Slot &s = ctx->slots[slot];
    if (!s.scanned) return fail(ctx, PFAC_E_STATE, "no scan yet");
    if (record_bytes) *record_bytes = s.last_rec_bytes;
    if (n_tiles) *n_tiles = s.last_tiles;
    if (used) *used = s.last_used;
    return PFAC_OK;
}

int pfac_scan_capacity_hint(pfac_ctx *ctx, int slot, uint64_t *capacity) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    Slot &s = ctx->slots[slot];
    if (!s.scanned || !capacity) return fail(ctx, PFAC_E_STATE, "pfac_scan_capacity_hint: no finished scan");
    // what the finished scan used (+ 1/8: a larger array means larger chunks, i.e. more unfilled space at the end)
    // and a floor from the match count
    const uint64_t a = s.last_used + s.last_used / 8, b = s.last_total + s.last_total / 4;
    *capacity = (a > b ? a : b) + 65536;
    return PFAC_OK;
}

int pfac_records_expand(pfac_ctx *ctx, int slot, const void *d_records, uint64_t first, uint64_t n, pfac_record *d_out) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    Slot &s = ctx->slots[slot];
    const void *src = d_records ? d_records : s.d_records;
    if (!s.scanned) return fail(ctx, PFAC_E_STATE, "pfac_records_expand without a scan");
    if (!src || (!d_out && n) || ((uintptr_t)d_out & 7)) return fail(ctx, PFAC_E_ARG, "pfac_records_expand: bad buffer");
    USE_DEVICE(ctx);
    return expand_records(ctx, s, src, first, n, d_out);
}

int pfac_records_d2h(pfac_ctx *ctx, int slot, const void *d_records, pfac_record *host, uint64_t first, uint64_t n) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    Slot &s = ctx->slots[slot];
    const void *src = d_records ? d_records : s.d_records;
    if (!src || (!host && n)) return fail(ctx, PFAC_E_ARG, "pfac_records_d2h: null buffer");
    if (n == 0) return PFAC_OK;
    if (!s.scanned) return fail(ctx, PFAC_E_STATE, "pfac_records_d2h without a scan");
    USE_DEVICE(ctx);
    if (n > s.wide_cap) {
        if (s.d_wide) { HIP_TRY(ctx, hipStreamSynchronize(s.stream)); HIP_TRY(ctx, hipFree(s.d_wide)); s.d_wide = nullptr; s.wide_cap = 0; }
        HIP_TRY(ctx, hipMalloc((void **)&s.d_wide, n * sizeof(pfac_record)));
        s.wide_cap = n;
    }
    rc = expand_records(ctx, s, src, first, n, s.d_wide);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(host, s.d_wide, n * sizeof(pfac_record), hipMemcpyDeviceToHost, s.stream));
    return PFAC_OK;
}

int pfac_records_d2h_packed(pfac_ctx *ctx, int slot, const void *d_records, void *host_words, uint64_t n_words,
                            uint64_t *host_tile_index) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    Slot &s = ctx->slots[slot];
    const void *src = d_records ? d_records : s.d_records;
    if (!s.scanned || s.last_rec_bytes == 8) return fail(ctx, PFAC_E_STATE, "pfac_records_d2h_packed: the slot's last scan did not produce compact records");
    if (!src || (!host_words && n_words) || !host_tile_index) return fail(ctx, PFAC_E_ARG, "pfac_records_d2h_packed: null buffer");
    if (n_words > s.last_cap) return fail(ctx, PFAC_E_ARG, "pfac_records_d2h_packed: more words than the record array holds");
    USE_DEVICE(ctx);
    if (n_words) HIP_TRY(ctx, hipMemcpyAsync(host_words, src, n_words * (uint64_t)s.last_rec_bytes, hipMemcpyDeviceToHost, s.stream));
    if (s.last_tiles) HIP_TRY(ctx, hipMemcpyAsync(host_tile_index, s.d_tile_index, s.last_tiles * 8, hipMemcpyDeviceToHost, s.stream));
    return PFAC_OK;
}

int pfac_records_packed_device(pfac_ctx *ctx, int slot, const void *d_records, void *d_words_out, uint64_t n_words,
                               uint64_t *d_tile_index_out) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    Slot &s = ctx->slots[slot];
    const void *src = d_records ? d_records : s.d_records;
    if (!s.scanned || s.pending || s.last_rec_bytes == 8) return fail(ctx, PFAC_E_STATE, "pfac_records_packed_device: the slot's last finished scan did not produce compact records");
    if (s.last_used > s.last_cap) return fail(ctx, PFAC_E_OVERFLOW, "the slot's last scan overflowed its record heap: scan again with a larger one");
    if (!d_tile_index_out || (d_words_out && !src)) return fail(ctx, PFAC_E_ARG, "pfac_records_packed_device: null buffer");
    if (n_words > s.last_cap) return fail(ctx, PFAC_E_ARG, "pfac_records_packed_device: more words than the record array holds");
    USE_DEVICE(ctx);
    if (d_words_out && n_words && d_words_out != src)
        HIP_TRY(ctx, hipMemcpyAsync(d_words_out, src, n_words * (uint64_t)s.last_rec_bytes, hipMemcpyDeviceToDevice, s.stream));
    if (s.last_tiles) HIP_TRY(ctx, hipMemcpyAsync(d_tile_index_out, s.d_tile_index, s.last_tiles * 8, hipMemcpyDeviceToDevice, s.stream));
    return PFAC_OK;
}

int pfac_emit_text_device(pfac_ctx *ctx, int slot, const void *d_records, uint64_t base, uint64_t *n_bytes) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (!n_bytes) return fail(ctx, PFAC_E_ARG, "null argument");
    *n_bytes = 0;
    if (!ctx->have_table) return fail(ctx, PFAC_E_STATE, "no table uploaded");
    Slot &s = ctx->slots[slot];
    const void *src = d_records ? d_records : s.d_records;
    if (!s.scanned || s.pending) return fail(ctx, PFAC_E_STATE, "pfac_emit_text_device needs a finished scan");
    if (s.last_used > s.last_cap) return fail(ctx, PFAC_E_OVERFLOW, "the slot's last scan overflowed its record heap: scan again with a larger one");
    if (base + (1ull << 32) >= 1000000000000000000ull) return fail(ctx, PFAC_E_ARG, "pfac_emit_text_device: positions must stay below 10^18");
    s.text_bytes = 0;
    if (s.last_total == 0 || s.last_tiles == 0) return PFAC_OK;
    if (!src) return fail(ctx, PFAC_E_ARG, "null record buffer");
    USE_DEVICE(ctx);
    const unsigned n_groups = (unsigned)((s.last_tiles + XGROUP - 1) / XGROUP);
    rc = ensure_gsum(ctx, s, n_groups);
    if (rc) return rc;
    const unsigned gblocks = (n_groups + 3) / 4;           // four waves (groups of 64 tiles) per 256-thread block
    auto sk = s.last_rec_bytes == 2 ? pfac_text_size_kernel<2> : (s.last_rec_bytes == 4 ? pfac_text_size_kernel<4> : pfac_text_size_kernel<8>);
    hipLaunchKernelGGL(sk, dim3(gblocks), dim3(256), 0, s.stream, src, s.d_tile_index, (unsigned long long)s.last_tiles,
                       (unsigned long long)s.last_cap, (unsigned long long)base, ctx->d_idmap, s.d_gsum, n_groups);
    hipLaunchKernelGGL(pfac_scan_groups_kernel, dim3(1), dim3(1024), 0, s.stream, s.d_gsum, n_groups);
    HIP_TRY(ctx, hipGetLastError());
    unsigned long long total = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&total, s.d_gsum + n_groups, 8, hipMemcpyDeviceToHost, s.stream));
    HIP_TRY(ctx, hipStreamSynchronize(s.stream));
    if (total + 32 > s.text_cap) {
        if (s.d_text) { HIP_TRY(ctx, hipFree(s.d_text)); s.d_text = nullptr; s.text_cap = 0; }
        const uint64_t cap = total + total / 8 + 4096;
        HIP_TRY(ctx, hipMalloc((void **)&s.d_text, cap));
        s.text_cap = cap;
    }
    auto fk = s.last_rec_bytes == 2 ? pfac_text_format_kernel<2> : (s.last_rec_bytes == 4 ? pfac_text_format_kernel<4> : pfac_text_format_kernel<8>);
    hipLaunchKernelGGL(fk, dim3(gblocks), dim3(256), 0, s.stream, src, s.d_tile_index, (unsigned long long)s.last_tiles,
                       (unsigned long long)s.last_cap, (unsigned long long)base, ctx->d_idmap, s.d_gsum, n_groups, s.d_text);
    HIP_TRY(ctx, hipGetLastError());
    s.text_bytes = total;
    *n_bytes = total;
    return PFAC_OK;
}

int pfac_text_d2h(pfac_ctx *ctx, int slot, void *host, uint64_t first, uint64_t n) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    Slot &s = ctx->slots[slot];
    if (first + n > s.text_bytes) return fail(ctx, PFAC_E_ARG, "pfac_text_d2h: range exceeds the text of the slot's last pfac_emit_text_device");
    if (n == 0) return PFAC_OK;
    if (!host) return fail(ctx, PFAC_E_ARG, "null buffer");
    USE_DEVICE(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(host, s.d_text + first, n, hipMemcpyDeviceToHost, s.stream));
    return PFAC_OK;
}

void *pfac_slot_text(pfac_ctx *ctx, int slot) { return check_slot(ctx, slot) ? nullptr : ctx->slots[slot].d_text; }

int pfac_slot_sync(pfac_ctx *ctx, int slot) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    USE_DEVICE(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->slots[slot].stream));
    if (ctx->slots[slot].h2d_issued) HIP_TRY(ctx, hipEventSynchronize(ctx->slots[slot].ev_h2d));   // (a copy nothing on the stream has waited for yet)
    return PFAC_OK;
}

int pfac_records_checksum(pfac_ctx *ctx, int slot, const void *d_records, uint64_t n, uint64_t base, uint64_t *checksum) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (!checksum) return fail(ctx, PFAC_E_ARG, "null argument");
    if (!ctx->have_table) return fail(ctx, PFAC_E_STATE, "no table uploaded");
    Slot &s = ctx->slots[slot];
    const void *src = d_records ? d_records : s.d_records;
    if (!src && n) return fail(ctx, PFAC_E_ARG, "null record buffer");
    if (n && !s.scanned) return fail(ctx, PFAC_E_STATE, "pfac_records_checksum without a scan");
    USE_DEVICE(ctx);
    HIP_TRY(ctx, hipMemsetAsync(s.d_sum, 0, 16, s.stream));
    if (n && s.last_tiles) {
        auto ck = s.last_rec_bytes == 2 ? pfac_checksum_kernel<2> : (s.last_rec_bytes == 4 ? pfac_checksum_kernel<4> : pfac_checksum_kernel<8>);
        hipLaunchKernelGGL(ck, dim3(1024), dim3(256), 0, s.stream, src, s.d_tile_index, (unsigned long long)s.last_tiles,
                           (unsigned long long)s.last_cap, (unsigned long long)base, ctx->d_idmap, s.d_sum);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipMemcpyAsync(s.h_ctl + 8, s.d_sum, 8, hipMemcpyDeviceToHost, s.stream));
    HIP_TRY(ctx, hipStreamSynchronize(s.stream));
    *checksum = ((uint64_t)s.h_ctl[9] << 32) | s.h_ctl[8];
    return PFAC_OK;
}

int pfac_fill_tiled(pfac_ctx *ctx, int slot, void *d_dst, uint64_t n, const void *host_pattern, uint32_t period, uint64_t phase) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (!d_dst || !host_pattern || period == 0 || ((uintptr_t)d_dst & 15)) return fail(ctx, PFAC_E_ARG, "bad argument to pfac_fill_tiled");
    Slot &s = ctx->slots[slot];
    USE_DEVICE(ctx);
    unsigned char *d_pat = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&d_pat, period));
    hipError_t e = hipMemcpy(d_pat, host_pattern, period, hipMemcpyHostToDevice);
    if (e == hipSuccess && n) {
        hipLaunchKernelGGL(pfac_fill_tiled_kernel, dim3(2048), dim3(256), 0, s.stream, static_cast<unsigned char *>(d_dst),
                           (unsigned long long)n, d_pat, period, (unsigned long long)phase);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s.stream);
    (void)hipFree(d_pat);
    if (e != hipSuccess) return fail(ctx, PFAC_E_HIP, std::string("pfac_fill_tiled: ") + hipGetErrorString(e));
    return PFAC_OK;
}

int pfac_fill_random(pfac_ctx *ctx, int slot, void *d_dst, uint64_t n, uint64_t seed) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (!d_dst || ((uintptr_t)d_dst & 7) || (n & 7)) return fail(ctx, PFAC_E_ARG, "pfac_fill_random: dst and n must be multiples of 8");
    Slot &s = ctx->slots[slot];
    USE_DEVICE(ctx);
    if (n) {
        hipLaunchKernelGGL(pfac_fill_random_kernel, dim3(2048), dim3(256), 0, s.stream,
                           static_cast<unsigned long long *>(d_dst), (unsigned long long)(n / 8), (unsigned long long)seed);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipStreamSynchronize(s.stream));
    return PFAC_OK;
}

int pfac_scan_info(pfac_ctx *ctx, int *variant, int *tile_bytes, int *grid_blocks, int *lds_bytes) {
    if (!ctx) return fail(nullptr, PFAC_E_ARG, "null context");
    if (!ctx->have_table) return fail(ctx, PFAC_E_STATE, "no table uploaded");
    if (variant) *variant = ctx->variant;
    if (tile_bytes) *tile_bytes = WTILE;
    if (grid_blocks) *grid_blocks = ctx->grid_blocks;
    if (lds_bytes) *lds_bytes = (ctx->dense && ctx->stage_cap_d) ? ctx->lds_bytes_d : ctx->sparse().lds_bytes;
    return PFAC_OK;
}

int pfac_scan_staging(pfac_ctx *ctx, int *buffers, uint32_t *records_per_buffer) {
    if (!ctx) return fail(nullptr, PFAC_E_ARG, "null context");
    if (!ctx->have_table) return fail(ctx, PFAC_E_STATE, "no table uploaded");
    const bool dense = ctx->dense && ctx->stage_cap_d;
    if (buffers) *buffers = dense ? 1 : ctx->sparse().nbuf;
    // (dense mode's second form has no staging buffer: what bounds a tile there is the wave's record log)
    if (records_per_buffer) *records_per_buffer = dense ? (ctx->dense2 ? ctx->d2log_cap : ctx->stage_cap_d) : ctx->sparse().stage_cap;
    return PFAC_OK;
}

// GPU_Malloc_Memory + GPU_TraceTable + GPU_Free_memory (master_kernel.cu:188-524), one synchronous call.
int pfac_trace_table_compat(const pfac_thread_data *d, int device) {
    if (!d || !d->input_string || !d->match_result || !d->s0Table || !d->r || !d->HT || !d->val || d->input_size < 0 ||
        d->max_pat_len < 1 || d->max_pat_len > HALO_MAX - 1)
        return fail(nullptr, PFAC_E_ARG, "bad pfac_thread_data");
    int wbit = 0;
    if (d->width < 1 || d->width > 4096 || (d->width & (d->width - 1))) return fail(nullptr, PFAC_E_ARG, "width must be a power of two <= 4096");
    while ((d->width >> wbit) != 1) wbit++;
    const int max_row = (int)(((int64_t)d->state_num * 256) / d->width) + 1;
    const int ht = d->HTSize > 0 ? d->HTSize : 1;
    std::vector<int32_t> blob((size_t)PFAC_BLOB_HEADER_WORDS + 256 + max_row + 2 * (size_t)ht + d->final_state_num, 0);
    blob[0] = PFAC_BLOB_MAGIC; blob[1] = PFAC_BLOB_VERSION; blob[2] = d->width; blob[3] = wbit;
    blob[4] = d->final_state_num; blob[5] = d->final_state_num; blob[6] = d->state_num; blob[7] = d->max_pat_len;
    blob[8] = max_row; blob[9] = ht;
    int32_t *p = blob.data() + PFAC_BLOB_HEADER_WORDS;
    memcpy(p, d->s0Table, 256 * 4); p += 256;
    memcpy(p, d->r, (size_t)max_row * 4); p += max_row;
    if (d->HTSize > 0) { memcpy(p, d->HT, (size_t)ht * 4); memcpy(p + ht, d->val, (size_t)ht * 4); }
    else { p[0] = -1; p[ht] = -1; }
    p += 2 * (size_t)ht;
    for (int i = 0; i < d->final_state_num; i++) p[i] = i;
    pfac_ctx *ctx = nullptr;
    int rc = pfac_ctx_create(device, 1, &ctx);
    if (rc) return rc;
    const uint64_t N = (uint64_t)d->input_size;
    uint64_t cap = N / 4 + 4096;
    std::vector<pfac_record> rec;
    uint64_t n = 0;
    rc = pfac_table_upload(ctx, blob.data(), blob.size());
    if (!rc) rc = pfac_slot_reserve(ctx, 0, N, cap);
    if (!rc && N) rc = pfac_slot_h2d(ctx, 0, d->input_string, N, 0);
    for (int attempt = 0; !rc && attempt < 4; attempt++) {
        rc = pfac_scan_async(ctx, 0, nullptr, N, N, nullptr, 0);
        if (!rc) rc = pfac_scan_finish(ctx, 0, &n);
        if (rc == PFAC_E_OVERFLOW && attempt < 3) {
            uint64_t hint = 0;
            rc = pfac_scan_capacity_hint(ctx, 0, &hint);
            cap = hint > 2 * cap ? hint : 2 * cap;
            if (!rc) rc = pfac_slot_reserve(ctx, 0, N, cap);
            continue;
        }
        break;
    }
    if (!rc) {
        rec.resize(n);
        rc = pfac_records_d2h(ctx, 0, nullptr, rec.data(), 0, n);
        if (!rc) rc = pfac_slot_sync(ctx, 0);
    }
    if (!rc) {
        // expand into the reference's dense layout (master_kernel.cu:104-115, memset :236)
        memset(d->match_result, 0xFF, (size_t)N * d->max_pat_len * sizeof(unsigned int));
        uint64_t k = 0;
        while (k < n) {
            uint64_t pos = rec[k].pos, j = 0;
            while (k < n && rec[k].pos == pos) {
                if (j < (uint64_t)d->max_pat_len) d->match_result[pos * d->max_pat_len + j] = rec[k].state;
                j++; k++;
            }
        }
    }
    std::string msg = ctx->err;
    pfac_ctx_destroy(ctx);
    if (rc) g_err = msg;
    return rc;
}

}  // extern "C"

/*
 * pfac_hip.hip -- the MI355X (gfx950 / CDNA4) PFAC scan: kernels + C-ABI runtime.
 *
 * Replaces master_kernel.cu of the reference (TraceTable_kernel :92-180,
 * SUBSEG_MATCH :37-74, GPU_Malloc_Memory :188-257, GPU_TraceTable :277-455,
 * GPU_Free_memory :457-524).  Written for 64-wide wavefronts; not a
 * translation: the reference gives every offset a dense max_pat_len-slot
 * result row (4*max_pat_len bytes of HBM traffic per input byte, three times
 * over); this kernel emits compact, globally ORDERED 8-byte records in a
 * single pass over the input.
 *
 * Kernel structure (one persistent 256-thread workgroup = 4 waves):
 *   ticket   tiles (16 KiB of input) are handed out in order by one atomic,
 *            so a workgroup holding tile t knows tiles < t are running or done
 *   stage    each wave loads its own 4 KiB with 16-B-per-lane buffer loads
 *            (hardware bounds check -> bytes past n_avail read as 0) and
 *            mirrors them + the max_pat_len-1 halo into LDS
 *   root     per byte one LDS lookup in a 256-entry "root has an edge" flag
 *            table -> 16-bit survivor mask per lane          (mk.cu:41)
 *   compact  wave prefix sum packs survivors' positions into an LDS queue, so
 *            the walk runs on dense lanes instead of 1-in-13 active ones
 *   walk     state = PHF(state, byte) from LDS (small tables) or L2 (large)
 *            until dead or out of input; final states are counted (mk.cu:49-71)
 *   order    tile match counts go through a decoupled look-back (single-pass
 *            chained scan over 8-byte {flag,value} words, agent-scope relaxed
 *            atomics) -> every tile learns its first record index
 *   emit     survivors are walked again and write (pos, state) records at
 *            base + wave prefix, i.e. sorted by (position, pattern length),
 *            which is exactly the reference's output order (main.cc:341-349)
 */
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "pfac.h"

namespace {

// ---------------------------------------------------------------------------
// geometry
constexpr int WAVE = 64;
constexpr int BLOCK = 256;
constexpr int NW = BLOCK / WAVE;
constexpr int SUB = WAVE * 16;             // bytes one wave covers with one 16-B-per-lane load
constexpr int SUBS = 4;                    // such loads per wave per tile
constexpr int WCHUNK = SUB * SUBS;         // 4 KiB per wave
constexpr int TILE = NW * WCHUNK;          // 16 KiB per workgroup iteration
constexpr int HALO_MAX = 1024;             // >= max_pat_len - 1 (patterns are < 1024 bytes), multiple of 16

constexpr int OFF_TILE = 0;
constexpr int OFF_QUEUE = OFF_TILE + TILE + HALO_MAX + 16;
constexpr int OFF_S0 = OFF_QUEUE + NW * SUB * 2;
constexpr int OFF_FLAG = OFF_S0 + 256 * 4;
constexpr int OFF_MISC = OFF_FLAG + 256;
constexpr int OFF_TAB = OFF_MISC + 64;
constexpr int LDS_BASE_BYTES = OFF_TAB;
constexpr int LDS_TABLE_MAX = 40 * 1024;   // tables up to this size are staged in LDS (variant 0)

constexpr unsigned long long ST_AGG = 1ull << 62;
constexpr unsigned long long ST_INCL = 2ull << 62;
constexpr unsigned long long ST_VAL = (1ull << 62) - 1;
constexpr unsigned SPIN_MAX = 1u << 20;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ScanArgs {
    const unsigned char *in;
    unsigned long long n_owned, n_avail;
    pfac_record *out;
    unsigned long long out_cap;
    const int *s0;
    const int *r;
    const int2 *T;
    int r_words, t_entries;
    int ht_size, wbit, num_final, halo;
    unsigned n_tiles;
    unsigned *ctl;                 // [0] ticket, [1] error flags, [2..3] total matches (u64)
    unsigned long long *status;    // one look-back word per tile
};

// ---------------------------------------------------------------------------
// wave helpers (wave = 64 lanes)
__device__ __forceinline__ unsigned wave_incl_scan(unsigned x, int lane) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        unsigned t = __shfl_up(x, d, WAVE);
        if (lane >= d) x += t;
    }
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

// ---------------------------------------------------------------------------
// decoupled look-back over per-tile match counts.  Each status word is ONE
// naturally aligned 8-byte granule {flag:2, value:62} moved only by relaxed
// agent-scope atomics (global_load/store ... sc1): the data is the flag, so no
// fence is needed and nothing else is handed between workgroups.
__device__ __forceinline__ unsigned long long st_load(unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by wave 0 only, all 64 lanes.  Returns the number of matches in all
// tiles before `tile` (wave-uniform).  Every spin is bounded: on timeout the
// error word is set and the kernel still terminates.
__device__ unsigned long long lookback(unsigned long long *status, unsigned tile, unsigned long long tot, int lane,
                                       unsigned *err) {
    if (tile == 0) {
        if (lane == 0) st_store(&status[0], ST_INCL | tot);
        return 0;
    }
    if (lane == 0) st_store(&status[tile], ST_AGG | tot);
    unsigned long long excl = 0;
    long long idx = (long long)tile - 1 - lane;       // lane L inspects predecessor tile-1-L
    bool failed = false;
    for (;;) {
        unsigned long long st = ST_INCL;              // tiles before 0: inclusive prefix 0
        bool bad = false;
        if (idx >= 0) {
            unsigned spins = 0;
            for (;;) {
                st = st_load(&status[idx]);
                if (st >> 62) break;
                if (++spins >= SPIN_MAX) { bad = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if (__any(bad)) { failed = true; break; }
        unsigned long long incl_mask = __ballot((st >> 62) == 2);
        unsigned long long v = st & ST_VAL;
        if (incl_mask) {
            int first = __ffsll((long long)incl_mask) - 1;   // nearest predecessor that knows its prefix
            if (lane > first) v = 0;
            excl += wave_sum64(v);
            break;
        }
        excl += wave_sum64(v);
        idx -= WAVE;
    }
    if (failed) {
        if (lane == 0) atomicOr(err, 1u);
        excl = 0;
    }
    if (lane == 0) st_store(&status[tile], ST_INCL | ((excl + tot) & ST_VAL));
    return excl;
}

// ---------------------------------------------------------------------------
// One step of the perfect-hash lookup (master_kernel.cu:52-63): returns the
// next state or -1.  HT/val are interleaved as int2 {owner row, next state}.
__device__ __forceinline__ int phf_step(const int *R, const int2 *T, int state, int ch, int wbit, int ht_size) {
    const int key = (state << 8) | ch;
    const int row = key >> wbit;
    const int col = key & ((1 << wbit) - 1);
    const int idx = R[row] + col;
    if ((unsigned)idx >= (unsigned)ht_size) return -1;
    const int2 e = T[idx];
    return e.x == row ? e.y : -1;
}

// Walk from tile-local position `pos`; counts final states reached, keeps the
// first two in m0/m1.  lim = first tile-local byte that may not be read.
__device__ __forceinline__ unsigned walk(const unsigned char *tile, const int *s0, const int *R, const int2 *T,
                                         unsigned pos, unsigned lim, int wbit, int ht_size, int num_final,
                                         unsigned &m0, unsigned &m1) {
    unsigned n = 0;
    int s = s0[tile[pos]];
    unsigned p = pos + 1;
    while (s >= 0) {
        if (s < num_final) {
            m0 = n == 0 ? (unsigned)s : m0;     // selects, not an indexed pair (that would go to scratch)
            m1 = n == 1 ? (unsigned)s : m1;
            n++;
        }
        if (p >= lim) break;
        s = phf_step(R, T, s, tile[p], wbit, ht_size);
        p++;
    }
    return n;
}

// Same walk, storing every final state from the `skip`-th on (rare path: more
// than two patterns start at one offset).
__device__ __forceinline__ void walk_store(const unsigned char *tile, const int *s0, const int *R, const int2 *T,
                                           unsigned pos, unsigned lim, int wbit, int ht_size, int num_final,
                                           unsigned skip, pfac_record *out, unsigned long long ri,
                                           unsigned long long cap, unsigned gpos) {
    unsigned n = 0;
    int s = s0[tile[pos]];
    unsigned p = pos + 1;
    while (s >= 0) {
        if (s < num_final) {
            if (n >= skip && ri + n < cap) {
                pfac_record rec;
                rec.pos = gpos;
                rec.state = (unsigned)s;
                out[ri + n] = rec;
            }
            n++;
        }
        if (p >= lim) break;
        s = phf_step(R, T, s, tile[p], wbit, ht_size);
        p++;
    }
}

// One pass of one wave over its 4 KiB chunk.  WRITE == false: returns the
// wave's match count.  WRITE == true: emits records starting at index wrun.
template <bool WRITE>
__device__ __forceinline__ unsigned long long wave_pass(const ScanArgs &a, const unsigned char *tile, const int *s0,
                                                        const int *R, const int2 *T, unsigned short *q,
                                                        const unsigned (&masks)[SUBS], int wave, int lane,
                                                        unsigned lim, unsigned long long tile_base,
                                                        unsigned long long wrun) {
    unsigned lane_total = 0;
#pragma unroll
    for (int j = 0; j < SUBS; j++) {
        const unsigned mask = masks[j];
        const unsigned cnt = __popc(mask);
        const unsigned incl = wave_incl_scan(cnt, lane);
        const unsigned S = bcast_last(incl);
        if (S == 0) continue;
        const unsigned lpos = wave * WCHUNK + j * SUB + lane * 16;
        unsigned o = incl - cnt;
        for (unsigned m = mask; m; m &= m - 1) q[o++] = (unsigned short)(lpos + (__ffs(m) - 1));
        wave_lds_sync();
        for (unsigned base = 0; base < S; base += WAVE) {
            const unsigned qi = base + lane;
            unsigned n = 0, m0 = 0, m1 = 0, pos = 0;
            if (qi < S) {
                pos = q[qi];
                n = walk(tile, s0, R, T, pos, lim, a.wbit, a.ht_size, a.num_final, m0, m1);
            }
            if (!WRITE) {
                lane_total += n;
            } else {
                const unsigned inc = wave_incl_scan(n, lane);
                const unsigned long long ri = wrun + (inc - n);
                const unsigned gpos = (unsigned)(tile_base + pos);
                if (n > 0 && ri < a.out_cap) {
                    pfac_record rec;
                    rec.pos = gpos;
                    rec.state = m0;
                    a.out[ri] = rec;
                }
                if (n > 1 && ri + 1 < a.out_cap) {
                    pfac_record rec;
                    rec.pos = gpos;
                    rec.state = m1;
                    a.out[ri + 1] = rec;
                }
                if (n > 2)
                    walk_store(tile, s0, R, T, pos, lim, a.wbit, a.ht_size, a.num_final, 2, a.out, ri, a.out_cap, gpos);
                wrun += bcast_last(inc);
            }
        }
        wave_lds_sync();   // queue is reused by the next sub-tile
    }
    if (!WRITE) return wave_sum64(lane_total);
    return wrun;
}

template <bool TLDS>
__global__ __launch_bounds__(BLOCK) void pfac_scan_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *tile = smem + OFF_TILE;
    unsigned short *queue = reinterpret_cast<unsigned short *>(smem + OFF_QUEUE);
    int *s0 = reinterpret_cast<int *>(smem + OFF_S0);
    unsigned char *flag = smem + OFF_FLAG;
    unsigned *misc = reinterpret_cast<unsigned *>(smem + OFF_MISC);   // [0..3] wave totals, [4] tile, [6..7] base
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wave = tid >> 6;

    // ---- once per workgroup: root row, root flags, (small) PHF tables -> LDS
    {
        const int v = a.s0[tid];
        s0[tid] = v;
        flag[tid] = v >= 0 ? 1 : 0;
    }
    const int *R = a.r;
    const int2 *T = a.T;
    if (TLDS) {
        int *lr = reinterpret_cast<int *>(smem + OFF_TAB);
        int2 *lt = reinterpret_cast<int2 *>(smem + OFF_TAB + ((a.r_words * 4 + 15) & ~15));
        for (int i = tid; i < a.r_words; i += BLOCK) lr[i] = a.r[i];
        for (int i = tid; i < a.t_entries; i += BLOCK) lt[i] = a.T[i];
        R = lr;
        T = lt;
    }
    unsigned short *q = queue + wave * SUB;

    for (;;) {
        __syncthreads();                       // previous tile fully consumed (LDS + misc)
        if (tid == 0) misc[4] = atomicAdd(&a.ctl[0], 1u);
        __syncthreads();
        const unsigned t = misc[4];
        if (t >= a.n_tiles) break;
        const unsigned long long tile_base = (unsigned long long)t * TILE;
        const unsigned long long remain = a.n_avail - tile_base;           // > 0
        const unsigned lim = remain < (unsigned long long)(TILE + a.halo) ? (unsigned)remain : (unsigned)(TILE + a.halo);

        // ---- stage: own 4 KiB per wave (+ halo by wave 0) -> LDS.  The buffer descriptor covers
        // whole 16-B units only, so every dword of a load is either fully inside or reads as 0.
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned char *>(a.in + tile_base), 0, (int)(lim & ~15u), 0x00020000);
        {
            u32x4 w[SUBS];
#pragma unroll
            for (int j = 0; j < SUBS; j++)
                w[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, wave * WCHUNK + j * SUB + lane * 16, 0, 0);
            u32x4 hw = {0u, 0u, 0u, 0u};
            const bool has_halo = wave == 0 && lane * 16 < a.halo;
            if (has_halo) hw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, TILE + lane * 16, 0, 0);
#pragma unroll
            for (int j = 0; j < SUBS; j++)
                *reinterpret_cast<u32x4 *>(tile + wave * WCHUNK + j * SUB + lane * 16) = w[j];
            if (has_halo) *reinterpret_cast<u32x4 *>(tile + TILE + lane * 16) = hw;
        }
        __syncthreads();                       // tile + halo visible to every wave
        if (lim & 15u) {                       // ragged end of the input (last tile only): patch the tail bytes
            if (tid < (int)(lim & 15u)) tile[(lim & ~15u) + tid] = a.in[tile_base + (lim & ~15u) + tid];
            __syncthreads();
        }

        // ---- root test: one flag lookup per byte -> 16-bit survivor mask per lane per sub-tile
        unsigned masks[SUBS];
#pragma unroll
        for (int j = 0; j < SUBS; j++) {
            const unsigned off = wave * WCHUNK + j * SUB + lane * 16;
            const u32x4 w = *reinterpret_cast<const u32x4 *>(tile + off);
            unsigned m = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const unsigned b = (w[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                m |= (unsigned)flag[b] << k;
            }
            // only offsets below n_owned start a walk
            const unsigned long long g = tile_base + off;
            if (g + 16 > a.n_owned) m = g >= a.n_owned ? 0u : (m & ((1u << (unsigned)(a.n_owned - g)) - 1u));
            masks[j] = m;
        }

        // ---- pass 1: count
        const unsigned long long wtot = wave_pass<false>(a, tile, s0, R, T, q, masks, wave, lane, lim, tile_base, 0);
        if (lane == 0) misc[wave] = (unsigned)wtot;
        __syncthreads();
        if (wave == 0) {
            const unsigned long long tot = (unsigned long long)misc[0] + misc[1] + misc[2] + misc[3];
            const unsigned long long excl = lookback(a.status, t, tot, lane, &a.ctl[1]);
            if (lane == 0) {
                misc[6] = (unsigned)excl;
                misc[7] = (unsigned)(excl >> 32);
                if (t == a.n_tiles - 1) {
                    a.ctl[2] = (unsigned)(excl + tot);
                    a.ctl[3] = (unsigned)((excl + tot) >> 32);
                }
            }
        }
        __syncthreads();
        // ---- pass 2: emit (skipped by waves without matches)
        if (misc[wave] != 0) {
            unsigned long long wrun = ((unsigned long long)misc[7] << 32) | misc[6];
            for (int k = 0; k < wave; k++) wrun += misc[k];
            wave_pass<true>(a, tile, s0, R, T, q, masks, wave, lane, lim, tile_base, wrun);
        }
    }
}

// ---------------------------------------------------------------------------
// small service kernels

// blob (int32 image, pfac.h) -> device layout {s0[256] | r[max_row] | pad | T[ht_size] int2 | idmap}
__global__ void pfac_repack_kernel(const int *blob, int *s0, int *r, int2 *T, int *idmap, int max_row, int ht_size,
                                   int num_final) {
    const int *b_s0 = blob + PFAC_BLOB_HEADER_WORDS;
    const int *b_r = b_s0 + 256;
    const int *b_HT = b_r + max_row;
    const int *b_val = b_HT + ht_size;
    const int *b_id = b_val + ht_size;
    const int stride = gridDim.x * blockDim.x;
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = i0; i < 256; i += stride) s0[i] = b_s0[i];
    for (int i = i0; i < max_row; i += stride) r[i] = b_r[i];
    for (int i = i0; i < ht_size; i += stride) T[i] = make_int2(b_HT[i], b_val[i]);
    for (int i = i0; i < num_final; i += stride) idmap[i] = b_id[i];
}

__device__ __forceinline__ unsigned long long match_hash(unsigned long long pos, unsigned id) {
    unsigned long long x = (pos + 1) * 0x9E3779B97F4A7C15ull ^ ((unsigned long long)id * 0xC2B2AE3D27D4EB4Full);
    x ^= x >> 29;
    return x * 0xBF58476D1CE4E5B9ull;
}

__global__ void pfac_checksum_kernel(const pfac_record *rec, unsigned long long n, unsigned long long base,
                                     const int *idmap, unsigned long long *out) {
    unsigned long long sum = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const pfac_record r = rec[i];
        sum += match_hash(base + r.pos, (unsigned)idmap[r.state]);
    }
    sum = wave_sum64(sum);
    if ((threadIdx.x & (WAVE - 1)) == 0) atomicAdd(out, sum);
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
    pfac_record *d_records = nullptr;
    uint64_t record_cap = 0;
    unsigned *d_ctl = nullptr;            // 16 control words followed by the status array
    uint64_t status_cap = 0;              // tiles
    unsigned *h_ctl = nullptr;            // pinned: [0] ticket [1] err [2..3] total, [4..5] checksum
    unsigned long long *d_sum = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint64_t last_cap = 0;
    bool scanned = false;
};

}  // namespace

struct pfac_ctx {
    int device = 0;
    int n_cu = 0;
    std::vector<Slot> slots;
    // table
    int *d_tab = nullptr;
    size_t tab_bytes = 0;
    int *d_s0 = nullptr, *d_r = nullptr, *d_idmap = nullptr;
    int2 *d_T = nullptr;
    int width_bit = 0, num_final = 0, max_pat_len = 0, max_row = 0, ht_size = 0, state_num = 0;
    bool have_table = false;
    int variant = 1;
    int lds_bytes = 0;
    int grid_blocks = 0;
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

int check_slot(pfac_ctx *ctx, int slot) {
    if (!ctx) return fail(nullptr, PFAC_E_ARG, "null context");
    if (slot < 0 || slot >= (int)ctx->slots.size()) return fail(ctx, PFAC_E_ARG, "bad slot index");
    return PFAC_OK;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int ensure_status(pfac_ctx *ctx, Slot &s, uint64_t n_tiles) {
    if (s.d_ctl && s.status_cap >= n_tiles) return PFAC_OK;
    if (s.d_ctl) HIP_TRY(ctx, hipFree(s.d_ctl));
    s.d_ctl = nullptr;
    uint64_t cap = n_tiles < 4096 ? 4096 : n_tiles;
    HIP_TRY(ctx, hipMalloc((void **)&s.d_ctl, 64 + cap * 8 + 16));
    s.status_cap = cap;
    return PFAC_OK;
}

int configure_kernel(pfac_ctx *ctx) {
    // table bytes if staged in LDS: r (16-B rounded) + T
    const size_t tbytes = align_up((size_t)ctx->max_row * 4, 16) + (size_t)ctx->ht_size * 8;
    ctx->variant = tbytes <= (size_t)LDS_TABLE_MAX ? 0 : 1;
    ctx->lds_bytes = LDS_BASE_BYTES + (ctx->variant == 0 ? (int)align_up(tbytes, 16) : 0);
    int per_cu = 0;
    if (ctx->variant == 0) {
        HIP_TRY(ctx, hipFuncSetAttribute((const void *)pfac_scan_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_bytes));
        HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pfac_scan_kernel<true>, BLOCK, ctx->lds_bytes));
    } else {
        HIP_TRY(ctx, hipFuncSetAttribute((const void *)pfac_scan_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_bytes));
        HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pfac_scan_kernel<false>, BLOCK, ctx->lds_bytes));
    }
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    ctx->grid_blocks = ctx->n_cu * per_cu;
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
    hipLaunchKernelGGL(pfac_repack_kernel, dim3(256), dim3(256), 0, stream, d_blob, ctx->d_s0, ctx->d_r, ctx->d_T,
                       ctx->d_idmap, max_row, ht_size, num_final);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    ctx->width_bit = wbit; ctx->num_final = num_final; ctx->max_pat_len = max_pat_len;
    ctx->max_row = max_row; ctx->ht_size = ht_size; ctx->state_num = state_num;
    ctx->have_table = true;
    return configure_kernel(ctx);
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
    HIP_TRY(ctx, hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(ctx, hipGetDeviceProperties(&prop, device));
    ctx->n_cu = prop.multiProcessorCount;
    ctx->slots.resize(n_streams);
    for (auto &s : ctx->slots) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&s.own_stream, hipStreamNonBlocking));
        s.stream = s.own_stream;
        HIP_TRY(ctx, hipHostMalloc((void **)&s.h_ctl, 64, hipHostMallocDefault));
        memset(s.h_ctl, 0, 64);
        HIP_TRY(ctx, hipMalloc((void **)&s.d_sum, 16));
        HIP_TRY(ctx, hipEventCreate(&s.ev0));
        HIP_TRY(ctx, hipEventCreate(&s.ev1));
    }
    *out = ctx;
    return PFAC_OK;
}

void pfac_ctx_destroy(pfac_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (auto &s : ctx->slots) {
        if (s.own_stream) (void)hipStreamSynchronize(s.own_stream);
        if (s.d_input) (void)hipFree(s.d_input);
        if (s.d_records) (void)hipFree(s.d_records);
        if (s.d_ctl) (void)hipFree(s.d_ctl);
        if (s.d_sum) (void)hipFree(s.d_sum);
        if (s.h_ctl) (void)hipHostFree(s.h_ctl);
        if (s.ev0) (void)hipEventDestroy(s.ev0);
        if (s.ev1) (void)hipEventDestroy(s.ev1);
        if (s.own_stream) (void)hipStreamDestroy(s.own_stream);
    }
    if (ctx->d_tab) (void)hipFree(ctx->d_tab);
    delete ctx;
}

int pfac_table_upload(pfac_ctx *ctx, const int32_t *blob, size_t n_words) {
    if (!ctx || !blob || n_words < PFAC_BLOB_HEADER_WORDS) return fail(ctx, PFAC_E_ARG, "bad argument to pfac_table_upload");
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
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

int pfac_slot_reserve(pfac_ctx *ctx, int slot, uint64_t input_bytes, uint64_t record_capacity) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Slot &s = ctx->slots[slot];
    if (input_bytes > s.input_cap) {
        if (s.d_input) { HIP_TRY(ctx, hipFree(s.d_input)); s.d_input = nullptr; s.input_cap = 0; }
        const uint64_t cap = align_up(input_bytes, TILE) + HALO_MAX + 256;
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
pfac_record *pfac_slot_records(pfac_ctx *ctx, int slot) { return check_slot(ctx, slot) ? nullptr : ctx->slots[slot].d_records; }
void *pfac_slot_stream(pfac_ctx *ctx, int slot) { return check_slot(ctx, slot) ? nullptr : (void *)ctx->slots[slot].stream; }

int pfac_slot_set_stream(pfac_ctx *ctx, int slot, void *stream_handle) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    Slot &s = ctx->slots[slot];
    s.stream = stream_handle ? reinterpret_cast<hipStream_t>(stream_handle) : s.own_stream;
    return PFAC_OK;
}

int pfac_slot_h2d(pfac_ctx *ctx, int slot, const void *host, uint64_t n_bytes, uint64_t dst_offset) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    Slot &s = ctx->slots[slot];
    if (!host || dst_offset + n_bytes > s.input_cap) return fail(ctx, PFAC_E_ARG, "pfac_slot_h2d: range exceeds the reserved input buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(s.d_input + dst_offset, host, n_bytes, hipMemcpyHostToDevice, s.stream));
    return PFAC_OK;
}

int pfac_scan_async(pfac_ctx *ctx, int slot, const void *d_input, uint64_t n_owned, uint64_t n_avail,
                    pfac_record *d_records, uint64_t capacity) {
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
    if (!d_records && capacity) return fail(ctx, PFAC_E_ARG, "pfac_scan_async: no record buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t n_tiles = (n_owned + TILE - 1) / TILE;
    s.last_cap = capacity;
    s.scanned = true;
    rc = ensure_status(ctx, s, n_tiles);
    if (rc) return rc;
    // control words + the status words this launch polls, zeroed every call
    HIP_TRY(ctx, hipMemsetAsync(s.d_ctl, 0, 64 + align_up(n_tiles * 8, 16), s.stream));
    HIP_TRY(ctx, hipEventRecord(s.ev0, s.stream));
    if (n_tiles > 0) {
        ScanArgs a;
        a.in = in; a.n_owned = n_owned; a.n_avail = n_avail;
        a.out = d_records; a.out_cap = capacity;
        a.s0 = ctx->d_s0; a.r = ctx->d_r; a.T = ctx->d_T;
        a.r_words = ctx->max_row; a.t_entries = ctx->ht_size;
        a.ht_size = ctx->ht_size; a.wbit = ctx->width_bit; a.num_final = ctx->num_final;
        int halo = ctx->max_pat_len > 1 ? ctx->max_pat_len - 1 : 0;
        a.halo = (halo + 15) & ~15;
        a.n_tiles = (unsigned)n_tiles;
        a.ctl = s.d_ctl;
        a.status = reinterpret_cast<unsigned long long *>(s.d_ctl + 16);
        uint64_t grid = (uint64_t)ctx->grid_blocks < n_tiles ? (uint64_t)ctx->grid_blocks : n_tiles;
        if (ctx->variant == 0)
            hipLaunchKernelGGL(pfac_scan_kernel<true>, dim3((unsigned)grid), dim3(BLOCK), ctx->lds_bytes, s.stream, a);
        else
            hipLaunchKernelGGL(pfac_scan_kernel<false>, dim3((unsigned)grid), dim3(BLOCK), ctx->lds_bytes, s.stream, a);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipEventRecord(s.ev1, s.stream));
    HIP_TRY(ctx, hipMemcpyAsync(s.h_ctl, s.d_ctl, 16, hipMemcpyDeviceToHost, s.stream));
    return PFAC_OK;
}

int pfac_scan_finish(pfac_ctx *ctx, int slot, uint64_t *n_matches) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    Slot &s = ctx->slots[slot];
    if (!s.scanned) return fail(ctx, PFAC_E_STATE, "pfac_scan_finish without a scan");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(s.stream));
    const uint64_t total = ((uint64_t)s.h_ctl[3] << 32) | s.h_ctl[2];
    if (n_matches) *n_matches = total;
    if (s.h_ctl[1] != 0) return fail(ctx, PFAC_E_INTERNAL, "scan kernel reported a look-back timeout");
    if (total > s.last_cap) return fail(ctx, PFAC_E_OVERFLOW, "more matches than record capacity");
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

int pfac_records_d2h(pfac_ctx *ctx, int slot, const pfac_record *d_records, pfac_record *host, uint64_t first, uint64_t n) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    Slot &s = ctx->slots[slot];
    const pfac_record *src = d_records ? d_records : s.d_records;
    if (!src || (!host && n)) return fail(ctx, PFAC_E_ARG, "pfac_records_d2h: null buffer");
    if (n == 0) return PFAC_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(host, src + first, n * sizeof(pfac_record), hipMemcpyDeviceToHost, s.stream));
    return PFAC_OK;
}

int pfac_slot_sync(pfac_ctx *ctx, int slot) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->slots[slot].stream));
    return PFAC_OK;
}

int pfac_records_checksum(pfac_ctx *ctx, int slot, const pfac_record *d_records, uint64_t n, uint64_t base, uint64_t *checksum) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (!checksum) return fail(ctx, PFAC_E_ARG, "null argument");
    if (!ctx->have_table) return fail(ctx, PFAC_E_STATE, "no table uploaded");
    Slot &s = ctx->slots[slot];
    const pfac_record *src = d_records ? d_records : s.d_records;
    if (!src && n) return fail(ctx, PFAC_E_ARG, "null record buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemsetAsync(s.d_sum, 0, 16, s.stream));
    if (n) {
        hipLaunchKernelGGL(pfac_checksum_kernel, dim3(1024), dim3(256), 0, s.stream, src, (unsigned long long)n,
                           (unsigned long long)base, ctx->d_idmap, s.d_sum);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipMemcpyAsync(s.h_ctl + 4, s.d_sum, 8, hipMemcpyDeviceToHost, s.stream));
    HIP_TRY(ctx, hipStreamSynchronize(s.stream));
    *checksum = ((uint64_t)s.h_ctl[5] << 32) | s.h_ctl[4];
    return PFAC_OK;
}

int pfac_fill_tiled(pfac_ctx *ctx, int slot, void *d_dst, uint64_t n, const void *host_pattern, uint32_t period, uint64_t phase) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (!d_dst || !host_pattern || period == 0 || ((uintptr_t)d_dst & 15)) return fail(ctx, PFAC_E_ARG, "bad argument to pfac_fill_tiled");
    Slot &s = ctx->slots[slot];
    HIP_TRY(ctx, hipSetDevice(ctx->device));
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
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
    if (tile_bytes) *tile_bytes = TILE;
    if (grid_blocks) *grid_blocks = ctx->grid_blocks;
    if (lds_bytes) *lds_bytes = ctx->lds_bytes;
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
    for (int attempt = 0; !rc && attempt < 2; attempt++) {
        rc = pfac_scan_async(ctx, 0, nullptr, N, N, nullptr, 0);
        if (!rc) rc = pfac_scan_finish(ctx, 0, &n);
        if (rc == PFAC_E_OVERFLOW && attempt == 0) { cap = n; rc = pfac_slot_reserve(ctx, 0, N, cap); continue; }
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

/*
 * include/pfac.h -- C-ABI of the MI355X-native PFAC matcher.
 *
 * Two shared libraries implement it:
 *   libpfac_host.so  (plain C, no GPU)  : pattern file -> PHF-compressed state-transition table
 *   libpfac_hip.so   (HIP, gfx950)      : device contexts, table upload, the scan kernel, records
 *
 * Every entry point below names the reference interface it replaces
 * (paths relative to mickeyjoe666/PHFPFAC regex_GPU_PHF/).  The reference has
 * no FFI: main.cc calls three C++-linkage functions (main.cc:35-37) and
 * #includes its table builder (main.cc:5-6).  A maintainer switches to this
 * library by replacing those call sites; see INTEGRATION.md.
 *
 * Conventions: plain pointers and integers only; every function returns 0 on
 * success or a negative pfac_status; nothing calls exit() (the reference exits
 * on every error, e.g. master_kernel.cu:240-244).  No function falls back to
 * a CPU implementation: without a usable GPU the HIP library reports
 * PFAC_E_NO_DEVICE / PFAC_E_HIP.
 */
#ifndef PFAC_H
#define PFAC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PFAC_OK = 0,
    PFAC_E_ARG = -1,        /* bad argument (NULL, misaligned pointer, width not a power of two <= 4096, ...) */
    PFAC_E_IO = -2,         /* cannot open / read a file */
    PFAC_E_PATTERN = -3,    /* pattern file violates the reader's rules (length >= 1024, empty line, no trailing '\n') */
    PFAC_E_NOMEM = -4,
    PFAC_E_NO_DEVICE = -5,  /* no HIP device / bad device index */
    PFAC_E_HIP = -6,        /* a HIP runtime call failed; see pfac_last_error() */
    PFAC_E_STATE = -7,      /* call order violated (scan before table upload, ...) */
    PFAC_E_OVERFLOW = -8,   /* more matches than the record buffer holds (count is still exact) */
    PFAC_E_INTERNAL = -9    /* kernel reported an internal fault (a bounded wait of the scan protocol timed out) */
} pfac_status;

/* ------------------------------------------------------------------ */
/* Host side (libpfac_host.so): the CreateTable/ + PHF/ path.          */

/*
 * The PHF-compressed transition table in the reference's own terms
 * (struct thread_data, main.cc:19-32 / master_kernel.cu:15-28):
 *   s0[256]          root row  PFAC[initial_state][*]            (main.cc:200)
 *   r[max_row]       row displacement, may be negative, -1 empty (phf.c:67,197)
 *   HT[ht_size]      owning row of each slot, -1 free            (phf.c:211)
 *   val[ht_size]     next state of each slot                     (phf.c:216)
 *   idmap[num_final] final state -> pattern id (1-based line no) (create_table_reorder.c:318)
 * lookup(state, ch): key=(state<<8)+ch; row=key>>width_bit; col=key&(width-1);
 *   idx=r[row]+col; 0<=idx<ht_size && HT[idx]==row ? val[idx] : -1   (master_kernel.cu:52-63)
 * States: finals are 0..num_final-1 (index in the sorted pattern list), the
 * root is num_final+1, internal states follow (create_table_reorder.c:287-292).
 * Unlike the reference, ONE automaton covers the whole pattern file: the
 * 4*streamnum pattern chunks (create_table_reorder.c:217) are not needed
 * because the input, not the pattern set, is what gets sharded.
 */
typedef struct pfac_table {
    int32_t width;          /* power of two, 1 <= width <= 4096 (phf.c:8,161) */
    int32_t width_bit;      /* log2(width) (master_kernel.cu:397-398) */
    int32_t n_patterns;     /* lines in the pattern file */
    int32_t num_final;      /* == n_patterns (duplicates keep their own, unreachable, final state) */
    int32_t state_num;
    int32_t max_pat_len;
    int32_t max_row;        /* entries in r   = (state_num*256)/width + 1 (master_kernel.cu:212) */
    int32_t ht_size;        /* entries in HT and val */
    int32_t n_keys;         /* transitions stored */
    int32_t *s0;
    int32_t *r;
    int32_t *HT;
    int32_t *val;
    int32_t *idmap;
} pfac_table;

/* Replaces create_PFAC_table_reorder() + FFDM() (main.cc:108,125):
 * read_pattern (create_table_reorder.c:53), sort (comp_pat :21), trie
 * (patternsToPFAC :277) and the row-displacement perfect hash (phf.c:151),
 * in near-linear time and without the 4 GiB-per-chunk preallocation. */
int pfac_table_build_file(const char *pattern_file, int width, pfac_table **out, char *err, size_t err_len);
/* Escape-aware variant: the reference's read_pattern_ext()/fgetc_ext() (create_table_reorder.c:131-185,
 * ctdef.h:37-99, dead code there): \a \b \t \n \v \f \r \' \" \\ \ooo \xNN inside patterns; only a real
 * newline separates patterns, so patterns may contain newline bytes.  Everything downstream is unchanged. */
int pfac_table_build_file_escaped(const char *pattern_file, int width, pfac_table **out, char *err, size_t err_len);
/* Same as pfac_table_build_file, from a memory image of a pattern file. */
int pfac_table_build_mem(const void *patterns, size_t n_bytes, int width, pfac_table **out, char *err, size_t err_len);
/* Pattern-partition mode -- the reference's own multi-GPU scheme (create_table_reorder.c:217-247), kept as a
 * fallback for automata that outgrow L2/MALL: the table of partition `part` of `n_parts` of the SORTED pattern
 * list (k = n / P patterns each, the last one also takes n % P; ids stay whole-file line numbers, max_pat_len
 * stays the global maximum).  A cut never separates identical lines, so duplicates resolve inside one
 * partition ("last line wins") instead of overflowing result slots as main.cc:308-315 does.  Every partition
 * scans the whole input; pfac_merge_partitions() below is the merge of main.cc:304-324. */
int pfac_table_build_file_part(const char *pattern_file, int width, int part, int n_parts, pfac_table **out, char *err,
                               size_t err_len);
int pfac_table_build_mem_part(const void *patterns, size_t n_bytes, int width, int part, int n_parts, pfac_table **out,
                              char *err, size_t err_len);
void pfac_table_free(pfac_table *t);

/* Character-class patterns -- the front end the reference sketches in CreateTable/charset_table_reorder.c:45-168,
 * 321-427 (orphaned and not compilable there): a pattern is a sequence of single (escape-aware) characters and classes
 * "[...]" / "[^...]" with "l-r" ranges; no repetition.  The subset construction yields an acyclic DFA that is used as
 * the PFAC table unchanged (finals first, root = num_final + 1).  A final state can stand for several patterns:
 * outputs->ids[outputs->first[s] .. outputs->first[s+1]) lists them, ascending; table->idmap[s] is the first one.
 * num_final counts FINAL STATES here, n_patterns the lines of the file.  Parity with the reference: unpinned (its code
 * for this cannot be built); pinned against an independent brute-force matcher (oracle/charclass_oracle.py). */
typedef struct pfac_outputs {
    int32_t n_states;       /* == table->num_final */
    int32_t *first;         /* [n_states + 1] */
    int32_t *ids;           /* pattern ids (1-based line numbers) */
} pfac_outputs;
int pfac_table_build_file_charclass(const char *pattern_file, int width, pfac_table **out, pfac_outputs **outputs, char *err,
                                    size_t err_len);
int pfac_table_build_mem_charclass(const void *patterns, size_t n_bytes, int width, pfac_table **out, pfac_outputs **outputs,
                                   char *err, size_t err_len);
void pfac_outputs_free(pfac_outputs *o);

/* The device lookup evaluated on the host (property tests; never used on the scan path). */
int32_t pfac_table_lookup(const pfac_table *t, int32_t state, int32_t ch);

/* Flat int32 image of a table: what gets uploaded, and what RCCL broadcasts
 * between ranks.  Layout: 16-word header {magic, version, width, width_bit,
 * n_patterns, num_final, state_num, max_pat_len, max_row, ht_size, n_keys,
 * 0...} then s0[256], r[max_row], HT[ht_size], val[ht_size], idmap[num_final]. */
#define PFAC_BLOB_MAGIC 0x50464143 /* "PFAC" */
#define PFAC_BLOB_VERSION 1
#define PFAC_BLOB_HEADER_WORDS 16
size_t pfac_table_blob_words(const pfac_table *t);
int pfac_table_to_blob(const pfac_table *t, int32_t *blob, size_t n_words);
int pfac_table_from_blob(const int32_t *blob, size_t n_words, pfac_table **out);
/* Wrap arrays produced by the reference's own FFDM() (main.cc:72-76,125) so a
 * reference build can feed this library without rebuilding its tables. */
int pfac_table_from_reference_arrays(const int32_t *s0, const int32_t *r, const int32_t *HT, const int32_t *val,
                                     const int32_t *idmap, int32_t width, int32_t state_num, int32_t num_final,
                                     int32_t ht_size, int32_t max_pat_len, pfac_table **out);

/* Text emitter: replaces main.cc:335-350.  One line per record,
 * "At position %4d, match pattern %d\n" with pos = base + rec.pos and
 * pattern = idmap[rec.state].  Appends to an open FILE* (void* to keep stdio
 * out of the ABI); returns bytes written (>= 0) or a negative status. */
typedef struct pfac_record {
    uint32_t pos;       /* start offset, relative to the first byte of the scanned range */
    uint32_t state;     /* final state reached (== index into idmap) */
} pfac_record;
/*
 * On the DEVICE the scan writes the COMPACT form of the same list: one word per match,  (pos & 4095) | state << 12,
 * as wide as the automaton needs -- 16 bits for at most 16 final states, 32 bits up to 2^20, else an 8-byte
 * pfac_record (pfac_scan_format tells which) -- into a record HEAP, plus an ordered TILE INDEX: the records of 4 KiB
 * input tile t are the PFAC_TIX_COUNT(tile_index[t]) words that start at word PFAC_TIX_FIRST(tile_index[t]), in
 * (position, pattern length) order, with  pos = t * 4096 + (word & 4095).  Walking the index in tile order yields the
 * reference's output order; the heap itself has small gaps and no global order (a workgroup fills chunks of it with
 * the tiles it scans -- a globally contiguous array would cost a chip-wide prefix over batches that are still being
 * scanned).  pfac_records_d2h / pfac_records_expand deliver ONE sorted pfac_record array whatever the form.
 */
#define PFAC_TILE_BYTES 4096
#define PFAC_PACKED_POS(word) ((uint32_t)(word) & 4095u)
#define PFAC_PACKED_STATE(word) ((uint32_t)(word) >> 12)
#define PFAC_TIX_FIRST(e) ((uint64_t)(e) & ((1ull << 40) - 1))
#define PFAC_TIX_COUNT(e) ((uint32_t)((uint64_t)(e) >> 40))
/* idmap == NULL: rec.state already holds the pattern id (the output of pfac_merge_partitions). */
int64_t pfac_emit_records(void *file, const pfac_record *rec, uint64_t n, uint64_t base, const int32_t *idmap);
/* Character-class tables: one line per (record, pattern ending in the record's final state), ascending pattern id. */
int64_t pfac_emit_records_multi(void *file, const pfac_record *rec, uint64_t n, uint64_t base, const pfac_outputs *outputs);
/* Same bytes, produced by n_threads host threads (size pass, prefix sum, format + pwrite in place); the serial
 * fprintf loop is the end-to-end wall once the scan runs at TB/s.  The file must be seekable; falls back to the
 * serial emitter for small n, n_threads < 2 or pipes. */
int64_t pfac_emit_records_mt(void *file, const pfac_record *rec, uint64_t n, uint64_t base, const int32_t *idmap,
                             int n_threads);
/* The same text straight from the compact device form (record heap + tile index as pfac_records_d2h_packed
 * delivers them), tiles in order: position = base + t * 4096 + PFAC_PACKED_POS(word), pattern =
 * idmap[PFAC_PACKED_STATE(word)]; record_bytes = 2 or 4 and n_words = heap words held by `words` (*used), as
 * pfac_scan_format reports.  A tile whose records would lie outside words[0, n_words) -> PFAC_E_ARG, nothing written.
 * n_threads < 2: serial. */
int64_t pfac_emit_packed(void *file, const void *words, uint64_t n_words, int record_bytes, const uint64_t *tile_index,
                         uint64_t n_tiles, uint64_t base, const int32_t *idmap, int n_threads);
/* Merge of per-partition match lists, replaces main.cc:304-324.  lists[k] (counts[k] records, sorted by
 * position as the scan emits them) comes from partition k of pfac_table_build_file_part(); the result is
 * ordered by (position, partition) -- i.e. by (position, pattern length), the reference's output order -- and
 * its `state` field holds the PATTERN ID: idmaps[k][state] (idmaps == NULL or idmaps[k] == NULL: the list
 * already holds ids).  Returns the number of records written, PFAC_E_OVERFLOW if out_cap is too small. */
int64_t pfac_merge_partitions(const pfac_record *const *lists, const uint64_t *counts, const int32_t *const *idmaps,
                              int n_parts, pfac_record *out, uint64_t out_cap);

/* ------------------------------------------------------------------ */
/* Device side (libpfac_hip.so): the master_kernel.cu path.            */

typedef struct pfac_ctx pfac_ctx;   /* one per GPU; holds what d_input_string/d_r/d_hash_table/
                                       d_match_result/d_val_table/d_s0Table held (main.cc:99-104) */

int pfac_device_count(int *n);                                       /* cudaGetDeviceCount, main.cc:50 */
/* Replaces cudaSetDevice + cudaStreamCreate (main.cc:183,209) and the
 * allocation half of GPU_Malloc_Memory (master_kernel.cu:188-257).
 * n_streams >= 1 independent pipeline slots ("streams per GPU", argv[2]). */
int pfac_ctx_create(int device, int n_streams, pfac_ctx **out);
void pfac_ctx_destroy(pfac_ctx *ctx);                                /* GPU_Free_memory, master_kernel.cu:457-524 */
const char *pfac_last_error(const pfac_ctx *ctx);                    /* ctx may be NULL: last error of the calling thread */

/* Table upload: the H2D copies of r/HT/s0/val and the texture binds
 * (master_kernel.cu:302-320,365-383).  blob is a pfac_table_to_blob image in
 * host memory; the _device variant takes an image already in this GPU's
 * memory (e.g. the receive buffer of an RCCL broadcast) and uses `stream_handle`
 * (a hipStream_t, may be NULL) for ordering. */
int pfac_table_upload(pfac_ctx *ctx, const int32_t *blob, size_t n_words);
int pfac_table_upload_device(pfac_ctx *ctx, const void *d_blob, size_t n_words, void *stream_handle);

/* Pinned host memory, replaces cudaHostAlloc(..., cudaHostAllocPortable) (main.cc:147,161). */
int pfac_host_alloc(void **p, size_t n_bytes);
void pfac_host_free(void *p);
/* Make an existing, page-aligned host range DMA-able in place -- e.g. a MAP_SHARED mapping of the input file: the H2D
 * copy then reads the page cache itself, and no CPU thread copies the input at all (gphf's default ingest).  Ranges
 * must not overlap; unregister before unmapping.  Fails (PFAC_E_HIP) where the driver cannot pin the pages. */
int pfac_host_register(void *p, size_t n_bytes);
int pfac_host_unregister(void *p);

/* Per-slot device buffers owned by the context.  Input capacity is rounded up
 * so the kernel's tile loads stay in bounds (master_kernel.cu:217 pads by
 * one tile + 512 B for the same reason). */
int pfac_slot_reserve(pfac_ctx *ctx, int slot, uint64_t input_bytes, uint64_t record_capacity);
void *pfac_slot_input(pfac_ctx *ctx, int slot);          /* device pointer, 256-B aligned */
void *pfac_slot_records(pfac_ctx *ctx, int slot);        /* device pointer (record_capacity x 8 bytes, either record form) */
void *pfac_slot_stream(pfac_ctx *ctx, int slot);         /* the slot's hipStream_t */
/* Use an EXTERNAL stream (e.g. torch's current stream) for a slot; NULL restores the slot's own. */
int pfac_slot_set_stream(pfac_ctx *ctx, int slot, void *stream_handle);

/* Async H2D of input bytes into the slot's input buffer at dst_offset
 * (cudaMemcpy H2D, master_kernel.cu:359, made asynchronous on the slot's stream). */
int pfac_slot_h2d(pfac_ctx *ctx, int slot, const void *host, uint64_t n_bytes, uint64_t dst_offset);
/* Block until the slot's last pfac_slot_h2d has left the host buffer (which may then be refilled while the scan that
 * follows it on the stream is still running): what lets a reader pool run ahead of the copies. */
int pfac_slot_h2d_wait(pfac_ctx *ctx, int slot);
/* The same question without blocking: 1 = it has, 0 = not yet, negative = error. */
int pfac_slot_h2d_done(pfac_ctx *ctx, int slot);

/*
 * The scan: replaces the kernel launch of GPU_TraceTable (master_kernel.cu:396-423).
 *   d_input   device pointer, 16-B aligned; NULL = the slot's own input buffer
 *   n_owned   start offsets [0, n_owned) are matched and reported   (<= 2^32)
 *   n_avail   bytes readable from d_input, n_owned <= n_avail; walks that
 *             start in the owned range may read up to n_avail (the halo of
 *             max_pat_len-1 bytes that belongs to the next shard) and never beyond
 *   d_records device pointer for the record heap, 16-B aligned, NULL = the slot's;
 *             at most capacity x 8 bytes are written (capacity x 2 or x 4 in the compact forms)
 *   capacity  records the heap holds.  It needs some slack over the match count (chunks a workgroup has not
 *             filled: at most capacity/16, plus gaps below 1 %); the match count is exact even when the
 *             heap overflows, and pfac_scan_capacity_hint() then says what to reserve
 * Through the tile index (which lives in the slot) the records are ordered by (pos, pattern length) == the
 * reference's output order (main.cc:341-349).  Asynchronous on the slot's stream.
 */
int pfac_scan_async(pfac_ctx *ctx, int slot, const void *d_input, uint64_t n_owned, uint64_t n_avail,
                    void *d_records, uint64_t capacity);
/* Wait for the slot and fetch the exact number of matches.  Returns
 * PFAC_E_OVERFLOW (with *n_matches set) when capacity was exceeded. */
int pfac_scan_finish(pfac_ctx *ctx, int slot, uint64_t *n_matches);
/* After pfac_scan_finish: a record capacity that the same scan fits (the finished scan's heap use + margin). */
int pfac_scan_capacity_hint(pfac_ctx *ctx, int slot, uint64_t *capacity);
/* Kernel time of the slot's last scan (hipEvent pair around the launch, the
 * analogue of "2. MASTER: The elapsed time is %f ms", master_kernel.cu:400-421). */
int pfac_scan_elapsed_ms(pfac_ctx *ctx, int slot, float *ms);
/* D2H of records [first, first+n) of the sorted sequence as pfac_record (the compact replacement of the dense
 * cudaMemcpy D2H, master_kernel.cu:428).  Asynchronous; pfac_slot_sync() completes it. */
int pfac_records_d2h(pfac_ctx *ctx, int slot, const void *d_records, pfac_record *host, uint64_t first, uint64_t n);
int pfac_slot_sync(pfac_ctx *ctx, int slot);
/* Record form of the slot's last finished scan: *record_bytes = 2 or 4 (compact words) or 8 (pfac_record in the
 * heap); *n_tiles = tiles scanned = entries of the tile index; *used = heap records in use (<= capacity unless it
 * overflowed). */
int pfac_scan_format(pfac_ctx *ctx, int slot, int *record_bytes, uint64_t *n_tiles, uint64_t *used);
/* Records [first, first+n) of the slot's last scan, SORTED, as pfac_record in DEVICE memory (d_out, 8-B aligned), on
 * the slot's stream -- for consumers that stay on the GPU (the RCCL record gather). */
int pfac_records_expand(pfac_ctx *ctx, int slot, const void *d_records, uint64_t first, uint64_t n, pfac_record *d_out);
/* D2H of the compact form itself: heap words [0, n_words) (n_words = *used of pfac_scan_format; record_bytes each)
 * and the n_tiles entries of the tile index (2 or 4 bytes per match over PCIe instead of 8; pfac_emit_packed() prints
 * from it).  PFAC_E_STATE when the last scan was not compact. */
int pfac_records_d2h_packed(pfac_ctx *ctx, int slot, const void *d_records, void *host_words, uint64_t n_words,
                            uint64_t *host_tile_index);

/* GPU-side text emitter -- the fprintf loop of main.cc:335-350 run on the device: the records of the slot's last finished
 * scan -> the lines "At position %4d, match pattern %d\n" (position = base + pos, pattern = idmap[state] of the uploaded
 * table), in the reference's output order, into a device buffer the slot owns; *n_bytes = their total size.  Line length
 * depends on the digit counts: one kernel sizes the lines per 64 tiles, a prefix sum places them, a third formats -- the
 * host only copies finished text (pfac_text_d2h, asynchronous on the slot's stream; pfac_slot_sync completes it) and
 * write()s it.  base + 2^32 must stay below 10^18.  Byte-identical to pfac_emit_records / pfac_emit_packed.
 * (Character-class tables, whose final states may stand for several patterns, print on the host: pfac_emit_records_multi.) */
int pfac_emit_text_device(pfac_ctx *ctx, int slot, const void *d_records, uint64_t base, uint64_t *n_bytes);
int pfac_text_d2h(pfac_ctx *ctx, int slot, void *host, uint64_t first, uint64_t n_bytes);
void *pfac_slot_text(pfac_ctx *ctx, int slot);            /* device pointer of that text (valid until the slot's next pfac_emit_text_device) */

/* The compact form handed to a consumer that STAYS ON THE DEVICE (the RCCL record gather sends it: 2 or 4 bytes per
 * match plus 8 bytes per 4 KiB tile over xGMI instead of 8-byte pfac_records): the n_tiles entries of the slot's tile
 * index -> d_tile_index_out and, unless d_words_out is NULL or the record heap itself, heap words [0, n_words) ->
 * d_words_out (device pointers; n_words = *used of pfac_scan_format).  Asynchronous on the slot's stream. */
int pfac_records_packed_device(pfac_ctx *ctx, int slot, const void *d_records, void *d_words_out, uint64_t n_words,
                               uint64_t *d_tile_index_out);

/* Order-independent 64-bit checksum of ALL records of the slot's last scan (sum over records of
 * mix(base+pos, idmap[state])), computed on the GPU; used for full-size parity checks where materialising
 * the text is not practical.  n = the scan's match count (0: checksum of nothing). */
int pfac_records_checksum(pfac_ctx *ctx, int slot, const void *d_records, uint64_t n, uint64_t base,
                          uint64_t *checksum);

/* Synthetic input generators, written straight into device memory (the
 * reference built big inputs by tiling a small text, creatbiginput.sh:2-5).
 *   tiled : byte i = pattern[(phase + i) % period]
 *   random: byte i = byte (i&7) of splitmix64(seed + (i>>3))  (counter based) */
int pfac_fill_tiled(pfac_ctx *ctx, int slot, void *d_dst, uint64_t n, const void *host_pattern, uint32_t period,
                    uint64_t phase);
int pfac_fill_random(pfac_ctx *ctx, int slot, void *d_dst, uint64_t n, uint64_t seed);

/* Kernel introspection for bench/DESIGN: variant chosen for the uploaded table
 * (0 = tables in LDS, 1 = tables via L2), tile bytes, grid size, LDS bytes. */
int pfac_scan_info(pfac_ctx *ctx, int *variant, int *tile_bytes, int *grid_blocks, int *lds_bytes);
/* ... and the staging layout the NEXT scan will use: buffers per wave (3 / 2: a tile's records leave two / one
 * round(s) after it was scanned; 1: dense mode, emitted at once) and records per buffer (a tile with more is
 * walked a second time; 4096 = dense mode's second form, where it is the wave's record log in device memory that
 * bounds a tile).  The layout follows the match density of the scans before. */
int pfac_scan_staging(pfac_ctx *ctx, int *buffers, uint32_t *records_per_buffer);

/* ------------------------------------------------------------------ */
/* Drop-in shaped like the reference seam (main.cc:19-37).             */

/* Field-for-field struct thread_data (main.cc:19-32). */
typedef struct pfac_thread_data {
    unsigned char *input_string;
    int input_size;
    int state_num;
    int final_state_num;
    unsigned int *match_result;     /* dense: input_size * max_pat_len slots, 0xFFFFFFFF = empty */
    int HTSize;
    int width;
    int *s0Table;
    int max_pat_len;
    int *r;
    int *HT;
    int *val;
} pfac_thread_data;

/* GPU_Malloc_Memory + GPU_TraceTable + GPU_Free_memory in one synchronous
 * call (master_kernel.cu:188-524): uploads dataset's tables and input to
 * `device`, scans, and fills dataset->match_result in the reference's dense
 * layout (slot j of position i = j-th final state reached from i, rest
 * 0xFFFFFFFF) so that main.cc:304-350 can run unchanged on it. */
int pfac_trace_table_compat(const pfac_thread_data *dataset, int device);

#ifdef __cplusplus
}
#endif
#endif /* PFAC_H */

/*
 * gphf.c -- the command-line driver, same surface as the reference's main()
 * (regex_GPU_PHF/main.cc:45-352):
 *
 *     gphf <pattern file name> <streamnum> <PHF width> <input file name>
 *
 * writes GPU_match_result.txt in the current directory, one line per match,
 * "At position %4d, match pattern %d\n" (main.cc:335-350), byte-identical to
 * the reference for every input in its parity domain.
 *
 * What differs underneath (all through the C-ABI of include/pfac.h):
 *   - ONE automaton for the whole pattern file; the INPUT is cut into chunks
 *     (owned bytes + max_pat_len-1 bytes of halo) that are dealt round-robin
 *     to the workers (one per visible GPU), instead of the pattern set being
 *     partitioned (create_table_reorder.c:217-247);
 *   - streaming, ZERO-COPY ingest: the input file is mapped, a registrar thread
 *     makes the mapping DMA-able piece by piece (256 MiB ahead of the copies,
 *     unpinned behind them) and the H2D copies read the page cache itself -- no
 *     CPU thread ever touches the input bytes (the reference fread()s the whole
 *     file into one cudaHostAlloc buffer, main.cc:147-155), inputs larger than
 *     host RAM work, and the pipeline runs at the host link's rate.  Where the
 *     driver cannot pin page-cache pages (or with PFAC_INGEST=pread) a POOL of
 *     reader threads pread()s the chunks, piece by piece, into pinned staging
 *     buffers ahead of the copies instead;
 *   - <streamnum> really is the number of pipeline slots per GPU: chunk k+1 is
 *     copied H2D while chunk k is scanned and chunk k-1's results return (the
 *     reference creates streams, main.cc:209, and never uses them);
 *   - the device contexts are created while the host still builds the table;
 *   - results: the GPU FORMATS the text itself (pfac_emit_text_device: the
 *     fprintf loop of main.cc:341-349 as three kernels); finished text comes
 *     back through a ring of pinned buffers and a pool of writer threads
 *     pwrite()s it at offsets the chunk order fixes.  PFAC_EMIT=host selects
 *     the host formatter instead (compact records -- 2 or 4 bytes per match
 *     plus 8 bytes per 4 KiB tile -- D2H, printed in chunk order by
 *     pfac_emit_packed on several threads).
 * There is no CPU matching path in this program: without a GPU it fails.
 *
 * Environment: PFAC_GPUS=n limits the number of GPUs used; PFAC_WORKERS_PER_GPU=m runs m independent workers (host
 * thread + context + pipeline slots each) on every GPU -- the chunks are dealt round-robin over all n * m workers, so the
 * multi-worker dealing and the shared in-order output can be exercised on a single device; PFAC_CHUNK_MB sets the chunk
 * size (default 32); PFAC_INGEST=pread makes a pool of PFAC_READ_THREADS threads (default: cores, at most 16) copy the file into
 * pinned staging buffers instead of mapping it (also the fallback where the driver cannot pin page-cache pages);
 * PFAC_EMIT=host|device; PFAC_EMIT_THREADS the host formatter's / the writer pool's threads; PFAC_TIMELINE=1 prints milestones.
 */
#define _FILE_OFFSET_BITS 64
#define _GNU_SOURCE
#include "pfac.h"

#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef struct {
    uint64_t base;          /* global offset of the chunk's first owned byte */
    uint64_t n_owned, n_avail;
    /* host formatter: host copy of the chunk's records: 8-byte form (automata beyond 2^20 final states) ... */
    pfac_record *rec;
    void *words;            /* ... or the compact form: record heap (16- or 32-bit words) + tile index (pfac.h) */
    int word_bytes;
    uint64_t *tix;
    uint64_t n_tiles;
    uint64_t n_words;       /* heap words copied back (pfac_scan_format's *used) */
    uint64_t n_rec;
    /* device formatter: bytes of text this chunk prints, and where they go in the file */
    uint64_t text_bytes, text_off;
    int sized;              /* text_bytes is known          (guarded by g_mu) */
    int done;               /* results handed over          (guarded by g_mu) */
} chunk_t;

enum { ST_UNALLOC = 0, ST_FREE, ST_READING, ST_READY, ST_INFLIGHT };
typedef struct {
    void *buf;              /* pinned: chunk_bytes + the longest halo */
    int state;              /* guarded by g_mu */
    int pieces_left;
    int io_error;
} stage_t;

typedef struct worker {
    int index, device, n_workers, n_streams;
    int fd;                         /* input file */
    uint64_t chunk_bytes;
    chunk_t *chunks;                /* all chunks; this worker takes k = index, index + n_workers, ... */
    int n_chunks, n_mine;
    stage_t *stage;
    int n_stage;
    int next_read;                  /* ordinal (among this worker's chunks) of the next chunk to hand to the readers */
    uint64_t piece;
    int emit_device;
    int out_fd;
    int ingest_mmap;                /* chunks are copied H2D straight out of the page cache (registered in place); else pread() into staging */
    unsigned char *halo_buf;        /* ingest_mmap: n_streams x 1 KiB pinned: a chunk's halo (the first bytes of the NEXT chunk's range,
                                       which that chunk registers itself) goes through here */

    double kernel_ms, setup_ms, table_wait_ms, read_wait_ms, drain_ms, text_ms;   /* where this worker's wall time went */
    double issue_h2d_ms, issue_scan_ms;
    uint64_t matches;
    int internal_retries;           /* scans repeated after PFAC_E_INTERNAL (a protocol timeout: a bug, reported, never hidden) */
    int rc;
    char err[256];
} worker_t;

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_cv = PTHREAD_COND_INITIALIZER;
static int g_emitted = 0;           /* chunks whose results have left the pipeline, in order */
static int g_window = 0;            /* a worker may run at most this many chunks ahead of that */
static int g_failed = 0;
/* the table, built by main() while the workers create their contexts */
static const int32_t *g_blob = NULL;
static size_t g_blob_words = 0;
static int g_table_ready = 0;
/* device formatter: file offsets are handed out in chunk order as the chunks' text sizes come in */
static int g_next_off_chunk = 0;
static uint64_t g_text_total = 0;

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
}

/* PFAC_TIMELINE=1: milestones on stderr, milliseconds since program start */
static double g_t_start = 0;
static int g_timeline = 0;
static unsigned char *g_map = NULL;     /* zero-copy ingest: the input file, mapped MAP_SHARED (the page cache itself) */
static uint64_t g_map_len = 0;
/* ... registered for DMA piece by piece (a piece = a whole number of chunks, ~256 MiB) by ONE registrar thread that runs a
 * bounded distance ahead of the copies; a piece is unpinned when the last copy out of it has completed */
static uint64_t g_piece_bytes = 0;
static int g_n_pieces = 0;
static int g_reg_upto = 0;          /* pieces [0, g_reg_upto) have been registered            (guarded by g_mu) */
static int g_reg_low = 0;           /* pieces [0, g_reg_low) have been consumed and unpinned   (guarded by g_mu) */
static int *g_piece_left = NULL;    /* copies still to come out of each piece                  (guarded by g_mu) */
static int g_zero_copy = -1;        /* -1 not tried yet, 0 refused by the driver, 1 in use */
enum { REG_LOOKAHEAD = 4 };         /* pieces pinned ahead of the oldest unfinished one (bounds the pinned page cache) */
static int g_skip_read = 0, g_skip_gpu = 0;     /* diagnostics (PFAC_GPHF_SKIP=read|gpu): time one half of the pipeline alone; results are wrong */
#define MILESTONE(...) do { if (g_timeline) { fprintf(stderr, "[%9.2f ms] ", now_ms() - g_t_start); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); } } while (0)

static void set_failed(void) {
    pthread_mutex_lock(&g_mu);
    g_failed = 1;
    pthread_cond_broadcast(&g_cv);
    pthread_mutex_unlock(&g_mu);
}

static int fail(worker_t *w, pfac_ctx *ctx, int rc, const char *what) {
    snprintf(w->err, sizeof w->err, "worker %d (GPU %d): %s: %s", w->index, w->device, what, ctx ? pfac_last_error(ctx) : "");
    w->rc = rc;
    set_failed();
    return rc;
}

static int read_fully(int fd, void *dst, uint64_t n, uint64_t off) {
    unsigned char *p = (unsigned char *)dst;
    while (n) {
        ssize_t r = pread(fd, p, n > (1u << 30) ? (1u << 30) : (size_t)n, (off_t)off);
        if (r <= 0) return -1;
        p += r; off += (uint64_t)r; n -= (uint64_t)r;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------
 * Thread pool (readers and writers share the code): a job is a function and four words; FIFO. */
typedef struct { void (*fn)(void *a, uint64_t x, uint64_t y, uint64_t z); void *a; uint64_t x, y, z; } job_t;
typedef struct {
    pthread_mutex_t mu;
    pthread_cond_t cv, idle_cv;
    job_t *q;
    size_t cap, head, count;
    int stop, n_threads, busy;
    pthread_t *th;
} pool_t;

static void *pool_main(void *arg) {
    pool_t *p = (pool_t *)arg;
    pthread_mutex_lock(&p->mu);
    for (;;) {
        while (!p->count && !p->stop) pthread_cond_wait(&p->cv, &p->mu);
        if (!p->count && p->stop) break;
        job_t j = p->q[p->head];
        p->head = (p->head + 1) % p->cap;
        p->count--;
        p->busy++;
        pthread_mutex_unlock(&p->mu);
        j.fn(j.a, j.x, j.y, j.z);
        pthread_mutex_lock(&p->mu);
        p->busy--;
        if (!p->count && !p->busy) pthread_cond_broadcast(&p->idle_cv);
    }
    pthread_mutex_unlock(&p->mu);
    return NULL;
}

static int pool_start(pool_t *p, int n_threads, size_t cap) {
    memset(p, 0, sizeof *p);
    pthread_mutex_init(&p->mu, NULL);
    pthread_cond_init(&p->cv, NULL);
    pthread_cond_init(&p->idle_cv, NULL);
    p->q = (job_t *)malloc(cap * sizeof(job_t));
    p->th = (pthread_t *)malloc((size_t)n_threads * sizeof(pthread_t));
    if (!p->q || !p->th) return -1;
    p->cap = cap;
    for (int i = 0; i < n_threads; i++)
        if (pthread_create(&p->th[p->n_threads], NULL, pool_main, p) == 0) p->n_threads++;
    return p->n_threads > 0 ? 0 : -1;
}

static void pool_push(pool_t *p, job_t j) {
    pthread_mutex_lock(&p->mu);
    if (p->count == p->cap) {                       /* grow (rare: the queue is sized for the pieces in flight) */
        job_t *nq = (job_t *)malloc(2 * p->cap * sizeof(job_t));
        for (size_t i = 0; i < p->count; i++) nq[i] = p->q[(p->head + i) % p->cap];
        free(p->q);
        p->q = nq; p->head = 0; p->cap *= 2;
    }
    p->q[(p->head + p->count) % p->cap] = j;
    p->count++;
    pthread_cond_signal(&p->cv);
    pthread_mutex_unlock(&p->mu);
}

static void pool_wait_idle(pool_t *p) {
    pthread_mutex_lock(&p->mu);
    while (p->count || p->busy) pthread_cond_wait(&p->idle_cv, &p->mu);
    pthread_mutex_unlock(&p->mu);
}

static void pool_stop(pool_t *p) {
    pthread_mutex_lock(&p->mu);
    p->stop = 1;
    pthread_cond_broadcast(&p->cv);
    pthread_mutex_unlock(&p->mu);
    for (int i = 0; i < p->n_threads; i++) pthread_join(p->th[i], NULL);
    free(p->q); free(p->th);
}

static pool_t g_readers, g_writers;

/* ---- reader pool: one job = one piece of one chunk -> its place in a pinned staging buffer.  A single thread copies
 * out of the page cache at ~5 GB/s, far below what the H2D link takes: the pieces of the chunks in flight are read by
 * all the pool's threads, ahead of the worker that will copy them to the device. */
static void read_piece(void *a, uint64_t stage_idx, uint64_t chunk_ord, uint64_t off) {
    worker_t *w = (worker_t *)a;
    stage_t *st = &w->stage[stage_idx];
    const chunk_t *c = &w->chunks[w->index + (int)chunk_ord * w->n_workers];
    const uint64_t n = c->n_avail - off < w->piece ? c->n_avail - off : w->piece;
    int rc = 0;
    if (g_skip_read) {
    } else {
        rc = read_fully(w->fd, (unsigned char *)st->buf + off, n, c->base + off);
    }
    pthread_mutex_lock(&g_mu);
    if (rc) st->io_error = 1;
    if (--st->pieces_left == 0) {
        st->state = ST_READY;
        pthread_cond_broadcast(&g_cv);
    }
    pthread_mutex_unlock(&g_mu);
}

/* hand the next chunks of this worker to the readers while staging buffers are free (g_mu held; needs the chunk plan,
 * i.e. the table's halo) */
static void schedule_reads(worker_t *w) {
    if (!g_table_ready || g_failed) return;
    while (w->next_read < w->n_mine) {
        const int b = w->next_read % w->n_stage;
        stage_t *st = &w->stage[b];
        if (st->state != ST_FREE) break;
        const chunk_t *c = &w->chunks[w->index + w->next_read * w->n_workers];
        const int pieces = (int)((c->n_avail + w->piece - 1) / w->piece);
        st->io_error = 0;
        st->state = ST_READING;
        st->pieces_left = pieces;
        for (int p = 0; p < pieces; p++) {
            const job_t j = {read_piece, w, (uint64_t)b, (uint64_t)w->next_read, (uint64_t)p * w->piece};
            pool_push(&g_readers, j);
        }
        w->next_read++;
    }
}

/* ---- writer pool (device formatter): finished text arrives in pinned buffers; the pieces of a buffer are pwrite()n by
 * several threads (one thread copies into the page cache at a few GB/s), the last one returns the buffer. */
enum { TEXT_BUFS_MAX = 6 };
#define TEXT_BUF_BYTES ((uint64_t)64 << 20)
#define WRITE_PIECE ((uint64_t)4 << 20)
typedef struct { void *buf; int pieces_left; int in_use; } textbuf_t;
static textbuf_t g_tbuf[TEXT_BUFS_MAX];
static int g_tbuf_n = 0;            /* allocated so far (lazily: a run without matches never pins any) */
static int g_write_error = 0;

static void write_piece(void *a, uint64_t fd_and_idx, uint64_t file_off, uint64_t n) {
    textbuf_t *tb = &g_tbuf[fd_and_idx & 0xFF];
    const int fd = (int)(fd_and_idx >> 8);
    const unsigned char *p = (const unsigned char *)a;
    int bad = 0;
    while (n) {
        ssize_t r = pwrite(fd, p, (size_t)n, (off_t)file_off);
        if (r <= 0) { bad = 1; break; }
        p += r; file_off += (uint64_t)r; n -= (uint64_t)r;
    }
    pthread_mutex_lock(&g_mu);
    if (bad) { g_write_error = 1; g_failed = 1; }
    if (--tb->pieces_left == 0) tb->in_use = 0;
    pthread_cond_broadcast(&g_cv);
    pthread_mutex_unlock(&g_mu);
}

/* a free pinned text buffer (allocating one while fewer than TEXT_BUFS_MAX exist); -1 on failure */
static int text_buffer_acquire(void) {
    pthread_mutex_lock(&g_mu);
    for (;;) {
        for (int i = 0; i < g_tbuf_n; i++)
            if (!g_tbuf[i].in_use && g_tbuf[i].buf) { g_tbuf[i].in_use = 1; pthread_mutex_unlock(&g_mu); return i; }
        if (g_tbuf_n < TEXT_BUFS_MAX) {
            const int i = g_tbuf_n++;
            g_tbuf[i].in_use = 1;
            pthread_mutex_unlock(&g_mu);
            void *p = NULL;
            if (pfac_host_alloc(&p, TEXT_BUF_BYTES)) return -1;
            pthread_mutex_lock(&g_mu);
            g_tbuf[i].buf = p;
            pthread_mutex_unlock(&g_mu);
            return i;
        }
        if (g_failed) { pthread_mutex_unlock(&g_mu); return -1; }
        pthread_cond_wait(&g_cv, &g_mu);
    }
}

/* ---- zero-copy ingest: the registrar.  Populating the page tables of a piece (large folios: microseconds) and registering
 * it (~0.1 ms per 32 MiB) is all the CPU ever does with the input; the H2D copies then read the page cache itself. */
static void *registrar(void *arg) {
    (void)arg;
    for (int i = 0; i < g_n_pieces; i++) {
        pthread_mutex_lock(&g_mu);
        while (i >= g_reg_low + REG_LOOKAHEAD && !g_failed) pthread_cond_wait(&g_cv, &g_mu);
        const int stop = g_failed;
        pthread_mutex_unlock(&g_mu);
        if (stop) break;
        unsigned char *p = g_map + (uint64_t)i * g_piece_bytes;
        uint64_t len = g_map_len - (uint64_t)i * g_piece_bytes;
        if (len > g_piece_bytes) len = g_piece_bytes;
#ifdef MADV_POPULATE_READ
        (void)madvise(p, (size_t)len, MADV_POPULATE_READ);
#endif
        if (pfac_host_register(p, (size_t)len)) {
            fprintf(stderr, "gphf: registering the input mapping failed: %s\n", pfac_last_error(NULL));
            set_failed();
            break;
        }
        pthread_mutex_lock(&g_mu);
        g_reg_upto = i + 1;
        pthread_cond_broadcast(&g_cv);
        pthread_mutex_unlock(&g_mu);
    }
    return NULL;
}

/* the copy of the chunk at `base` has completed: unpin its piece when that was the last one out of it (g_mu NOT held) */
static void piece_copy_done(uint64_t base) {
    const int pi = (int)(base / g_piece_bytes);
    pthread_mutex_lock(&g_mu);
    const int last = --g_piece_left[pi] == 0;
    pthread_mutex_unlock(&g_mu);
    if (!last) return;
    (void)pfac_host_unregister(g_map + (uint64_t)pi * g_piece_bytes);
    pthread_mutex_lock(&g_mu);
    g_piece_left[pi] = -1;
    while (g_reg_low < g_n_pieces && g_piece_left[g_reg_low] < 0) g_reg_low++;
    pthread_cond_broadcast(&g_cv);
    pthread_mutex_unlock(&g_mu);
}

static void chunk_done(chunk_t *c) {
    pthread_mutex_lock(&g_mu);
    c->done = 1;
    pthread_cond_broadcast(&g_cv);
    pthread_mutex_unlock(&g_mu);
}

/* finish the chunk that occupies `slot`: wait, fetch count, (re-scan on overflow), bring the results back, publish */
static int drain(worker_t *w, pfac_ctx *ctx, int slot, int k, uint64_t *cap) {
    chunk_t *c = &w->chunks[k];
    uint64_t n = 0;
    int rc = pfac_scan_finish(ctx, slot, &n);
    /* a record heap that was too small is grown and the chunk (still in the slot's input buffer) scanned again.  A
     * protocol timeout (PFAC_E_INTERNAL) gets ONE retry so that a long job is not lost -- but every bounded wait of the
     * kernel is inside one workgroup, so a timeout is a defect, not contention: it is printed when it happens and counted
     * in the summary, never swallowed */
    for (int attempt = 0, retried = 0; (rc == PFAC_E_OVERFLOW && attempt < 4) || (rc == PFAC_E_INTERNAL && !retried); attempt++) {
        if (rc == PFAC_E_INTERNAL) {
            retried = 1;
            w->internal_retries++;
            fprintf(stderr, "gphf: GPU %d: chunk at offset %llu: %s -- scanning it once more\n", w->device,
                    (unsigned long long)c->base, pfac_last_error(ctx));
        } else {
            uint64_t hint = 0;
            if ((rc = pfac_scan_capacity_hint(ctx, slot, &hint))) return fail(w, ctx, rc, "capacity hint");
            *cap = hint > 2 * *cap ? hint : 2 * *cap;
            if ((rc = pfac_slot_reserve(ctx, slot, 0, *cap))) return fail(w, ctx, rc, "reserve");
        }
        if ((rc = pfac_scan_async(ctx, slot, NULL, c->n_owned, c->n_avail, NULL, 0))) return fail(w, ctx, rc, "scan");
        rc = pfac_scan_finish(ctx, slot, &n);
    }
    if (rc) return fail(w, ctx, rc, "scan");
    float ms = 0;
    if (pfac_scan_elapsed_ms(ctx, slot, &ms) == 0) w->kernel_ms += ms;
    c->n_rec = n;
    w->matches += n;
    if (w->emit_device) {
        /* the GPU prints: size pass + prefix sum + format (three kernels), then the text comes back in pieces */
        const double t0 = now_ms();
        uint64_t tb = 0;
        if (n && (rc = pfac_emit_text_device(ctx, slot, NULL, c->base, &tb))) return fail(w, ctx, rc, "device text emitter");
        if (k < 2) MILESTONE("worker %d: chunk %d: %llu bytes of text formatted on the device", w->index, k, (unsigned long long)tb);
        /* file offsets go out in chunk order: this chunk's is known once every earlier chunk has reported its size */
        pthread_mutex_lock(&g_mu);
        c->text_bytes = tb;
        c->sized = 1;
        while (g_next_off_chunk < w->n_chunks && w->chunks[g_next_off_chunk].sized) {
            w->chunks[g_next_off_chunk].text_off = g_text_total;
            g_text_total += w->chunks[g_next_off_chunk].text_bytes;
            g_next_off_chunk++;
        }
        pthread_cond_broadcast(&g_cv);
        while (k >= g_next_off_chunk && !g_failed) pthread_cond_wait(&g_cv, &g_mu);
        const int stop = g_failed;
        pthread_mutex_unlock(&g_mu);
        if (stop) return -1;
        for (uint64_t off = 0; off < tb; off += TEXT_BUF_BYTES) {
            const uint64_t len = tb - off < TEXT_BUF_BYTES ? tb - off : TEXT_BUF_BYTES;
            const int bi = text_buffer_acquire();
            if (bi < 0) return fail(w, NULL, PFAC_E_NOMEM, "pinned text buffer");
            if ((rc = pfac_text_d2h(ctx, slot, g_tbuf[bi].buf, off, len)) || (rc = pfac_slot_sync(ctx, slot))) return fail(w, ctx, rc, "text d2h");
            const int pieces = (int)((len + WRITE_PIECE - 1) / WRITE_PIECE);
            pthread_mutex_lock(&g_mu);
            g_tbuf[bi].pieces_left = pieces;
            pthread_mutex_unlock(&g_mu);
            for (int p = 0; p < pieces; p++) {
                const uint64_t po = (uint64_t)p * WRITE_PIECE, pn = len - po < WRITE_PIECE ? len - po : WRITE_PIECE;
                const job_t j = {write_piece, (unsigned char *)g_tbuf[bi].buf + po, ((uint64_t)w->out_fd << 8) | (uint64_t)bi,
                                 c->text_off + off + po, pn};
                pool_push(&g_writers, j);
            }
        }
        w->text_ms += now_ms() - t0;
        chunk_done(c);
        return 0;
    }
    int rec_bytes = 0;
    uint64_t used = 0;
    if ((rc = pfac_scan_format(ctx, slot, &rec_bytes, &c->n_tiles, &used))) return fail(w, ctx, rc, "format");
    if (rec_bytes < 8) {                            /* 2 or 4 bytes per match over PCIe; the emitter prints from this form */
        c->word_bytes = rec_bytes;
        c->n_words = used;
        c->words = malloc((used ? used : 1) * (size_t)rec_bytes);
        c->tix = (uint64_t *)malloc((c->n_tiles ? c->n_tiles : 1) * sizeof(uint64_t));
        if (!c->words || !c->tix) return fail(w, NULL, PFAC_E_NOMEM, "out of host memory for records");
        if ((rc = pfac_records_d2h_packed(ctx, slot, NULL, c->words, used, c->tix))) return fail(w, ctx, rc, "d2h");
    } else {
        c->rec = (pfac_record *)malloc((n ? n : 1) * sizeof(pfac_record));
        if (!c->rec) return fail(w, NULL, PFAC_E_NOMEM, "out of host memory for records");
        if ((rc = pfac_records_d2h(ctx, slot, NULL, c->rec, 0, n))) return fail(w, ctx, rc, "d2h");
    }
    if ((rc = pfac_slot_sync(ctx, slot))) return fail(w, ctx, rc, "sync");
    chunk_done(c);
    return 0;
}

static void *worker(void *arg) {
    worker_t *w = (worker_t *)arg;
    pfac_ctx *ctx = NULL;
    uint64_t *cap = NULL;
    int *busy = NULL;               /* chunk index occupying each slot, or -1 */
    int freed = 0;                  /* this worker's chunks [0, freed) have given their staging buffers back */
    int issued = 0;                 /* ... and [0, issued) have had their H2D copies issued */
    const double ts = now_ms();
    int rc = pfac_ctx_create(w->device, w->n_streams, &ctx);
    if (rc) {
        char msg[200];
        snprintf(msg, sizeof msg, "context: %s", pfac_last_error(NULL));
        fail(w, NULL, rc, msg);
        return NULL;
    }
    cap = (uint64_t *)calloc((size_t)w->n_streams, sizeof(uint64_t));
    busy = (int *)malloc((size_t)w->n_streams * sizeof(int));
    for (int s = 0; s < w->n_streams; s++) busy[s] = -1;
    /* pinned staging and device buffers need no table: sized for the longest halo a table can ask for (patterns are shorter
     * than 1024 bytes).  The readers start on a staging buffer the moment it exists (and the chunk plan does). */
    if (w->ingest_mmap) {
        /* zero-copy ingest: can this driver pin the file's page-cache pages in place?  (one page, tried ONCE, by whichever
         * worker gets here first: two threads registering the same page would race) */
        static pthread_mutex_t probe_mu = PTHREAD_MUTEX_INITIALIZER;
        static pthread_t reg_thread;
        pthread_mutex_lock(&probe_mu);
        if (g_zero_copy < 0) {
            volatile unsigned char touch = g_map[0];
            (void)touch;
            g_zero_copy = pfac_host_register(g_map, 4096) == 0;
            if (g_zero_copy) {
                (void)pfac_host_unregister(g_map);
                if (pthread_create(&reg_thread, NULL, registrar, NULL) == 0) pthread_detach(reg_thread);
                else g_zero_copy = 0;
            }
        }
        w->ingest_mmap = g_zero_copy;
        pthread_mutex_unlock(&probe_mu);
    }
    if (w->ingest_mmap) {
        void *p = NULL;
        if ((rc = pfac_host_alloc(&p, (size_t)w->n_streams * 1024))) { fail(w, NULL, rc, "pinned halo buffer"); goto out; }
        w->halo_buf = (unsigned char *)p;
    }
    for (int b = 0; b < w->n_stage; b++) {
        if (!w->ingest_mmap) {
            void *p = NULL;
            if ((rc = pfac_host_alloc(&p, w->chunk_bytes + 1024 + 64))) { fail(w, NULL, rc, "pinned staging buffer"); goto out; }
            pthread_mutex_lock(&g_mu);
            w->stage[b].buf = p;
            w->stage[b].state = ST_FREE;
            schedule_reads(w);
            pthread_mutex_unlock(&g_mu);
        }
        if (b == 0)
            for (int s = 0; s < w->n_streams; s++) {
                cap[s] = w->chunk_bytes / 8 + 4096;
                if ((rc = pfac_slot_reserve(ctx, s, w->chunk_bytes + 1024, cap[s]))) { fail(w, ctx, rc, "reserve"); goto out; }
            }
    }
    w->setup_ms = now_ms() - ts;
    MILESTONE("worker %d: context, device buffers, %s ready", w->index, w->ingest_mmap ? "zero-copy ingest (page cache registered in place)" : "pinned staging buffers");
    {
        const double tw = now_ms();
        pthread_mutex_lock(&g_mu);
        while (!g_table_ready && !g_failed) pthread_cond_wait(&g_cv, &g_mu);
        const int stop = g_failed;
        pthread_mutex_unlock(&g_mu);
        if (stop) goto out;
        w->table_wait_ms = now_ms() - tw;
    }
    if ((rc = pfac_table_upload(ctx, g_blob, g_blob_words))) { fail(w, ctx, rc, "table upload"); goto out; }
    MILESTONE("worker %d: table installed", w->index);
    for (int j = 0; j < w->n_mine && !w->rc; j++) {
        const int k = w->index + j * w->n_workers;
        const int slot = j % w->n_streams;
        chunk_t *c = &w->chunks[k];
        const double td = now_ms();
        if (busy[slot] >= 0 && drain(w, ctx, slot, busy[slot], &cap[slot])) break;
        w->drain_ms += now_ms() - td;
        busy[slot] = -1;
        /* bounded memory: do not run further ahead of the in-order output than the window; then wait for the readers.
         * Staging buffers go back to the readers as soon as their copies have left them (asked without blocking, oldest
         * first), so the reads run ahead and several copies are always queued; the worker blocks on a copy only when the
         * buffer it needs next still holds an older chunk. */
        const double tr = now_ms();
        stage_t *st = &w->stage[j % w->n_stage];
        int stop = 0, ioerr = 0;
        pthread_mutex_lock(&g_mu);
        while (k >= g_emitted + g_window && !g_failed) pthread_cond_wait(&g_cv, &g_mu);
        for (;;) {
            int block = !w->ingest_mmap && st->state == ST_INFLIGHT;       /* (only ever true for the oldest unreleased chunk's buffer) */
            while (freed < j) {
                pthread_mutex_unlock(&g_mu);
                const int sl = freed % w->n_streams;
                const int done = block ? (pfac_slot_h2d_wait(ctx, sl) ? -1 : 1) : pfac_slot_h2d_done(ctx, sl);
                pthread_mutex_lock(&g_mu);
                if (done < 0) { rc = done; break; }
                if (!done) break;
                if (w->ingest_mmap) {                               /* the copy has left the page cache: its piece may be unpinned */
                    pthread_mutex_unlock(&g_mu);
                    piece_copy_done(w->chunks[w->index + freed * w->n_workers].base);
                    pthread_mutex_lock(&g_mu);
                } else {
                    w->stage[freed % w->n_stage].state = ST_FREE;
                }
                freed++;
                schedule_reads(w);
                block = 0;
            }
            if (rc < 0 || g_failed) break;
            if (w->ingest_mmap) {                                   /* zero-copy: has the registrar reached this chunk's piece? */
                if ((int)(c->base / g_piece_bytes) < g_reg_upto) break;
                pthread_cond_wait(&g_cv, &g_mu);
                continue;
            }
            if (st->state == ST_READY) break;
            if (st->state == ST_INFLIGHT) continue;     /* its copy was still running: wait for it (block) */
            struct timespec until;
            clock_gettime(CLOCK_REALTIME, &until);
            until.tv_nsec += 200000;                    /* the readers are at it: look again at the copies in 0.2 ms at the latest */
            if (until.tv_nsec >= 1000000000) { until.tv_nsec -= 1000000000; until.tv_sec++; }
            pthread_cond_timedwait(&g_cv, &g_mu, &until);
        }
        stop = g_failed;
        ioerr = w->ingest_mmap ? 0 : st->io_error;
        if (!stop && rc >= 0 && !w->ingest_mmap) st->state = ST_INFLIGHT;
        pthread_mutex_unlock(&g_mu);
        w->read_wait_ms += now_ms() - tr;
        if (rc < 0) { fail(w, ctx, rc, "h2d wait"); break; }
        if (stop) break;
        if (ioerr) { fail(w, NULL, PFAC_E_IO, "short read on the input file"); break; }
        if (g_skip_gpu) {                           /* diagnostic: the reader pool / the registrar alone */
            if (w->ingest_mmap) piece_copy_done(c->base);
            pthread_mutex_lock(&g_mu);
            if (!w->ingest_mmap) st->state = ST_FREE;
            freed = issued = j + 1;
            schedule_reads(w);
            pthread_mutex_unlock(&g_mu);
            chunk_done(c);
            continue;
        }
        const double ti = now_ms();
        if (w->ingest_mmap && c->n_avail) {
            /* straight out of the page cache (the chunk's piece is registered); the halo bytes may lie in the NEXT piece, a
             * registration of its own: they go through a small pinned buffer */
            unsigned char *src = g_map + c->base;
            const uint64_t main_n = c->n_avail < w->chunk_bytes ? c->n_avail : w->chunk_bytes, halo_n = c->n_avail - main_n;
            if ((rc = pfac_slot_h2d(ctx, slot, src, main_n, 0))) { fail(w, ctx, rc, "h2d"); break; }
            if (halo_n) {
                unsigned char *hb = w->halo_buf + (size_t)slot * 1024;
                memcpy(hb, src + main_n, (size_t)halo_n);
                if ((rc = pfac_slot_h2d(ctx, slot, hb, halo_n, main_n))) { fail(w, ctx, rc, "h2d (halo)"); break; }
            }
        } else if (c->n_avail && (rc = pfac_slot_h2d(ctx, slot, st->buf, c->n_avail, 0))) { fail(w, ctx, rc, "h2d"); break; }
        const double ti2 = now_ms();
        if ((rc = pfac_scan_async(ctx, slot, NULL, c->n_owned, c->n_avail, NULL, 0))) { fail(w, ctx, rc, "scan"); break; }
        w->issue_h2d_ms += ti2 - ti;
        w->issue_scan_ms += now_ms() - ti2;
        busy[slot] = k;
        issued = j + 1;
    }
    for (int s = 0; s < w->n_streams && !w->rc; s++) {      /* drain what is still in flight, oldest first */
        int lowest = -1;
        for (int t = 0; t < w->n_streams; t++)
            if (busy[t] >= 0 && (lowest < 0 || busy[t] < busy[lowest])) lowest = t;
        if (lowest < 0) break;
        drain(w, ctx, lowest, busy[lowest], &cap[lowest]);
        busy[lowest] = -1;
    }
out:
    MILESTONE("worker %d: last chunk drained", w->index);
    if (w->rc) set_failed();
    pool_wait_idle(&g_readers);     /* no reader may still write into the staging buffers freed below */
    if (w->ingest_mmap && !w->rc && !g_failed)      /* every scan has been waited for, so every copy has completed: the last pieces
                                                      * (only chunks whose copy was issued: unregistering what never was registered aborts in the runtime) */
        for (; freed < issued; freed++) piece_copy_done(w->chunks[w->index + freed * w->n_workers].base);
    for (int b = 0; b < w->n_stage; b++) pfac_host_free(w->stage[b].buf);
    pfac_host_free(w->halo_buf);
    free(cap); free(busy);
    pfac_ctx_destroy(ctx);
    return NULL;
}

int main(int argc, char *argv[]) {
    if (argc != 5) {                                            /* main.cc:93-96 */
        fprintf(stderr, "usage: %s <pattern file name> <streamnum> <PHF width> <input file name>\n", argv[0]);
        exit(-1);
    }
    const double t_start = now_ms();
    g_t_start = t_start;
    g_timeline = getenv("PFAC_TIMELINE") != NULL;
    if (getenv("PFAC_GPHF_SKIP")) { g_skip_read = strcmp(getenv("PFAC_GPHF_SKIP"), "read") == 0; g_skip_gpu = strcmp(getenv("PFAC_GPHF_SKIP"), "gpu") == 0; }
    int streamnum = atoi(argv[2]);
    int width = atoi(argv[3]);
    if (streamnum < 1) { fprintf(stderr, "streamnum must be >= 1\n"); return 1; }

    int fd = open(argv[4], O_RDONLY);                            /* main.cc:131-139 */
    struct stat st;
    if (fd < 0 || fstat(fd, &st)) { perror("Open input file failed."); return 1; }
    uint64_t N = st.st_size > 0 ? (uint64_t)st.st_size - 1 : 0;  /* the last byte is dropped, main.cc:138 */

    int rc, n_gpu = 0;
    MILESTONE("input opened");
    if ((rc = pfac_device_count(&n_gpu)) || n_gpu < 1) { fprintf(stderr, "no GPU available: %s\n", pfac_last_error(NULL)); return 1; }
    MILESTONE("HIP runtime up, %d device(s)", n_gpu);
    const char *lim = getenv("PFAC_GPUS");
    if (lim && atoi(lim) > 0 && atoi(lim) < n_gpu) n_gpu = atoi(lim);
    const int n_dev = n_gpu;
    const char *wpg = getenv("PFAC_WORKERS_PER_GPU");
    if (wpg && atoi(wpg) > 1 && atoi(wpg) <= 8) n_gpu *= atoi(wpg);      /* from here on n_gpu counts WORKERS */
    /* chunk size: the pipeline is bound by the host link (~55 GB/s), not by launches, so chunks stay small -- pinning the
     * staging buffers is what a short run pays for */
    uint64_t chunk = 32ull << 20;
    const char *cm = getenv("PFAC_CHUNK_MB");
    if (cm && atoll(cm) > 0) chunk = (uint64_t)atoll(cm) << 20;
    if (chunk > (1ull << 32)) chunk = 1ull << 32;
    const int n_chunks = (int)((N + chunk - 1) / chunk);
    if (n_chunks < n_gpu) n_gpu = n_chunks > 0 ? n_chunks : 1;
    chunk_t *chunks = (chunk_t *)calloc(n_chunks > 0 ? (size_t)n_chunks : 1, sizeof(chunk_t));
    g_window = 2 * n_gpu * streamnum + n_gpu;

    const char *output_file_name = "GPU_match_result.txt";       /* main.cc:335 */
    /* (an existing file is REMOVED, not truncated: ext4 allocates a truncated-and-rewritten file's blocks at close(), which
     * for gigabytes of text is a third of a second spent inside fclose) */
    (void)remove(output_file_name);
    FILE *fpout = fopen(output_file_name, "w");
    if (!fpout) { perror("Open output file failed.\n"); return 1; }
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    int emit_threads = ncpu > 16 ? 16 : (ncpu < 1 ? 1 : (int)ncpu);
    if (getenv("PFAC_EMIT_THREADS")) emit_threads = atoi(getenv("PFAC_EMIT_THREADS"));
    if (emit_threads < 1) emit_threads = 1;
    int read_threads = ncpu > 16 ? 16 : (ncpu < 1 ? 1 : (int)ncpu);
    if (getenv("PFAC_READ_THREADS")) read_threads = atoi(getenv("PFAC_READ_THREADS"));
    if (read_threads < 1) read_threads = 1;
    const char *em = getenv("PFAC_EMIT");
    const int emit_device = !(em && strcmp(em, "host") == 0);
    uint64_t piece = chunk / 4;
    if (piece > (4ull << 20)) piece = 4ull << 20;
    if (piece < (64ull << 10)) piece = 64ull << 10;
    piece &= ~4095ull;

    /* ingest: by default the file is MAPPED and its page-cache pages are registered for DMA chunk by chunk -- no CPU thread
     * ever copies the input (PFAC_INGEST=pread: the reader pool copies it into pinned staging buffers instead, also the
     * fallback where the mapping or the registration is refused) */
    const char *ing = getenv("PFAC_INGEST");
    if (!(ing && strcmp(ing, "pread") == 0) && st.st_size > 0) {
        void *m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
        if (m != MAP_FAILED) {
            g_map = (unsigned char *)m;
            g_map_len = ((uint64_t)st.st_size + 4095) & ~4095ull;
            const char *pm = getenv("PFAC_PIECE_MB");               /* bytes registered per hipHostRegister call (default 256 MiB) */
            const uint64_t piece = pm && atoi(pm) > 0 ? (uint64_t)atoi(pm) << 20 : 256ull << 20;
            const uint64_t per = piece / chunk;
            g_piece_bytes = chunk * (per ? per : 1);             /* whole chunks: a chunk's own bytes never straddle two registrations */
            g_n_pieces = (int)((g_map_len + g_piece_bytes - 1) / g_piece_bytes);
            g_piece_left = (int *)calloc((size_t)g_n_pieces + 1, sizeof(int));
            for (int k = 0; k < n_chunks; k++) g_piece_left[((uint64_t)k * chunk) / g_piece_bytes]++;
            for (int i = 0; i < g_n_pieces; i++) if (g_piece_left[i] == 0) g_piece_left[i] = -1;    /* (a tail piece no chunk starts in) */
        }
    }
    if (pool_start(&g_readers, read_threads, 1024) || pool_start(&g_writers, emit_device ? emit_threads : 1, 256)) {
        fprintf(stderr, "cannot start the I/O thread pools\n");
        return 1;
    }

    /* the workers create their contexts, device buffers and pinned staging while the table is being built here */
    const double t2 = now_ms();
    worker_t *ws = (worker_t *)calloc((size_t)n_gpu, sizeof(worker_t));
    pthread_t *th = (pthread_t *)malloc((size_t)n_gpu * sizeof(pthread_t));
    for (int g = 0; g < n_gpu; g++) {                            /* one host thread per GPU, main.cc:180-241 */
        worker_t *w = &ws[g];
        w->index = g; w->device = g % n_dev; w->n_workers = n_gpu; w->n_streams = streamnum;
        w->fd = fd; w->chunk_bytes = chunk; w->chunks = chunks; w->n_chunks = n_chunks;
        w->n_mine = n_chunks > g ? (n_chunks - g + n_gpu - 1) / n_gpu : 0;
        w->n_stage = streamnum + 1;
        w->stage = (stage_t *)calloc((size_t)w->n_stage, sizeof(stage_t));
        w->ingest_mmap = g_map != NULL;
        w->piece = piece; w->emit_device = emit_device; w->out_fd = fileno(fpout);
    }
    for (int g = 0; g < n_gpu; g++) pthread_create(&th[g], NULL, worker, &ws[g]);

    char err[256] = "";
    pfac_table *tab = NULL;
    int32_t *blob = NULL;
    const double t0 = now_ms();
    rc = pfac_table_build_file(argv[1], width, &tab, err, sizeof err);     /* main.cc:108,125 */
    const double t1 = now_ms();
    size_t words = 0;
    if (!rc) {
        words = pfac_table_blob_words(tab);
        blob = (int32_t *)malloc(words * sizeof(int32_t));
        if (!blob || pfac_table_to_blob(tab, blob, words)) { rc = PFAC_E_NOMEM; snprintf(err, sizeof err, "table image failed"); }
    }
    if (rc) {
        fprintf(stderr, "table build failed (%d): %s\n", rc, err);
        set_failed();
        for (int g = 0; g < n_gpu; g++) pthread_join(th[g], NULL);
        return 1;
    }
    printf("state num : %d\nfinal state num : %d\nmax pattern length : %d\n", tab->state_num, tab->num_final, tab->max_pat_len);
    printf("Number of keys    : %d\nwidth value       : %d\nr table size      : %7d\nHash table size   : %7d\n",
           tab->n_keys, tab->width, tab->max_row, tab->ht_size);
    printf("input size is %llu char\n", (unsigned long long)N);
    const uint64_t halo = tab->max_pat_len > 1 ? (uint64_t)tab->max_pat_len - 1 : 0;
    for (int k = 0; k < n_chunks; k++) {
        chunks[k].base = (uint64_t)k * chunk;
        chunks[k].n_owned = chunks[k].base + chunk <= N ? chunk : N - chunks[k].base;
        uint64_t end = chunks[k].base + chunks[k].n_owned + halo;
        if (end > N) end = N;                                    /* walks never read past the scanned bytes */
        chunks[k].n_avail = end - chunks[k].base;
    }
    pthread_mutex_lock(&g_mu);
    g_blob = blob; g_blob_words = words;
    g_table_ready = 1;
    MILESTONE("table built (%.1f ms), chunk plan published", t1 - t0);
    for (int g = 0; g < n_gpu; g++) schedule_reads(&ws[g]);      /* (for the staging buffers that exist already) */
    pthread_cond_broadcast(&g_cv);
    pthread_mutex_unlock(&g_mu);

    /* in chunk order: the host formatter prints here; with the device formatter only the frontier moves */
    double emit_ms = 0;
    int emit_failed = 0;
    for (int k = 0; k < n_chunks; k++) {
        pthread_mutex_lock(&g_mu);
        while (!chunks[k].done && !g_failed) pthread_cond_wait(&g_cv, &g_mu);
        const int ok = chunks[k].done;
        pthread_mutex_unlock(&g_mu);
        if (!ok) break;
        if (!emit_device) {
            double e0 = now_ms();
            const int64_t wrote = chunks[k].words
                ? pfac_emit_packed(fpout, chunks[k].words, chunks[k].n_words, chunks[k].word_bytes, chunks[k].tix, chunks[k].n_tiles, chunks[k].base, tab->idmap, emit_threads)
                : pfac_emit_records_mt(fpout, chunks[k].rec, chunks[k].n_rec, chunks[k].base, tab->idmap, emit_threads);
            if (wrote < 0) {
                fprintf(stderr, "write failed\n");
                emit_failed = 1;
                set_failed();
                break;
            }
            emit_ms += now_ms() - e0;
            free(chunks[k].rec); free(chunks[k].words); free(chunks[k].tix);
            chunks[k].rec = NULL; chunks[k].words = NULL; chunks[k].tix = NULL;
        }
        pthread_mutex_lock(&g_mu);
        g_emitted = k + 1;
        pthread_cond_broadcast(&g_cv);
        pthread_mutex_unlock(&g_mu);
    }
    MILESTONE("every chunk's results handed over");
    pool_wait_idle(&g_writers);
    const double t_results = now_ms();      /* the output is complete (what is left is teardown) */
    MILESTONE("writers idle: GPU_match_result.txt is complete");
    for (int g = 0; g < n_gpu; g++) pthread_join(th[g], NULL);
    MILESTONE("workers joined (contexts destroyed, host buffers unpinned)");
    pool_stop(&g_readers);
    pool_stop(&g_writers);
    if (g_write_error) { fprintf(stderr, "write failed\n"); emit_failed = 1; }
    fclose(fpout);
    MILESTONE("output closed");
    for (int i = 0; i < g_tbuf_n; i++) pfac_host_free(g_tbuf[i].buf);
    const double t3 = now_ms();
    MILESTONE("text buffers unpinned");
    double kernel_ms = 0, text_ms = 0;
    uint64_t total = 0;
    for (int g = 0; g < n_gpu; g++) {
        if (ws[g].rc) { fprintf(stderr, "%s\n", ws[g].err); return 1; }
        kernel_ms += ws[g].kernel_ms;
        text_ms += ws[g].text_ms;
        total += ws[g].matches;
    }
    if (emit_failed || g_failed) return 1;
    printf("/////////////////////////////////////////////\n");
    printf("0.Whole program: %lf mseconds\n", t3 - t_start);
    printf("1.Time for  create PFAC + Hashtable : %lf seconds (while the GPU contexts were being created)\n", (t1 - t0) / 1e3);
    printf("2.Time for  %d GPU match progress (%d worker(s); context + %s + H2D + kernel + D2H + emit, %d stream(s) each): %lf mseconds (%.3f GB/s end to end); teardown after it %.1f ms\n",
           n_dev < n_gpu ? n_dev : n_gpu, n_gpu, ws[0].ingest_mmap ? "page-cache mapping" : "read", streamnum, t_results - t2,
           t_results > t2 ? (double)N / (t_results - t2) / 1e6 : 0.0, t3 - t_results);
    printf("3.Kernel time summed over chunks: %lf mseconds (%.3f GB/s kernel-resident per GPU)\n", kernel_ms,
           kernel_ms > 0 ? (double)N / kernel_ms / 1e6 : 0.0);
    if (emit_device)
        printf("4.Time for  emit %llu matches (%llu bytes of text formatted on the GPU; D2H + pwrite by %d threads, overlapped with the scan): %lf mseconds\n",
               (unsigned long long)total, (unsigned long long)g_text_total, emit_threads, text_ms);
    else
        printf("4.Time for  emit %llu matches (host formatter on %d threads, overlapped with the scan): %lf mseconds\n", (unsigned long long)total, emit_threads, emit_ms);
    int retries = 0;
    for (int g = 0; g < n_gpu; g++) retries += ws[g].internal_retries;
    if (retries) printf("!! %d scan(s) were repeated after a kernel protocol timeout (PFAC_E_INTERNAL, see stderr): please report\n", retries);
    for (int g = 0; g < n_gpu; g++)
        printf("5.worker %d (GPU %d) host thread: setup (context, device buffers, pinned staging) %.1f ms, waiting for the table %.1f ms, for the %d readers %.1f ms, for scans/results %.1f ms\n",
               g, ws[g].device, ws[g].setup_ms, ws[g].table_wait_ms, read_threads, ws[g].read_wait_ms, ws[g].drain_ms);
    if (g_timeline)
        for (int g = 0; g < n_gpu; g++)
            fprintf(stderr, "worker %d: time inside pfac_slot_h2d calls %.1f ms, inside pfac_scan_async calls %.1f ms\n", g, ws[g].issue_h2d_ms, ws[g].issue_scan_ms);
    printf("matching process finshed\n");
    printf("/////////////////////////////////////////////\n");
    close(fd);
    free(blob);
    free(chunks);
    pfac_table_free(tab);
    return 0;
}

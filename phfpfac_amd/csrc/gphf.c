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
 *     to the visible GPUs, instead of the pattern set being partitioned
 *     (create_table_reorder.c:217-247);
 *   - streaming ingest: each GPU worker pread()s its next chunk into a pinned
 *     staging buffer of its pipeline slot, so the file is never resident as a
 *     whole (the reference reads it all into one cudaHostAlloc buffer,
 *     main.cc:147-155) and inputs larger than host RAM work;
 *   - <streamnum> really is the number of pipeline slots per GPU: chunk k+1 is
 *     read and copied H2D while chunk k is scanned and chunk k-1's records
 *     return (the reference creates streams, main.cc:209, and never uses them);
 *   - results come back as compact records -- 4 bytes per match plus 8 bytes per
 *     4 KiB of input (the tile index that orders them) -- not as a dense
 *     input_size x max_pat_len array (master_kernel.cu:235-236,428); an emitter
 *     thread prints finished chunks straight from that form, in input order,
 *     while later ones are still being scanned, so memory stays bounded.
 * There is no CPU matching path in this program: without a GPU it fails.
 *
 * Environment: PFAC_GPUS=n limits the number of GPUs used; PFAC_WORKERS_PER_GPU=m runs m independent workers (host
 * thread + context + pipeline slots each) on every GPU -- the chunks are dealt round-robin over all n * m workers, so the
 * multi-worker dealing and the shared in-order emitter can be exercised on a single device; PFAC_CHUNK_MB sets
 * the chunk size (default 64); PFAC_EMIT_THREADS the emitter's formatter threads;
 * PFAC_READ_THREADS the threads that pread() one chunk (default: cores / GPUs, at most 8).
 */
#define _FILE_OFFSET_BITS 64
#include "pfac.h"

#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef struct {
    uint64_t base;          /* global offset of the chunk's first owned byte */
    uint64_t n_owned, n_avail;
    pfac_record *rec;       /* host copy of the chunk's records: 8-byte form (automata beyond 2^20 final states) ... */
    void *words;            /* ... or the compact form: record heap (16- or 32-bit words) + tile index (pfac.h) */
    int word_bytes;
    uint64_t *tix;
    uint64_t n_tiles;
    uint64_t n_words;       /* heap words copied back (pfac_scan_format's *used) */
    uint64_t n_rec;
    int done;               /* guarded by g_mu */
} chunk_t;

typedef struct {
    int index, device, n_gpu, n_streams;    /* worker `index` of n_gpu workers (the name is historical: workers, not devices), on GPU `device` */
    const int32_t *blob;
    size_t blob_words;
    int fd;                         /* input file */
    uint64_t chunk_bytes, halo;
    chunk_t *chunks;                /* all chunks; this worker takes k = device, device + n_gpu, ... */
    int n_chunks;
    int read_threads;               /* threads that pread() one chunk into the pinned staging buffer */
    double kernel_ms, setup_ms, read_ms, drain_ms;   /* where this worker's wall time went */
    int internal_retries;           /* scans repeated after PFAC_E_INTERNAL (a protocol timeout: a bug, reported, never hidden) */
    int rc;
    char err[256];
} worker_t;

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_cv = PTHREAD_COND_INITIALIZER;
static int g_emitted = 0;           /* chunks the emitter has consumed */
static int g_window = 0;            /* a worker may run at most this many chunks ahead of the emitter */
static int g_failed = 0;

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
}

static int fail(worker_t *w, pfac_ctx *ctx, int rc, const char *what) {
    snprintf(w->err, sizeof w->err, "worker %d (GPU %d): %s: %s", w->index, w->device, what, ctx ? pfac_last_error(ctx) : "");
    w->rc = rc;
    pthread_mutex_lock(&g_mu);
    g_failed = 1;
    pthread_cond_broadcast(&g_cv);
    pthread_mutex_unlock(&g_mu);
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

/* One chunk is read by several threads, each pread()ing a 4 KiB-aligned slice: a single thread copies
 * out of the page cache at ~5 GB/s, far below what the H2D link and the scan take. */
typedef struct { int fd; unsigned char *dst; uint64_t n, off; int rc; } read_job;

static void *read_part(void *arg) {
    read_job *j = (read_job *)arg;
    j->rc = read_fully(j->fd, j->dst, j->n, j->off);
    return NULL;
}

static int read_parallel(int fd, void *dst, uint64_t n, uint64_t off, int n_threads) {
    enum { MAX_READERS = 32 };
    if (n_threads > MAX_READERS) n_threads = MAX_READERS;
    if (n_threads < 2 || n < (uint64_t)n_threads * (64u << 10)) return read_fully(fd, dst, n, off);
    read_job job[MAX_READERS];
    pthread_t th[MAX_READERS];
    int started[MAX_READERS];
    const uint64_t part = ((n + (uint64_t)n_threads - 1) / (uint64_t)n_threads + 4095) & ~4095ull;
    int used = 0, rc = 0;
    for (uint64_t at = 0; at < n; at += part, used++) {
        read_job *j = &job[used];
        j->fd = fd; j->dst = (unsigned char *)dst + at; j->off = off + at; j->rc = 0;
        j->n = n - at < part ? n - at : part;
        started[used] = used > 0 && pthread_create(&th[used], NULL, read_part, j) == 0;
    }
    for (int i = 0; i < used; i++)          /* slice 0, and any slice whose thread did not start, is read here */
        if (!started[i]) read_part(&job[i]);
    for (int i = 0; i < used; i++) {
        if (started[i]) pthread_join(th[i], NULL);
        if (job[i].rc) rc = -1;
    }
    return rc;
}

/* finish the chunk that occupies `slot`: wait, fetch count, (re-scan on overflow), copy records back, publish */
static int drain(worker_t *w, pfac_ctx *ctx, int slot, chunk_t *c, uint64_t *cap) {
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
    pthread_mutex_lock(&g_mu);
    c->done = 1;
    pthread_cond_broadcast(&g_cv);
    pthread_mutex_unlock(&g_mu);
    return 0;
}

static void *worker(void *arg) {
    worker_t *w = (worker_t *)arg;
    pfac_ctx *ctx = NULL;
    void **stage = NULL;            /* pinned staging buffer per slot */
    uint64_t *cap = NULL;
    int *busy = NULL;
    const double ts = now_ms();
    int rc = pfac_ctx_create(w->device, w->n_streams, &ctx);
    if (rc) {
        char msg[200];
        snprintf(msg, sizeof msg, "context: %s", pfac_last_error(NULL));
        fail(w, NULL, rc, msg);
        return NULL;
    }
    if ((rc = pfac_table_upload(ctx, w->blob, w->blob_words))) { fail(w, ctx, rc, "table upload"); goto out; }
    stage = (void **)calloc((size_t)w->n_streams, sizeof(void *));
    cap = (uint64_t *)calloc((size_t)w->n_streams, sizeof(uint64_t));
    busy = (int *)malloc((size_t)w->n_streams * sizeof(int));
    for (int s = 0; s < w->n_streams; s++) busy[s] = -1;
    for (int s = 0; s < w->n_streams; s++) {
        if ((rc = pfac_host_alloc(&stage[s], w->chunk_bytes + w->halo + 64))) { fail(w, NULL, rc, "pinned staging buffer"); goto out; }
        cap[s] = w->chunk_bytes / 8 + 4096;
        if ((rc = pfac_slot_reserve(ctx, s, w->chunk_bytes + w->halo, cap[s]))) { fail(w, ctx, rc, "reserve"); goto out; }
    }
    w->setup_ms = now_ms() - ts;
    for (int j = 0, k = w->index; k < w->n_chunks && !w->rc; j++, k += w->n_gpu) {
        const int slot = j % w->n_streams;
        chunk_t *c = &w->chunks[k];
        const double td = now_ms();
        if (busy[slot] >= 0 && drain(w, ctx, slot, &w->chunks[busy[slot]], &cap[slot])) break;
        w->drain_ms += now_ms() - td;
        busy[slot] = -1;
        /* bounded memory: do not run further ahead of the emitter than the window */
        pthread_mutex_lock(&g_mu);
        while (k >= g_emitted + g_window && !g_failed) pthread_cond_wait(&g_cv, &g_mu);
        const int stop = g_failed;
        pthread_mutex_unlock(&g_mu);
        if (stop) break;
        const double tr = now_ms();
        if (read_parallel(w->fd, stage[slot], c->n_avail, c->base, w->read_threads)) { fail(w, NULL, PFAC_E_IO, "short read on the input file"); break; }
        w->read_ms += now_ms() - tr;
        if ((rc = pfac_slot_h2d(ctx, slot, stage[slot], c->n_avail, 0))) { fail(w, ctx, rc, "h2d"); break; }
        if ((rc = pfac_scan_async(ctx, slot, NULL, c->n_owned, c->n_avail, NULL, 0))) { fail(w, ctx, rc, "scan"); break; }
        busy[slot] = k;
    }
    for (int s = 0; s < w->n_streams && !w->rc; s++) {      /* drain what is still in flight, oldest first */
        int lowest = -1;
        for (int t = 0; t < w->n_streams; t++)
            if (busy[t] >= 0 && (lowest < 0 || busy[t] < busy[lowest])) lowest = t;
        if (lowest < 0) break;
        drain(w, ctx, lowest, &w->chunks[busy[lowest]], &cap[lowest]);
        busy[lowest] = -1;
    }
out:
    if (stage) for (int s = 0; s < w->n_streams; s++) pfac_host_free(stage[s]);
    free(stage); free(cap); free(busy);
    pfac_ctx_destroy(ctx);
    return NULL;
}

int main(int argc, char *argv[]) {
    if (argc != 5) {                                            /* main.cc:93-96 */
        fprintf(stderr, "usage: %s <pattern file name> <streamnum> <PHF width> <input file name>\n", argv[0]);
        exit(-1);
    }
    int streamnum = atoi(argv[2]);
    int width = atoi(argv[3]);
    if (streamnum < 1) { fprintf(stderr, "streamnum must be >= 1\n"); return 1; }

    double t0 = now_ms();
    char err[256] = "";
    pfac_table *tab = NULL;
    int rc = pfac_table_build_file(argv[1], width, &tab, err, sizeof err);     /* main.cc:108,125 */
    if (rc) { fprintf(stderr, "table build failed (%d): %s\n", rc, err); return 1; }
    double t1 = now_ms();
    printf("state num : %d\nfinal state num : %d\nmax pattern length : %d\n", tab->state_num, tab->num_final, tab->max_pat_len);
    printf("Number of keys    : %d\nwidth value       : %d\nr table size      : %7d\nHash table size   : %7d\n",
           tab->n_keys, tab->width, tab->max_row, tab->ht_size);
    size_t words = pfac_table_blob_words(tab);
    int32_t *blob = (int32_t *)malloc(words * sizeof(int32_t));
    if (!blob || pfac_table_to_blob(tab, blob, words)) { fprintf(stderr, "table image failed\n"); return 1; }

    int fd = open(argv[4], O_RDONLY);                            /* main.cc:131-139 */
    struct stat st;
    if (fd < 0 || fstat(fd, &st)) { perror("Open input file failed."); return 1; }
    uint64_t N = st.st_size > 0 ? (uint64_t)st.st_size - 1 : 0;  /* the last byte is dropped, main.cc:138 */
    printf("input size is %llu char\n", (unsigned long long)N);

    int n_gpu = 0;
    if ((rc = pfac_device_count(&n_gpu)) || n_gpu < 1) { fprintf(stderr, "no GPU available: %s\n", pfac_last_error(NULL)); return 1; }
    const char *lim = getenv("PFAC_GPUS");
    if (lim && atoi(lim) > 0 && atoi(lim) < n_gpu) n_gpu = atoi(lim);
    const int n_dev = n_gpu;
    const char *wpg = getenv("PFAC_WORKERS_PER_GPU");
    if (wpg && atoi(wpg) > 1 && atoi(wpg) <= 8) n_gpu *= atoi(wpg);      /* from here on n_gpu counts WORKERS */
    uint64_t chunk = 64ull << 20;       /* small enough that pinning the staging buffers stays cheap, large enough to fill the GPU */
    const char *cm = getenv("PFAC_CHUNK_MB");
    if (cm && atoll(cm) > 0) chunk = (uint64_t)atoll(cm) << 20;
    if (chunk > (1ull << 32)) chunk = 1ull << 32;
    const uint64_t halo = tab->max_pat_len > 1 ? (uint64_t)tab->max_pat_len - 1 : 0;
    const int n_chunks = (int)((N + chunk - 1) / chunk);
    if (n_chunks < n_gpu) n_gpu = n_chunks > 0 ? n_chunks : 1;
    chunk_t *chunks = (chunk_t *)calloc(n_chunks > 0 ? (size_t)n_chunks : 1, sizeof(chunk_t));
    for (int k = 0; k < n_chunks; k++) {
        chunks[k].base = (uint64_t)k * chunk;
        chunks[k].n_owned = chunks[k].base + chunk <= N ? chunk : N - chunks[k].base;
        uint64_t end = chunks[k].base + chunks[k].n_owned + halo;
        if (end > N) end = N;                                    /* walks never read past the scanned bytes */
        chunks[k].n_avail = end - chunks[k].base;
    }
    g_window = 2 * n_gpu * streamnum + n_gpu;

    const char *output_file_name = "GPU_match_result.txt";       /* main.cc:335 */
    FILE *fpout = fopen(output_file_name, "w");
    if (!fpout) { perror("Open output file failed.\n"); return 1; }
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    int emit_threads = ncpu > 16 ? 16 : (ncpu < 1 ? 1 : (int)ncpu);
    if (getenv("PFAC_EMIT_THREADS")) emit_threads = atoi(getenv("PFAC_EMIT_THREADS"));

    int read_threads = (int)(ncpu / n_gpu);
    if (read_threads > 8) read_threads = 8;
    if (getenv("PFAC_READ_THREADS")) read_threads = atoi(getenv("PFAC_READ_THREADS"));
    if (read_threads < 1) read_threads = 1;

    double t2 = now_ms();
    worker_t *ws = (worker_t *)calloc((size_t)n_gpu, sizeof(worker_t));
    pthread_t *th = (pthread_t *)malloc((size_t)n_gpu * sizeof(pthread_t));
    for (int g = 0; g < n_gpu; g++) {                            /* one host thread per GPU, main.cc:180-241 */
        worker_t *w = &ws[g];
        w->index = g; w->device = g % n_dev; w->n_gpu = n_gpu; w->n_streams = streamnum; w->blob = blob; w->blob_words = words;
        w->fd = fd; w->chunk_bytes = chunk; w->halo = halo; w->chunks = chunks; w->n_chunks = n_chunks; w->read_threads = read_threads;
        pthread_create(&th[g], NULL, worker, w);
    }
    /* emitter: chunks in input order == position order; records of a chunk are already sorted */
    uint64_t total = 0;
    double emit_ms = 0;
    int emit_failed = 0;
    for (int k = 0; k < n_chunks; k++) {
        pthread_mutex_lock(&g_mu);
        while (!chunks[k].done && !g_failed) pthread_cond_wait(&g_cv, &g_mu);
        const int ok = chunks[k].done;
        pthread_mutex_unlock(&g_mu);
        if (!ok) break;
        double e0 = now_ms();
        const int64_t wrote = chunks[k].words
            ? pfac_emit_packed(fpout, chunks[k].words, chunks[k].n_words, chunks[k].word_bytes, chunks[k].tix, chunks[k].n_tiles, chunks[k].base, tab->idmap, emit_threads)
            : pfac_emit_records_mt(fpout, chunks[k].rec, chunks[k].n_rec, chunks[k].base, tab->idmap, emit_threads);
        if (wrote < 0) {
            fprintf(stderr, "write failed\n");
            emit_failed = 1;
            pthread_mutex_lock(&g_mu); g_failed = 1; pthread_cond_broadcast(&g_cv); pthread_mutex_unlock(&g_mu);
            break;
        }
        emit_ms += now_ms() - e0;
        total += chunks[k].n_rec;
        free(chunks[k].rec); free(chunks[k].words); free(chunks[k].tix);
        chunks[k].rec = NULL; chunks[k].words = NULL; chunks[k].tix = NULL;
        pthread_mutex_lock(&g_mu);
        g_emitted = k + 1;
        pthread_cond_broadcast(&g_cv);
        pthread_mutex_unlock(&g_mu);
    }
    for (int g = 0; g < n_gpu; g++) pthread_join(th[g], NULL);
    fclose(fpout);
    double t3 = now_ms();
    double kernel_ms = 0;
    for (int g = 0; g < n_gpu; g++) {
        if (ws[g].rc) { fprintf(stderr, "%s\n", ws[g].err); return 1; }
        kernel_ms += ws[g].kernel_ms;
    }
    if (emit_failed) return 1;
    printf("/////////////////////////////////////////////\n");
    printf("1.Time for  create PFAC + Hashtable : %lf seconds\n", (t1 - t0) / 1e3);
    printf("2.Time for  %d GPU match progress (%d worker(s); read + H2D + kernel + D2H + emit, %d stream(s) each): %lf mseconds (%.3f GB/s end to end)\n",
           n_dev < n_gpu ? n_dev : n_gpu, n_gpu, streamnum, t3 - t2, t3 > t2 ? (double)N / (t3 - t2) / 1e6 : 0.0);
    printf("3.Kernel time summed over chunks: %lf mseconds (%.3f GB/s kernel-resident per GPU)\n", kernel_ms,
           kernel_ms > 0 ? (double)N / kernel_ms / 1e6 : 0.0);
    printf("4.Time for  emit %llu matches (overlapped with the scan): %lf mseconds\n", (unsigned long long)total, emit_ms);
    int retries = 0;
    for (int g = 0; g < n_gpu; g++) retries += ws[g].internal_retries;
    if (retries) printf("!! %d scan(s) were repeated after a kernel protocol timeout (PFAC_E_INTERNAL, see stderr): please report\n", retries);
    for (int g = 0; g < n_gpu; g++)
        printf("5.worker %d (GPU %d) host thread: setup (context, table, pinned staging) %.1f ms, file read %.1f ms (%d threads), waiting for scans/readback %.1f ms\n",
               g, ws[g].device, ws[g].setup_ms, ws[g].read_ms, read_threads, ws[g].drain_ms);
    printf("matching process finshed\n");
    printf("/////////////////////////////////////////////\n");
    close(fd);
    free(blob);
    free(chunks);
    pfac_table_free(tab);
    return 0;
}

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
 *   - ONE automaton for the whole pattern file; the INPUT is sharded over the
 *     visible GPUs (contiguous byte ranges + max_pat_len-1 bytes of halo)
 *     instead of the pattern set (create_table_reorder.c:217-247);
 *   - <streamnum> really is the number of pipeline slots per GPU: chunk k+1 is
 *     copied H2D while chunk k is scanned and chunk k-1's records return
 *     (the reference creates streams, main.cc:209, and never uses them);
 *   - results come back as compact ordered records, not as a dense
 *     input_size x max_pat_len array (master_kernel.cu:235-236,428).
 * There is no CPU matching path in this program: without a GPU it fails.
 *
 * Environment: PFAC_GPUS=n limits the number of GPUs used; PFAC_CHUNK_MB sets
 * the pipeline chunk size (default 256).
 */
#include "pfac.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

typedef struct {
    uint64_t base;          /* global offset of the chunk's first owned byte */
    uint64_t n_owned, n_avail;
    pfac_record *rec;       /* host copy of the chunk's records */
    uint64_t n_rec;
} chunk_t;

typedef struct {
    int device, n_streams;
    const int32_t *blob;
    size_t blob_words;
    const unsigned char *input;     /* pinned host buffer, whole file */
    uint64_t N;                     /* total bytes scanned (filesize-1) */
    uint64_t lo, hi;                /* this GPU's owned range */
    uint64_t halo, chunk_bytes;
    chunk_t *chunks;
    int n_chunks;
    double kernel_ms;
    int rc;
    char err[256];
} worker_t;

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
}

static int fail(worker_t *w, pfac_ctx *ctx, int rc, const char *what) {
    snprintf(w->err, sizeof w->err, "GPU %d: %s: %s", w->device, what, pfac_last_error(ctx));
    w->rc = rc;
    return rc;
}

/* finish the chunk that occupies `slot`: wait, fetch count, (re-scan on overflow), copy records back */
static int drain(worker_t *w, pfac_ctx *ctx, int slot, chunk_t *c, uint64_t *cap) {
    uint64_t n = 0;
    int rc = pfac_scan_finish(ctx, slot, &n);
    if (rc == PFAC_E_OVERFLOW) {
        *cap = n + n / 8 + 4096;
        if ((rc = pfac_slot_reserve(ctx, slot, 0, *cap))) return fail(w, ctx, rc, "reserve");
        if ((rc = pfac_scan_async(ctx, slot, NULL, c->n_owned, c->n_avail, NULL, 0))) return fail(w, ctx, rc, "scan");
        rc = pfac_scan_finish(ctx, slot, &n);
    }
    if (rc) return fail(w, ctx, rc, "scan");
    float ms = 0;
    if (pfac_scan_elapsed_ms(ctx, slot, &ms) == 0) w->kernel_ms += ms;
    c->n_rec = n;
    c->rec = (pfac_record *)malloc((n ? n : 1) * sizeof(pfac_record));
    if (!c->rec) { w->rc = PFAC_E_NOMEM; snprintf(w->err, sizeof w->err, "out of host memory"); return w->rc; }
    if ((rc = pfac_records_d2h(ctx, slot, NULL, c->rec, 0, n))) return fail(w, ctx, rc, "d2h");
    if ((rc = pfac_slot_sync(ctx, slot))) return fail(w, ctx, rc, "sync");
    return 0;
}

static void *worker(void *arg) {
    worker_t *w = (worker_t *)arg;
    pfac_ctx *ctx = NULL;
    int rc = pfac_ctx_create(w->device, w->n_streams, &ctx);
    if (rc) { snprintf(w->err, sizeof w->err, "GPU %d: %s", w->device, pfac_last_error(NULL)); w->rc = rc; return NULL; }
    if ((rc = pfac_table_upload(ctx, w->blob, w->blob_words))) { fail(w, ctx, rc, "table upload"); goto out; }
    {
        uint64_t span = w->hi - w->lo;
        w->n_chunks = (int)((span + w->chunk_bytes - 1) / w->chunk_bytes);
        w->chunks = (chunk_t *)calloc(w->n_chunks ? (size_t)w->n_chunks : 1, sizeof(chunk_t));
        uint64_t *cap = (uint64_t *)calloc((size_t)w->n_streams, sizeof(uint64_t));
        int *busy = (int *)malloc((size_t)w->n_streams * sizeof(int));
        for (int s = 0; s < w->n_streams; s++) busy[s] = -1;
        for (int k = 0; k < w->n_chunks && !w->rc; k++) {
            int slot = k % w->n_streams;
            chunk_t *c = &w->chunks[k];
            c->base = w->lo + (uint64_t)k * w->chunk_bytes;
            c->n_owned = c->base + w->chunk_bytes <= w->hi ? w->chunk_bytes : w->hi - c->base;
            uint64_t end = c->base + c->n_owned + w->halo;
            if (end > w->N) end = w->N;                       /* walks never read past the scanned bytes */
            c->n_avail = end - c->base;
            if (busy[slot] >= 0 && drain(w, ctx, slot, &w->chunks[busy[slot]], &cap[slot])) break;
            busy[slot] = -1;
            if (cap[slot] == 0) cap[slot] = w->chunk_bytes / 8 + 4096;
            if ((rc = pfac_slot_reserve(ctx, slot, c->n_avail, cap[slot]))) { fail(w, ctx, rc, "reserve"); break; }
            if ((rc = pfac_slot_h2d(ctx, slot, w->input + c->base, c->n_avail, 0))) { fail(w, ctx, rc, "h2d"); break; }
            if ((rc = pfac_scan_async(ctx, slot, NULL, c->n_owned, c->n_avail, NULL, 0))) { fail(w, ctx, rc, "scan"); break; }
            busy[slot] = k;
        }
        for (int s = 0; s < w->n_streams && !w->rc; s++) {
            /* drain in chunk order of what is still in flight */
            int lowest = -1;
            for (int t = 0; t < w->n_streams; t++)
                if (busy[t] >= 0 && (lowest < 0 || busy[t] < busy[lowest])) lowest = t;
            if (lowest < 0) break;
            drain(w, ctx, lowest, &w->chunks[busy[lowest]], &cap[lowest]);
            busy[lowest] = -1;
        }
        free(cap); free(busy);
    }
out:
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

    FILE *fpin = fopen(argv[4], "rb");                           /* main.cc:131-155 */
    if (!fpin) { perror("Open input file failed."); return 1; }
    fseek(fpin, 0, SEEK_END);
    long fsz = ftell(fpin);
    rewind(fpin);
    uint64_t N = fsz > 0 ? (uint64_t)fsz - 1 : 0;                /* the last byte is dropped, main.cc:138 */
    printf("input size is %llu char\n", (unsigned long long)N);
    void *pinned = NULL;
    if ((rc = pfac_host_alloc(&pinned, N ? N : 1))) { fprintf(stderr, "pinned host alloc failed: %s\n", pfac_last_error(NULL)); return 1; }
    if (N && fread(pinned, 1, N, fpin) != N) { fprintf(stderr, "short read on %s\n", argv[4]); return 1; }
    fclose(fpin);

    int n_gpu = 0;
    if ((rc = pfac_device_count(&n_gpu)) || n_gpu < 1) { fprintf(stderr, "no GPU available: %s\n", pfac_last_error(NULL)); return 1; }
    const char *lim = getenv("PFAC_GPUS");
    if (lim && atoi(lim) > 0 && atoi(lim) < n_gpu) n_gpu = atoi(lim);
    uint64_t chunk = 256ull << 20;
    const char *cm = getenv("PFAC_CHUNK_MB");
    if (cm && atoll(cm) > 0) chunk = (uint64_t)atoll(cm) << 20;
    if (chunk > (1ull << 32)) chunk = 1ull << 32;
    /* shard boundaries on 16-byte multiples so every chunk pointer stays aligned */
    uint64_t per = (N + (uint64_t)n_gpu - 1) / (uint64_t)n_gpu;
    per = (per + 15) & ~15ull;
    if (N == 0) n_gpu = 1;
    else if ((N + per - 1) / per < (uint64_t)n_gpu) n_gpu = (int)((N + per - 1) / per);

    double t2 = now_ms();
    worker_t *ws = (worker_t *)calloc((size_t)n_gpu, sizeof(worker_t));
    pthread_t *th = (pthread_t *)malloc((size_t)n_gpu * sizeof(pthread_t));
    for (int g = 0; g < n_gpu; g++) {                            /* one host thread per GPU, main.cc:180-241 */
        worker_t *w = &ws[g];
        w->device = g; w->n_streams = streamnum; w->blob = blob; w->blob_words = words;
        w->input = (const unsigned char *)pinned; w->N = N;
        w->lo = (uint64_t)g * per; w->hi = w->lo + per < N ? w->lo + per : N;
        w->halo = tab->max_pat_len > 1 ? (uint64_t)tab->max_pat_len - 1 : 0;
        w->chunk_bytes = chunk;
        pthread_create(&th[g], NULL, worker, w);
    }
    for (int g = 0; g < n_gpu; g++) pthread_join(th[g], NULL);
    double t3 = now_ms();
    for (int g = 0; g < n_gpu; g++)
        if (ws[g].rc) { fprintf(stderr, "%s\n", ws[g].err); return 1; }

    const char *output_file_name = "GPU_match_result.txt";       /* main.cc:335 */
    FILE *fpout = fopen(output_file_name, "w");
    if (!fpout) { perror("Open output file failed.\n"); return 1; }
    uint64_t total = 0;
    double kernel_ms = 0;
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    int emit_threads = ncpu > 16 ? 16 : (ncpu < 1 ? 1 : (int)ncpu);
    if (getenv("PFAC_EMIT_THREADS")) emit_threads = atoi(getenv("PFAC_EMIT_THREADS"));
    for (int g = 0; g < n_gpu; g++) {                            /* shard order == position order */
        kernel_ms += ws[g].kernel_ms;
        for (int k = 0; k < ws[g].n_chunks; k++) {
            chunk_t *c = &ws[g].chunks[k];
            if (pfac_emit_records_mt(fpout, c->rec, c->n_rec, c->base, tab->idmap, emit_threads) < 0) { fprintf(stderr, "write failed\n"); return 1; }
            total += c->n_rec;
            free(c->rec);
        }
    }
    fclose(fpout);
    double t4 = now_ms();
    printf("/////////////////////////////////////////////\n");
    printf("1.Time for  create PFAC + Hashtable : %lf seconds\n", (t1 - t0) / 1e3);
    printf("2.Time for  %d GPU match progress (H2D + kernel + D2H, %d stream(s) each): %lf mseconds\n", n_gpu, streamnum, t3 - t2);
    printf("3.Kernel time summed over chunks: %lf mseconds (%.3f GB/s kernel-resident)\n", kernel_ms,
           kernel_ms > 0 ? (double)N / kernel_ms / 1e6 * n_gpu : 0.0);
    printf("4.Time for  emit %llu matches: %lf mseconds\n", (unsigned long long)total, t4 - t3);
    printf("matching process finshed\n");
    printf("/////////////////////////////////////////////\n");
    pfac_host_free(pinned);
    free(blob);
    pfac_table_free(tab);
    return 0;
}

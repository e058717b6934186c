/*
 * pfac_seam.cc -- GPU_Malloc_Memory / GPU_TraceTable / GPU_Free_memory (regex_GPU_PHF/main.cc:35-37) on top of the
 * C-ABI of pfac.h.  See include/pfac_seam.h.  Nothing here computes a match: the scan is the HIP kernel behind
 * pfac_scan_async, and every failure ends the process the way the reference's seam does.
 */
#include "pfac_seam.h"

#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "pfac.h"

namespace {

struct Seam {                   // what the six device pointers of the reference stand for
    pfac_ctx *ctx;
    unsigned long long capacity;
};

[[noreturn]] void die(const char *what, pfac_ctx *ctx) {
    fprintf(stderr, "%s: %s\n", what, pfac_last_error(ctx));      // master_kernel.cu:240-244: print, exit(1)
    exit(1);
}

}  // namespace

// master_kernel.cu:188-257: device buffers for one stream's chunk + (here) the chunk's tables
int GPU_Malloc_Memory(thread_data dataset, unsigned char **d_input_string, int **d_r, int **d_hash_table,
                      unsigned int **d_match_result, int **d_val_table, int **d_s0Table) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) { fprintf(stderr, "GPU_Malloc_Memory: no current HIP device\n"); exit(1); }
    Seam *s = new Seam{nullptr, 0};
    if (pfac_ctx_create(device, 1, &s->ctx)) die("GPU_Malloc_Memory: context", nullptr);
    pfac_table *tab = nullptr;                                    // the arrays FFDM() filled (main.cc:72-76,125), as they are
    if (pfac_table_from_reference_arrays(dataset.s0Table, dataset.r, dataset.HT, dataset.val, nullptr, dataset.width,
                                         dataset.state_num, dataset.final_state_num, dataset.HTSize, dataset.max_pat_len, &tab)) {
        fprintf(stderr, "GPU_Malloc_Memory: bad tables in thread_data\n");
        exit(1);
    }
    std::vector<int32_t> blob(pfac_table_blob_words(tab));
    if (pfac_table_to_blob(tab, blob.data(), blob.size())) { fprintf(stderr, "GPU_Malloc_Memory: table image\n"); exit(1); }
    pfac_table_free(tab);
    if (pfac_table_upload(s->ctx, blob.data(), blob.size())) die("GPU_Malloc_Memory: table upload", s->ctx);
    s->capacity = (unsigned long long)(dataset.input_size > 0 ? dataset.input_size : 0) / 4 + 65536;
    if (pfac_slot_reserve(s->ctx, 0, (uint64_t)(dataset.input_size > 0 ? dataset.input_size : 1), s->capacity))
        die("GPU_Malloc_Memory: device buffers", s->ctx);
    *d_input_string = reinterpret_cast<unsigned char *>(s);
    *d_r = *d_hash_table = *d_val_table = *d_s0Table = reinterpret_cast<int *>(s);
    *d_match_result = reinterpret_cast<unsigned int *>(s);
    return 0;
}

// master_kernel.cu:277-455: H2D, the scan, results back in the dense layout
int GPU_TraceTable(thread_data dataset, pfac_seam_stream /*unused, as in the reference*/, unsigned char *d_input_string, int *,
                   int *, unsigned int *, int *, int *) {
    Seam *s = reinterpret_cast<Seam *>(d_input_string);
    if (!s || !s->ctx || !dataset.match_result || dataset.input_size < 0) { fprintf(stderr, "GPU_TraceTable: bad arguments\n"); exit(1); }
    const uint64_t N = (uint64_t)dataset.input_size;
    if (N && pfac_slot_h2d(s->ctx, 0, dataset.input_string, N, 0)) die("GPU_TraceTable: H2D", s->ctx);
    uint64_t n = 0;
    int rc = PFAC_OK;
    for (int attempt = 0; attempt < 5; attempt++) {
        if (pfac_scan_async(s->ctx, 0, nullptr, N, N, nullptr, 0)) die("GPU_TraceTable: launch", s->ctx);
        rc = pfac_scan_finish(s->ctx, 0, &n);
        if (rc != PFAC_E_OVERFLOW) break;
        uint64_t hint = 0;
        if (pfac_scan_capacity_hint(s->ctx, 0, &hint)) die("GPU_TraceTable: capacity", s->ctx);
        s->capacity = hint > 2 * s->capacity ? hint : 2 * s->capacity;
        if (pfac_slot_reserve(s->ctx, 0, 0, s->capacity)) die("GPU_TraceTable: device buffers", s->ctx);
    }
    if (rc) die("GPU_TraceTable: scan", s->ctx);
    std::vector<pfac_record> rec(n ? n : 1);
    if (pfac_records_d2h(s->ctx, 0, nullptr, rec.data(), 0, n) || pfac_slot_sync(s->ctx, 0)) die("GPU_TraceTable: D2H", s->ctx);
    // slot j of position i = the j-th final state reached from i, the rest stays 0xFFFFFFFF (master_kernel.cu:67-70,236)
    const uint64_t L = (uint64_t)dataset.max_pat_len;
    memset(dataset.match_result, 0xFF, (size_t)(N * L) * sizeof(unsigned int));
    for (uint64_t k = 0; k < n;) {
        const uint64_t pos = rec[k].pos;
        for (uint64_t j = 0; k < n && rec[k].pos == pos; j++, k++)
            if (j < L) dataset.match_result[pos * L + j] = rec[k].state;
    }
    return 0;
}

// the overload a hipified main.cc:36 references (cudaStream_t -> hipStream_t); the stream stays unused
int GPU_TraceTable(thread_data dataset, ihipStream_t *, unsigned char *d_input_string, int *d_r, int *d_hash_table,
                   unsigned int *d_match_result, int *d_val_table, int *d_s0Table) {
    return GPU_TraceTable(dataset, static_cast<pfac_seam_stream>(nullptr), d_input_string, d_r, d_hash_table, d_match_result,
                          d_val_table, d_s0Table);
}

// master_kernel.cu:457-524
int GPU_Free_memory(unsigned char **d_input_string, int **d_r, int **d_hash_table, unsigned int **d_match_result,
                    int **d_val_table, int **d_s0Table) {
    Seam *s = d_input_string ? reinterpret_cast<Seam *>(*d_input_string) : nullptr;
    if (s) { pfac_ctx_destroy(s->ctx); delete s; }
    if (d_input_string) *d_input_string = nullptr;
    if (d_r) *d_r = nullptr;
    if (d_hash_table) *d_hash_table = nullptr;
    if (d_match_result) *d_match_result = nullptr;
    if (d_val_table) *d_val_table = nullptr;
    if (d_s0Table) *d_s0Table = nullptr;
    return 0;
}

/*
 * oracle/ref_harness.cc -- TEST INFRASTRUCTURE ONLY.
 *
 * Builds the REFERENCE's own host sources into oracle/_ref/libpfacref.so.
 * The reference files are compiled where they lie under /root/reference
 * (nothing is copied): this TU #includes
 *     regex_GPU_PHF/CreateTable/create_PFAC_table_reorder.c   (-> create_table_reorder.c, ctdef.h)
 *     regex_GPU_PHF/PHF/phf.c
 * exactly the way the reference's main.cc:5-6 does, and exposes what they
 * compute through a small extern "C" surface.  The CUDA parts of the reference
 * (master_kernel.cu) cannot be built here (no nvcc, no NVIDIA device); for a
 * complete run the harness hands the REFERENCE-BUILT tables to the oracle's
 * restatement of the kernel + merge + emit (pfac_oracle.c: tile_walk,
 * orc_scan_reference, orc_emit).
 *
 * Build notes (see oracle/Makefile):
 *   - compiled as C++ (phf.c:62 uses bare `RowStruct`), with <limits.h>
 *     pre-included (ctdef.h:9 uses INT_MAX), -w -fpermissive;
 *   - compiled at -O0: create_table_reorder() (ctr.c:201-251) is declared int
 *     and has no return statement, which g++ exploits at -O2;
 *   - INITIAL_PFAC_SIZE (ctr.c:10, a mutable global) is lowered before the call
 *     to avoid the 4 GiB-per-chunk preallocation; the reference's own doubling
 *     path (ctr.c:336-353) keeps behaviour identical.
 */
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>

#include "CreateTable/create_PFAC_table_reorder.c"
#include "PHF/phf.c"

#undef CHAR_SET
#include "pfac_oracle.h"

namespace {
struct Quiet {   // the reference printf()s table statistics; keep test logs readable
    int saved;
    Quiet() { fflush(stdout); saved = dup(1); int n = open("/dev/null", O_WRONLY); dup2(n, 1); close(n); }
    ~Quiet() { fflush(stdout); dup2(saved, 1); close(saved); }
};

orc_model *wrap(int P, int *state_num, int *final_num, int *max_len_arr, int max_len,
                int ***PFACs, int **idmaps) {
    orc_model *m = (orc_model *)calloc(1, sizeof *m);
    m->P = P;
    m->n_final = final_num; m->state_num = state_num; m->max_len_arr = max_len_arr;
    m->max_len = max_len; m->pfac = PFACs; m->idmap = idmaps;
    return m;
}
}  // namespace

extern "C" {

/* create_PFAC_table_reorder (cpt.c:6-11 -> ctr.c:201-251) as main.cc:108 calls
 * it: P = 4*streamnum chunks.  Returns an orc_model VIEW over the
 * reference-built arrays (never orc_free() it). */
orc_model *ref_build(const char *pattern_file, int streamnum) {
    Quiet q;
    INITIAL_PFAC_SIZE = 1 << 16;
    INITIAL_SIZE = 100000;
    int P = 4 * streamnum;
    int *state_num = (int *)calloc(P, sizeof(int));
    int *final_num = (int *)calloc(P, sizeof(int));
    int *max_len_arr = (int *)calloc(P, sizeof(int));
    int max_len = 0;
    int ***PFACs = (int ***)malloc(P * sizeof(int **));
    int **idmaps = (int **)malloc(P * sizeof(int *));
    create_PFAC_table_reorder((char *)pattern_file, state_num, final_num, streamnum,
                              max_len_arr, &max_len, PFACs, idmaps);
    return wrap(P, state_num, final_num, max_len_arr, max_len, PFACs, idmaps);
}

/* The same reference functions composed for ONE chunk (read_pattern ctr.c:53,
 * patternsToPFAC ctr.c:277): the shape of the older single-automaton build
 * whose logs the reference still holds (tmp.dat, experiment/*record). */
orc_model *ref_build_single(const char *pattern_file) {
    Quiet q;
    INITIAL_PFAC_SIZE = 1 << 16;
    INITIAL_SIZE = 100000;
    GPU_N = 1;
    int pattern_num = 0;
    pattern_s *all = (pattern_s *)malloc(INITIAL_SIZE * sizeof(pattern_s));
    all = read_pattern((char *)pattern_file, &pattern_num, all);
    int *state_num = (int *)calloc(1, sizeof(int));
    int *final_num = (int *)calloc(1, sizeof(int));
    int *max_len_arr = (int *)calloc(1, sizeof(int));
    int ***PFACs = (int ***)malloc(sizeof(int **));
    int **idmaps = (int **)malloc(sizeof(int *));
    PFACs[0] = (int **)malloc(INITIAL_PFAC_SIZE * sizeof(int *));
    idmaps[0] = (int *)malloc((pattern_num + 1) * sizeof(int));
    PFACs[0] = patternsToPFAC(all + 1, pattern_num, PFACs[0], &max_len_arr[0], &state_num[0], idmaps[0]);
    final_num[0] = pattern_num;
    orc_model *m = wrap(1, state_num, final_num, max_len_arr, max_len_arr[0], PFACs, idmaps);
    m->n_pat = pattern_num;
    return m;
}

/* The reference's escape-aware reader read_pattern_ext (ctr.c:131-185, fgetc_ext ctdef.h:37-99) -- dead code
 * there, called here directly -- then its trie builder over ONE chunk. */
orc_model *ref_build_single_ext(const char *pattern_file) {
    Quiet q;
    INITIAL_PFAC_SIZE = 1 << 16;
    INITIAL_SIZE = 100000;
    GPU_N = 1;
    int pattern_num = 0;
    pattern_s *all = (pattern_s *)malloc(INITIAL_SIZE * sizeof(pattern_s));
    read_pattern_ext((char *)pattern_file, &pattern_num, all);
    int *state_num = (int *)calloc(1, sizeof(int));
    int *final_num = (int *)calloc(1, sizeof(int));
    int *max_len_arr = (int *)calloc(1, sizeof(int));
    int ***PFACs = (int ***)malloc(sizeof(int **));
    int **idmaps = (int **)malloc(sizeof(int *));
    PFACs[0] = (int **)malloc(INITIAL_PFAC_SIZE * sizeof(int *));
    idmaps[0] = (int *)malloc((pattern_num + 1) * sizeof(int));
    PFACs[0] = patternsToPFAC(all + 1, pattern_num, PFACs[0], &max_len_arr[0], &state_num[0], idmaps[0]);
    final_num[0] = pattern_num;
    orc_model *m = wrap(1, state_num, final_num, max_len_arr, max_len_arr[0], PFACs, idmaps);
    m->n_pat = pattern_num;
    return m;
}

/* FFDM (phf.c:151) per chunk, as main.cc:123-126 calls it (serially here). */
int ref_ffdm(orc_model *m, int width) {
    Quiet q;
    int P = m->P;
    if (!m->r) {
        m->r = (int **)calloc(P, sizeof(int *)); m->HT = (int **)calloc(P, sizeof(int *));
        m->val = (int **)calloc(P, sizeof(int *)); m->HTSize = (int *)calloc(P, sizeof(int));
        m->MaxRow = (int *)calloc(P, sizeof(int)); m->NumKeys = (int *)calloc(P, sizeof(int));
        m->MaxKey = (int *)calloc(P, sizeof(int)); m->MaxOffset = (int *)calloc(P, sizeof(int));
        for (int c = 0; c < P; c++) {                        /* main.cc:72-76 */
            m->r[c] = (int *)malloc(ROW_MAX * sizeof(int));
            m->HT[c] = (int *)malloc(HASHTABLE_MAX * sizeof(int));
            m->val[c] = (int *)malloc(HASHTABLE_MAX * sizeof(int));
        }
    }
    m->width = width;
    for (int c = 0; c < P; c++) {
        m->HTSize[c] = FFDM(m->pfac[c], m->state_num[c], width, m->r[c], m->HT[c], m->val[c]);
        m->MaxRow[c] = (m->state_num[c] * 256) / width + 1;  /* master_kernel.cu:212 */
    }
    return 0;
}

/* Whole program: reference table build + FFDM, then the oracle's restatement
 * of kernel/merge/emit over those tables.  N = filesize-1 (main.cc:138). */
long long ref_run(const char *pattern_file, int streamnum, int width, const char *input_file, const char *out_path) {
    orc_model *m = ref_build(pattern_file, streamnum);
    ref_ffdm(m, width);
    FILE *f = fopen(input_file, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long n = ftell(f) - 1;
    rewind(f);
    unsigned char *buf = (unsigned char *)malloc(n > 0 ? n : 1);
    if (n > 0 && fread(buf, 1, n, f) != (size_t)n) return -2;
    fclose(f);
    orc_matches *o = orc_matches_new();
    if (orc_scan_reference(m, buf, n, o)) return -3;
    long long cnt = o->n;
    if (orc_emit(o, out_path) < 0) return -4;
    orc_matches_free(o);
    free(buf);
    return cnt;
}

}  // extern "C"

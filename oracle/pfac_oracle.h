/*
 * oracle/pfac_oracle.h -- TEST INFRASTRUCTURE ONLY (see pfac_oracle.c header).
 * Types shared by the CPU restatement (pfac_oracle.c), the serial Aho-Corasick
 * baseline (ac_serial.c) and the reference harness (ref_harness.cc).
 */
#ifndef PFAC_ORACLE_H
#define PFAC_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CHAR_SET 256
#define REF_BLOCK_SIZE 512          /* master_kernel.cu:8  */
#define REF_PAGE_SIZE_C 4096        /* master_kernel.cu:9-10 */
#define REF_EXTRA_BYTES 512         /* master_kernel.cu:11 (128 ints) */
#define REF_ROW_MAX 1048576         /* phf.c:7 */
#define REF_COL_MAX 4096            /* phf.c:8 */
#define REF_HASHTABLE_MAX (163840 * 20) /* phf.c:10 */

typedef struct {
    int id;             /* 1-based line number (ctr.c:81,100) */
    int len;
    unsigned char *pat;
} orc_pat;

typedef struct {
    int n_pat;
    orc_pat *pats;      /* sorted, 0-based here (reference stores from index 1) */
    int P;              /* number of pattern chunks = gpu_s * streamnum (ctr.c:217) */
    int *n_final;       /* [P] patterns (= final states) per chunk */
    int *state_num;     /* [P] */
    int *max_len_arr;   /* [P] */
    int max_len;
    int ***pfac;        /* [P][state][256] dense tries, -1 = no edge */
    int **idmap;        /* [P][final_state] -> pattern id */
    /* PHF (after orc_ffdm) */
    int width;
    int **r, **HT, **val;
    int *HTSize, *MaxRow, *NumKeys, *MaxKey, *MaxOffset;
    char err[256];
} orc_model;

typedef struct {
    int64_t n;          /* number of matches */
    int64_t cap;
    int64_t *pos;
    int32_t *id;
} orc_matches;


orc_model *orc_build(const char *pattern_file, int streamnum, int gpu_s);
orc_model *orc_build_ext(const char *pattern_file, int streamnum, int gpu_s);
const char *orc_error(const orc_model *m);
int orc_ffdm(orc_model *m, int width, int exact);
int orc_phf_lookup(const orc_model *m, int c, int state, int ch);
orc_matches *orc_matches_new(void);
void orc_matches_free(orc_matches *o);
int orc_scan_reference(orc_model *m, const unsigned char *input, int64_t N, orc_matches *out);
int orc_scan_spec(const orc_model *m, const unsigned char *input, int64_t N, orc_matches *out);
int64_t orc_emit(const orc_matches *o, const char *path);
void orc_free(orc_model *m);

#ifdef __cplusplus
}
#endif
#endif

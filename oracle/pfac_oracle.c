/*
 * oracle/pfac_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the reference PFAC pipeline
 * (mickeyjoe666/PHFPFAC, regex_GPU_PHF/), written from its behaviour, used
 * ONLY as the checker by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  Nothing under phfpfac_amd/ may import, link or execute it.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle.py)
 *   - against the table statistics the reference's own run logs hold
 *     (tmp.dat:2-12, experiment/{xaa,xab,xac,xad}record:2-12,
 *     experiment/englishdicall:2-12: state num, #keys, max key, max offset,
 *     r size, hash-table size),
 *   - against the duplicate-id rule its recorded outputs hold
 *     (experiment/GPU_match_resultxab.txt -> "may" = 1777),
 *   - and against oracle/_ref (the reference's real host sources compiled
 *     where they lie; see oracle/ref_harness.cc) array-for-array and
 *     output-byte-for-byte; the outputs are committed under tests/golden/.
 *
 * What each function restates (paths relative to regex_GPU_PHF/):
 *   orc_read_patterns   CreateTable/create_table_reorder.c:53-122  (read_pattern)
 *   orc_cmp_pat         CreateTable/create_table_reorder.c:21-45   (comp_pat)
 *   orc_build           CreateTable/create_table_reorder.c:201-274 (create_table_reorder, divide_patterns)
 *   build_trie          CreateTable/create_table_reorder.c:277-378 (patternsToPFAC)
 *   orc_ffdm            PHF/phf.c:62-291                           (InitArrays, ReadKey, SortRows, FFDM)
 *   tile_walk           master_kernel.cu:37-74, 92-180             (SUBSEG_MATCH, TraceTable_kernel)
 *   orc_scan_reference  master_kernel.cu:277-455 + main.cc:304-324 (GPU_TraceTable geometry, merge)
 *   orc_emit            main.cc:335-350                            (text emitter)
 *   orc_scan_spec       the semantics of SURVEY.md section 8(a) "parity domain",
 *                       walking the dense trie directly (no PHF, no tiles).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pfac_oracle.h"

/* ------------------------------------------------------------------ */
/* comp_pat: memcmp on the common prefix, then shorter first (ctr.c:21-45) */
static int orc_cmp_pat(const orc_pat *a, const orc_pat *b) {
    int min_len = a->len < b->len ? a->len : b->len;
    int r = memcmp(a->pat, b->pat, (size_t)min_len);
    if (r == 0) {
        if (a->len > b->len) return 1;
        if (a->len < b->len) return -1;
        return 0;
    }
    return r;
}

/* The reference calls glibc qsort (ctr.c:116), which is a merge sort whenever
 * its scratch buffer can be allocated, i.e. stable: equal strings keep file
 * order, so the LATER line gets the later final state and wins (ctr.c:366).
 * Restated as an explicit stable merge sort so the rule does not depend on libc. */
static void stable_sort(orc_pat *a, orc_pat *tmp, int n) {
    if (n < 2) return;
    int h = n / 2;
    stable_sort(a, tmp, h);
    stable_sort(a + h, tmp, n - h);
    int i = 0, j = h, k = 0;
    while (i < h && j < n) tmp[k++] = (orc_cmp_pat(&a[j], &a[i]) < 0) ? a[j++] : a[i++];
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, (size_t)n * sizeof(orc_pat));
}

/* read_pattern (ctr.c:53-122): bytes up to '\n' form one pattern; id is the
 * running 1-based count; length must stay below 1024; the file must end in
 * '\n' (otherwise the reference accumulates EOF bytes until the 1024 check
 * trips and exits).  Returns 0 or a negative error. */
/* fgetc_ext (ctdef.h:37-99): backslash escapes merged into one byte; a real newline becomes EOL (0x10A). */
#define ORC_EOL 0x10A
static int orc_fgetc_ext(FILE *fp) {
    int ch0 = fgetc(fp), ch1;
    int value = 0;
    if (ch0 == '\\') {
        ch1 = fgetc(fp);
        if (feof(fp)) return ch0;
        if (ch1 >= '0' && ch1 <= '9') {                     /* isdigit: "%3o" then fails on 8 and 9, value stays 0 */
            ungetc(ch1, fp);
            if (fscanf(fp, "%3o", (unsigned *)&value) != 1) value = 0;
            return (int)((char)value);
        }
        switch (ch1) {
            case 'a': return '\a';
            case 'b': return '\b';
            case 't': return '\t';
            case 'n': return '\n';
            case 'v': return '\v';
            case 'f': return '\f';
            case 'r': return '\r';
            case '\'': case '\"': case '\\': return ch1;
            case 'x':
                if (fscanf(fp, "%2x", (unsigned *)&value) != 1) value = 0;
                return (int)((char)value);
            default:
                ungetc(ch1, fp);
                return ch0;
        }
    }
    if (ch0 == '\n') return ORC_EOL;
    return ch0;
}

static int orc_ext_mode = 0;   /* set by orc_build_ext around orc_read_patterns: read_pattern_ext (ctr.c:131-185) */

static int orc_read_patterns(orc_model *m, const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { snprintf(m->err, sizeof m->err, "cannot open pattern file %s", path); return -1; }
    int cap = 1024, n = 0;
    orc_pat *p = (orc_pat *)malloc((size_t)cap * sizeof *p);
    unsigned char str[1024];
    for (;;) {
        int len = 0, ch;
        for (;;) {
            ch = orc_ext_mode ? orc_fgetc_ext(f) : fgetc(f);
            str[len++] = (unsigned char)ch;
            if (len >= 1024) {
                snprintf(m->err, sizeof m->err, "Pattern %d length over 1024.", n + 1);
                fclose(f); free(p); return -2;
            }
            if (ch == (orc_ext_mode ? ORC_EOL : '\n')) { len -= 1; n += 1; break; }
        }
        if (n > cap) { cap *= 2; p = (orc_pat *)realloc(p, (size_t)cap * sizeof *p); }
        p[n - 1].id = n;
        p[n - 1].len = len;
        p[n - 1].pat = (unsigned char *)malloc(len ? (size_t)len : 1);
        memcpy(p[n - 1].pat, str, (size_t)len);
        ch = fgetc(f);
        if (feof(f)) break;
        ungetc(ch, f);
    }
    fclose(f);
    orc_pat *tmp = (orc_pat *)malloc((size_t)n * sizeof *tmp);
    stable_sort(p, tmp, n);
    free(tmp);
    m->n_pat = n;
    m->pats = p;
    return 0;
}

/* patternsToPFAC (ctr.c:277-378): finals first -- final state i in [0,n) is the
 * i-th pattern of the sorted chunk, state n is unused, root = n+1, internal
 * states from n+2 in creation order; the last byte of pattern i sets
 * PFAC[state][ch] = i unconditionally (ctr.c:366). */
static int **build_trie(const orc_pat *pats, int n, int *max_len, int *state_num, int *idmap) {
    int initial_state = n + 1;
    int state_count = initial_state + 1;
    int cap = state_count + 1024;
    int **T = (int **)malloc((size_t)cap * sizeof *T);
    for (int x = 0; x < cap; x++) {
        T[x] = (int *)malloc(CHAR_SET * sizeof(int));
        memset(T[x], 0xFF, CHAR_SET * sizeof(int));
    }
    int state = initial_state;
    for (int i = 0; i < n; i++) {
        const orc_pat *cur = &pats[i];
        idmap[i] = cur->id;
        if (cur->len > *max_len) *max_len = cur->len;
        int j;
        for (j = 0; j < cur->len - 1; j++) {
            int ch = cur->pat[j];
            if (T[state][ch] == -1) {
                T[state][ch] = state_count;
                state = state_count;
                state_count += 1;
                if (state_count >= cap) {
                    int ncap = cap * 2;
                    T = (int **)realloc(T, (size_t)ncap * sizeof *T);
                    for (int x = cap; x < ncap; x++) {
                        T[x] = (int *)malloc(CHAR_SET * sizeof(int));
                        memset(T[x], 0xFF, CHAR_SET * sizeof(int));
                    }
                    cap = ncap;
                }
            } else {
                state = T[state][ch];
            }
        }
        T[state][cur->pat[j]] = i;
        state = initial_state;
    }
    /* rows [state_count, cap) are never referenced again; keep them so that a
     * lookup of any state < state_count stays valid, free the rest lazily. */
    for (int x = state_count; x < cap; x++) { free(T[x]); T[x] = NULL; }
    *state_num = state_count;
    return T;
}

/* create_table_reorder + divide_patterns (ctr.c:201-274): P = gpu_s*streamnum
 * contiguous chunks of the sorted list; chunks 0..P-2 hold k = n/P patterns,
 * the last k + n%P.  The reference hard-codes gpu_s = 4 (ctr.c:207). */
orc_model *orc_build(const char *pattern_file, int streamnum, int gpu_s) {
    orc_model *m = (orc_model *)calloc(1, sizeof *m);
    if (orc_read_patterns(m, pattern_file) != 0) return m;
    int P = gpu_s * streamnum;
    if (P < 1) { snprintf(m->err, sizeof m->err, "bad chunk count %d", P); return m; }
    m->P = P;
    m->n_final = (int *)calloc((size_t)P, sizeof(int));
    m->state_num = (int *)calloc((size_t)P, sizeof(int));
    m->max_len_arr = (int *)calloc((size_t)P, sizeof(int));
    m->pfac = (int ***)calloc((size_t)P, sizeof(int **));
    m->idmap = (int **)calloc((size_t)P, sizeof(int *));
    int k = m->n_pat / P;
    int l = k + m->n_pat % P;
    for (int c = 0; c < P; c++) {
        int cnt = (c == P - 1) ? l : k;
        m->idmap[c] = (int *)malloc((size_t)(cnt ? cnt : 1) * sizeof(int));
        m->pfac[c] = build_trie(m->pats + (size_t)c * k, cnt, &m->max_len_arr[c], &m->state_num[c], m->idmap[c]);
        if (m->max_len_arr[c] > m->max_len) m->max_len = m->max_len_arr[c];
        m->n_final[c] = cnt;
    }
    return m;
}

/* same pipeline with read_pattern_ext() as the reader (dead code in the reference: nothing calls it) */
orc_model *orc_build_ext(const char *pattern_file, int streamnum, int gpu_s) {
    orc_ext_mode = 1;
    orc_model *m = orc_build(pattern_file, streamnum, gpu_s);
    orc_ext_mode = 0;
    return m;
}

const char *orc_error(const orc_model *m) { return m->err[0] ? m->err : NULL; }
int orc_num_chunks(const orc_model *m) { return m->P; }
int orc_num_patterns(const orc_model *m) { return m->n_pat; }
int orc_max_len(const orc_model *m) { return m->max_len; }
int orc_state_num(const orc_model *m, int c) { return m->state_num[c]; }
int orc_final_num(const orc_model *m, int c) { return m->n_final[c]; }
int orc_chunk_max_len(const orc_model *m, int c) { return m->max_len_arr[c]; }
const int *orc_trie_row(const orc_model *m, int c, int state) { return m->pfac[c][state]; }
const int *orc_idmap(const orc_model *m, int c) { return m->idmap[c]; }
int orc_sorted_pattern(const orc_model *m, int i, int *id, int *len, const unsigned char **bytes) {
    if (i < 0 || i >= m->n_pat) return -1;
    *id = m->pats[i].id; *len = m->pats[i].len; *bytes = m->pats[i].pat;
    return 0;
}

/* ------------------------------------------------------------------ */
/* FFDM (phf.c:151-291).  exact != 0 reproduces the reference's table LAYOUT
 * (its O(rows^2) exchange sort, phf.c:126-139, is not stable, so the layout
 * depends on that exact loop); exact == 0 uses a stable O(R log R) order and
 * is for pattern sets where the quadratic sort is infeasible -- the layout
 * then differs but every lookup returns the same value. */
typedef struct { int row, cnt; int *cols; } ffdm_row;

static int row_cmp_desc(const void *a, const void *b) {
    const ffdm_row *x = (const ffdm_row *)a, *y = (const ffdm_row *)b;
    if (x->cnt != y->cnt) return y->cnt - x->cnt;
    return x->row - y->row;
}

static int ffdm_one(orc_model *m, int c, int width, int exact) {
    int **ary = m->pfac[c];
    int ary_size = m->state_num[c];
    if (width > REF_COL_MAX || width < 1) { snprintf(m->err, sizeof m->err, "width may not exceed %d", REF_COL_MAX); return -1; }
    int64_t nkeys_all = (int64_t)ary_size * CHAR_SET;
    int maxrow_bound = (int)((nkeys_all + width - 1) / width) + 1;
    if (maxrow_bound > REF_ROW_MAX) { snprintf(m->err, sizeof m->err, "Row > ROW_MAX(%d)", REF_ROW_MAX); return -3; }
    ffdm_row *Row = (ffdm_row *)calloc((size_t)maxrow_bound, sizeof *Row);
    for (int i = 0; i < maxrow_bound; i++) Row[i].row = i;
    int NumKeys = 0, MaxKey = 0;
    /* ReadKey (phf.c:90-117): pass 1 counts, pass 2 fills (same column order) */
    for (int64_t key = 0; key < nkeys_all; key++) {
        if (ary[key / CHAR_SET][key % CHAR_SET] < 0) continue;
        Row[key / width].cnt++;
        NumKeys++;
        if (key > MaxKey) MaxKey = (int)key;
    }
    for (int i = 0; i < maxrow_bound; i++) {
        if (Row[i].cnt) Row[i].cols = (int *)malloc((size_t)Row[i].cnt * sizeof(int));
        Row[i].cnt = 0;
    }
    for (int64_t key = 0; key < nkeys_all; key++) {
        if (ary[key / CHAR_SET][key % CHAR_SET] < 0) continue;
        ffdm_row *R = &Row[key / width];
        R->cols[R->cnt++] = (int)(key % width);
    }
    int MaxRow = MaxKey / width + 1;                       /* phf.c:174 */
    if (exact) {                                           /* SortRows, phf.c:126-139 */
        for (int i = 0; i < MaxRow - 1; i++)
            for (int j = i + 1; j < MaxRow; j++)
                if (Row[i].cnt < Row[j].cnt) { ffdm_row t = Row[i]; Row[i] = Row[j]; Row[j] = t; }
    } else {
        qsort(Row, (size_t)MaxRow, sizeof *Row, row_cmp_desc);
    }
    int *r = (int *)malloc((size_t)MaxRow * sizeof(int));
    memset(r, 0xFF, (size_t)MaxRow * sizeof(int));          /* phf.c:67 */
    int htcap = NumKeys + 2 * width + 16;
    if (htcap > REF_HASHTABLE_MAX) htcap = REF_HASHTABLE_MAX;
    if (exact) htcap = REF_HASHTABLE_MAX;                   /* same search bound as phf.c:188 */
    int *HT = (int *)malloc((size_t)htcap * sizeof(int));
    int *val = (int *)malloc((size_t)htcap * sizeof(int));
    memset(HT, 0xFF, (size_t)htcap * sizeof(int));
    memset(val, 0xFF, (size_t)htcap * sizeof(int));
    int MaxOffset = 0;
    int first_free = 0;  /* exact==0 only: no slot below this index is free */
    for (int ndx = 0; ndx < MaxRow && Row[ndx].cnt > 0; ndx++) {   /* phf.c:184 */
        int row = Row[ndx].row, cnt = Row[ndx].cnt, *cols = Row[ndx].cols;
        int offset = -cols[0];                              /* phf.c:188 */
        if (!exact && first_free - cols[0] > offset) offset = first_free - cols[0];
        for (;; offset++) {
            if (offset >= htcap - width) {
                if (exact) { snprintf(m->err, sizeof m->err, "failed to fit row %d into the hash table", row); return -4; }
                int ncap = htcap * 2;
                HT = (int *)realloc(HT, (size_t)ncap * sizeof(int));
                val = (int *)realloc(val, (size_t)ncap * sizeof(int));
                memset(HT + htcap, 0xFF, (size_t)(ncap - htcap) * sizeof(int));
                memset(val + htcap, 0xFF, (size_t)(ncap - htcap) * sizeof(int));
                htcap = ncap;
            }
            int i;
            for (i = 0; i < cnt; i++) if (HT[offset + cols[i]] != -1) break;
            if (i == cnt) break;
        }
        r[row] = offset;                                    /* phf.c:197 */
        if (offset > MaxOffset) MaxOffset = offset;
        for (int i = 0; i < cnt; i++) {
            int64_t key = (int64_t)row * width + cols[i];
            HT[offset + cols[i]] = row;                     /* phf.c:211 */
            val[offset + cols[i]] = ary[key / CHAR_SET][key % CHAR_SET];   /* phf.c:216 */
        }
        if (!exact) while (first_free < htcap && HT[first_free] != -1) first_free++;
    }
    int HTSize = 0;                                         /* phf.c:232-236 */
    for (int i = MaxOffset; i < MaxOffset + width && i < htcap; i++)
        if (HT[i] >= 0 || val[i] >= 0) HTSize = i + 1;
    for (int i = 0; i < maxrow_bound; i++) free(Row[i].cols);
    free(Row);
    m->r[c] = r; m->HT[c] = HT; m->val[c] = val;
    m->HTSize[c] = HTSize; m->MaxRow[c] = MaxRow; m->NumKeys[c] = NumKeys;
    m->MaxKey[c] = MaxKey; m->MaxOffset[c] = MaxOffset;
    return 0;
}

int orc_ffdm(orc_model *m, int width, int exact) {
    int P = m->P;
    if (!m->r) {
        m->r = (int **)calloc((size_t)P, sizeof(int *));
        m->HT = (int **)calloc((size_t)P, sizeof(int *));
        m->val = (int **)calloc((size_t)P, sizeof(int *));
        m->HTSize = (int *)calloc((size_t)P, sizeof(int));
        m->MaxRow = (int *)calloc((size_t)P, sizeof(int));
        m->NumKeys = (int *)calloc((size_t)P, sizeof(int));
        m->MaxKey = (int *)calloc((size_t)P, sizeof(int));
        m->MaxOffset = (int *)calloc((size_t)P, sizeof(int));
    }
    for (int c = 0; c < P; c++) {
        free(m->r[c]); free(m->HT[c]); free(m->val[c]);
        m->r[c] = m->HT[c] = m->val[c] = NULL;
    }
    m->width = width;
    for (int c = 0; c < P; c++) {
        int rc = ffdm_one(m, c, width, exact);
        if (rc) return rc;
    }
    return 0;
}

int orc_phf_stat(const orc_model *m, int c, int what) {
    if (!m->NumKeys) return -1;   /* orc_ffdm not run */
    switch (what) {
        case 0: return m->NumKeys[c];
        case 1: return m->MaxKey[c];
        case 2: return m->MaxOffset[c];
        case 3: return m->MaxRow[c];     /* "r table size" */
        case 4: return m->HTSize[c];     /* "Hash table size" */
        default: return -1;
    }
}
const int *orc_phf_r(const orc_model *m, int c) { return m->r[c]; }
const int *orc_phf_HT(const orc_model *m, int c) { return m->HT[c]; }
const int *orc_phf_val(const orc_model *m, int c) { return m->val[c]; }

/* The device lookup (master_kernel.cu:52-63) as a function. */
int orc_phf_lookup(const orc_model *m, int c, int state, int ch) {
    int wbit;
    for (wbit = 0; (m->width >> wbit) != 1; wbit++) ;       /* master_kernel.cu:397-398 */
    int key = (state << 8) + ch;
    int row = key >> wbit;
    int col = key & ((1 << wbit) - 1);
    int index = m->r[c][row] + col;
    if (index < 0 || index >= m->HTSize[c]) return -1;
    if (m->HT[c][index] == row) return m->val[c][index];
    return -1;
}

/* ------------------------------------------------------------------ */
static void matches_push(orc_matches *o, int64_t pos, int32_t id) {
    if (o->n == o->cap) {
        o->cap = o->cap ? o->cap * 2 : 4096;
        o->pos = (int64_t *)realloc(o->pos, (size_t)o->cap * sizeof(int64_t));
        o->id = (int32_t *)realloc(o->id, (size_t)o->cap * sizeof(int32_t));
    }
    o->pos[o->n] = pos; o->id[o->n] = id; o->n++;
}
orc_matches *orc_matches_new(void) { return (orc_matches *)calloc(1, sizeof(orc_matches)); }
void orc_matches_free(orc_matches *o) { if (o) { free(o->pos); free(o->id); free(o); } }
int64_t orc_matches_count(const orc_matches *o) { return o->n; }
const int64_t *orc_matches_pos(const orc_matches *o) { return o->pos; }
const int32_t *orc_matches_id(const orc_matches *o) { return o->id; }

/* One thread block of TraceTable_kernel (master_kernel.cu:92-180) for chunk c.
 * d_in is the device input buffer (num_blocks*4096+512 bytes, bytes >= N are
 * whatever cudaMalloc left there; the oracle zero-fills them), dense is the
 * per-chunk slot array with max_pat_len slots per position. */
static void tile_walk(const orc_model *m, int c, const unsigned char *d_in, int input_size,
                      int gbid, int num_blocks, int boundary, int wbit, unsigned int *dense, int max_pat_len) {
    const int num_final = m->n_final[c];
    const int HTSize = m->HTSize[c];
    const int *r = m->r[c], *HT = m->HT[c], *val = m->val[c];
    const int *s0 = m->pfac[c][num_final + 1];              /* main.cc:200 */
    unsigned char s_in[REF_PAGE_SIZE_C + REF_EXTRA_BYTES];
    memcpy(s_in, d_in + (size_t)gbid * REF_PAGE_SIZE_C, sizeof s_in);   /* :127-135 */
    int bdy = (gbid == num_blocks - 1) ? boundary : REF_PAGE_SIZE_C + REF_EXTRA_BYTES;   /* :141-144 */
    for (int j = 0; j < REF_PAGE_SIZE_C / REF_BLOCK_SIZE; j++) {
        for (int tid = 0; tid < REF_BLOCK_SIZE; tid++) {
            unsigned int *match = dense + ((size_t)gbid * REF_PAGE_SIZE_C + tid + (size_t)j * REF_BLOCK_SIZE) * max_pat_len;
            int pos = tid + j * REF_BLOCK_SIZE;             /* SUBSEG_MATCH :38 */
            int ch = s_in[pos];
            if (pos < input_size) {                         /* :40 (tile-local vs global, as in the reference) */
                int state = s0[ch];
                int matchi = 0;
                if (state >= 0) {
                    if (state < num_final) match[matchi++] = (unsigned)state;
                    pos += 1;
                    for (;;) {
                        if (pos >= bdy) break;
                        ch = s_in[pos];
                        int key = (state << 8) + ch;
                        int row = key >> wbit;
                        int col = key & ((1 << wbit) - 1);
                        int index = r[row] + col;
                        if (index < 0 || index >= HTSize) state = -1;
                        else if (HT[index] == row) state = val[index];
                        else state = -1;
                        if (state == -1) break;
                        if (state < num_final) match[matchi++] = (unsigned)state;
                        pos += 1;
                    }
                }
            }
        }
    }
}

/* GPU_TraceTable geometry (master_kernel.cu:330-346) + one tile_walk per block
 * per chunk, then the host merge (main.cc:304-324).  Requires orc_ffdm first.
 * N = number of input bytes scanned (the CLI passes filesize-1, main.cc:138). */
int orc_scan_reference(orc_model *m, const unsigned char *input, int64_t N64, orc_matches *out) {
    if (!m->r) { snprintf(m->err, sizeof m->err, "orc_ffdm not run"); return -1; }
    if (N64 <= 0) return 0;
    if (N64 * (int64_t)m->max_len >= ((int64_t)1 << 32) || N64 >= ((int64_t)1 << 31)) {
        snprintf(m->err, sizeof m->err, "outside the reference's 32-bit domain"); return -2;
    }
    int N = (int)N64;
    int wbit;
    for (wbit = 0; (m->width >> wbit) != 1; wbit++) ;
    int num_blocks = (N + REF_PAGE_SIZE_C - 1) / REF_PAGE_SIZE_C;
    int boundary = N - (num_blocks - 1) * REF_PAGE_SIZE_C;
    size_t dsz = (size_t)num_blocks * REF_PAGE_SIZE_C + REF_EXTRA_BYTES;   /* mk.cu:217 */
    unsigned char *d_in = (unsigned char *)calloc(dsz, 1);
    memcpy(d_in, input, (size_t)N);                                       /* mk.cu:359 */
    int L = m->max_len;
    size_t agg_n = (size_t)N * L;
    int *agg = (int *)malloc(agg_n * sizeof(int));
    memset(agg, 0xFF, agg_n * sizeof(int));                                /* main.cc:305 */
    for (int c = 0; c < m->P; c++) {
        int Lc = m->max_len_arr[c];
        if (Lc == 0) continue;
        /* the kernel may store past N*Lc for offsets >= N of the last tile; the
         * reference's D2H copies only N*Lc ints (mk.cu:428) -- allocate the overrun */
        size_t dn = ((size_t)num_blocks * REF_PAGE_SIZE_C + 1) * Lc;
        unsigned int *dense = (unsigned int *)malloc(dn * sizeof(unsigned int));
        memset(dense, 0xFF, dn * sizeof(unsigned int));                    /* mk.cu:236 */
        for (int b = 0; b < num_blocks; b++)
            tile_walk(m, c, d_in, N, b, num_blocks, boundary, wbit, dense, Lc);
        for (int i = 0; i < N; i++) {                                      /* main.cc:307-321 */
            size_t k = (size_t)i * L;
            while (k < agg_n && agg[k] != -1) k++;
            for (int j = 0; j < Lc; j++) {
                unsigned int s = dense[(size_t)i * Lc + j];
                if (s != 0xFFFFFFFFu) { if (k < agg_n) agg[k++] = m->idmap[c][s]; }
                else break;
            }
        }
        free(dense);
    }
    for (int i = 0; i < N; i++)                                            /* main.cc:341-349 */
        for (int j = 0; j < L; j++) {
            int v = agg[(size_t)i * L + j];
            if (v != -1) matches_push(out, i, v); else break;
        }
    free(agg); free(d_in);
    return 0;
}

/* Spec semantics (SURVEY.md 8a "parity domain"): for every start offset i,
 * for chunk c ascending, walk chunk c's dense trie from the root over
 * input[i..N); every final state reached reports (i, id).  No tiles, no PHF,
 * 64-bit positions.  Equal to orc_scan_reference inside the parity domain. */
int orc_scan_spec(const orc_model *m, const unsigned char *input, int64_t N, orc_matches *out) {
    for (int64_t i = 0; i < N; i++) {
        for (int c = 0; c < m->P; c++) {
            const int nf = m->n_final[c];
            int **T = m->pfac[c];
            int state = nf + 1;
            for (int64_t p = i; p < N; p++) {
                state = T[state][input[p]];
                if (state < 0) break;
                if (state < nf) matches_push(out, i, m->idmap[c][state]);
            }
        }
    }
    return 0;
}

/* Text emitter (main.cc:335-350): "At position %4d, match pattern %d\n".
 * Positions are int in the reference; %4lld prints identically wherever the
 * reference is defined.  Returns bytes written or -1. */
int64_t orc_emit(const orc_matches *o, const char *path) {
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    int64_t bytes = 0;
    for (int64_t k = 0; k < o->n; k++)
        bytes += fprintf(f, "At position %4lld, match pattern %d\n", (long long)o->pos[k], o->id[k]);
    fclose(f);
    return bytes;
}

void orc_free(orc_model *m) {
    if (!m) return;
    for (int c = 0; c < m->P; c++) {
        if (m->pfac && m->pfac[c]) {
            for (int s = 0; s < m->state_num[c]; s++) free(m->pfac[c][s]);
            free(m->pfac[c]);
        }
        if (m->idmap) free(m->idmap[c]);
        if (m->r) { free(m->r[c]); free(m->HT[c]); free(m->val[c]); }
    }
    for (int i = 0; i < m->n_pat; i++) free(m->pats[i].pat);
    free(m->pats); free(m->n_final); free(m->state_num); free(m->max_len_arr);
    free(m->pfac); free(m->idmap); free(m->r); free(m->HT); free(m->val);
    free(m->HTSize); free(m->MaxRow); free(m->NumKeys); free(m->MaxKey); free(m->MaxOffset);
    free(m);
}

/* ------------------------------------------------------------------ */
#ifdef ORC_MAIN
/* CPU-only gphf: same argv as the reference (main.cc:93-96), writes
 * GPU_match_result.txt in the CWD.  BASELINE config 1 ("CPU plumbing"). */
int main(int argc, char **argv) {
    if (argc != 5) {
        fprintf(stderr, "usage: %s <pattern file name> <streamnum> <PHF width> <input file name>\n", argv[0]);
        return 255;
    }
    orc_model *m = orc_build(argv[1], atoi(argv[2]), 4);
    if (orc_error(m)) { fprintf(stderr, "%s\n", orc_error(m)); return 1; }
    if (orc_ffdm(m, atoi(argv[3]), 1)) { fprintf(stderr, "%s\n", orc_error(m)); return 1; }
    FILE *f = fopen(argv[4], "rb");
    if (!f) { perror("Open input file failed."); return 1; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f) - 1;                                   /* main.cc:138 */
    rewind(f);
    unsigned char *buf = (unsigned char *)malloc(n > 0 ? (size_t)n : 1);
    if (n > 0 && fread(buf, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);
    orc_matches *o = orc_matches_new();
    if (orc_scan_reference(m, buf, n, o)) { fprintf(stderr, "%s\n", orc_error(m)); return 1; }
    if (orc_emit(o, "GPU_match_result.txt") < 0) { perror("Open output file failed.\n"); return 1; }
    printf("input size is %ld char\n%lld matches\n", n, (long long)orc_matches_count(o));
    return 0;
}
#endif

/*
 * oracle/ac_serial.c -- TEST INFRASTRUCTURE ONLY (CPU baseline + cross-check).
 *
 * Classic serial Aho-Corasick: goto trie + failure links compiled into a full
 * next[state][256] DFA, ONE table lookup per input byte, output lists chained
 * through dictionary-suffix links.  The reference has no CPU matcher at all
 * (SURVEY.md section 0.10); this is the "serial Aho-Corasick CPU baseline" that
 * BASELINE.json's north_star asks to be timed next to the GPU numbers
 * (bench.py cpu_baseline, kind "port"), and an independent algorithm that the
 * PFAC restatement is cross-checked against as a SET of (start, id) pairs
 * (tests/test_oracle.py).
 *
 * Pattern ids and the duplicate rule follow the reference: id = 1-based line
 * number (create_table_reorder.c:81,100); among identical strings the one that
 * sorts last (= later line) wins (create_table_reorder.c:366).
 * A hit ending at byte e for a pattern of length L is reported at
 * start = e - L + 1, which is the position key of GPU_match_result.txt
 * (master_kernel.cu:37-74 reports by START offset).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pfac_oracle.h"

typedef struct {
    int n_states;
    int32_t *next;      /* [n_states][256] full DFA */
    int32_t *out_id;    /* pattern id ending exactly at this state, or 0 */
    int32_t *out_len;   /* its length */
    int32_t *dict;      /* dictionary-suffix link: nearest proper suffix state with out_id != 0, or 0 */
} ac_model;

ac_model *ac_build(const orc_model *m) {
    /* upper bound on trie nodes: 1 + total pattern bytes */
    size_t total = 1;
    for (int i = 0; i < m->n_pat; i++) total += (size_t)m->pats[i].len;
    ac_model *a = (ac_model *)calloc(1, sizeof *a);
    int32_t *next = (int32_t *)malloc(total * 256 * sizeof(int32_t));
    memset(next, 0xFF, 256 * sizeof(int32_t));
    int32_t *out_id = (int32_t *)calloc(total, sizeof(int32_t));
    int32_t *out_len = (int32_t *)calloc(total, sizeof(int32_t));
    int n = 1;
    for (int i = 0; i < m->n_pat; i++) {        /* sorted order: later duplicate overwrites */
        int s = 0;
        for (int j = 0; j < m->pats[i].len; j++) {
            int ch = m->pats[i].pat[j];
            if (next[(size_t)s * 256 + ch] < 0) {
                memset(next + (size_t)n * 256, 0xFF, 256 * sizeof(int32_t));
                next[(size_t)s * 256 + ch] = n++;
            }
            s = next[(size_t)s * 256 + ch];
        }
        out_id[s] = m->pats[i].id;
        out_len[s] = m->pats[i].len;
    }
    int32_t *fail = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    int32_t *dict = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    int32_t *queue = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    int qh = 0, qt = 0;
    for (int ch = 0; ch < 256; ch++) {
        int t = next[ch];
        if (t < 0) next[ch] = 0; else { fail[t] = 0; queue[qt++] = t; }
    }
    while (qh < qt) {
        int s = queue[qh++];
        int f = fail[s];
        dict[s] = out_id[f] ? f : dict[f];
        for (int ch = 0; ch < 256; ch++) {
            int t = next[(size_t)s * 256 + ch];
            if (t < 0) next[(size_t)s * 256 + ch] = next[(size_t)f * 256 + ch];
            else { fail[t] = next[(size_t)f * 256 + ch]; queue[qt++] = t; }
        }
    }
    free(fail); free(queue);
    a->n_states = n;
    a->next = (int32_t *)realloc(next, (size_t)n * 256 * sizeof(int32_t));
    a->out_id = out_id; a->out_len = out_len; a->dict = dict;
    return a;
}

int ac_num_states(const ac_model *a) { return a->n_states; }

/* order-independent 64-bit checksum of a match set; the same formula is used on
 * GPU records (phfpfac_amd.records_checksum) for full-size parity checks */
static inline uint64_t match_hash(uint64_t pos, uint32_t id) {
    uint64_t x = (pos + 1) * 0x9E3779B97F4A7C15ull ^ ((uint64_t)id * 0xC2B2AE3D27D4EB4Full);
    x ^= x >> 29;
    return x * 0xBF58476D1CE4E5B9ull;
}
uint64_t orc_match_hash(uint64_t pos, uint32_t id) { return match_hash(pos, id); }

/* The timed leg: one DFA step per byte, every hit counted and folded into the
 * checksum (nothing stored).  Returns the number of matches. */
int64_t ac_scan_count(const ac_model *a, const unsigned char *in, int64_t N, uint64_t *checksum) {
    const int32_t *next = a->next, *out_id = a->out_id, *out_len = a->out_len, *dict = a->dict;
    int s = 0;
    int64_t cnt = 0;
    uint64_t sum = 0;
    for (int64_t e = 0; e < N; e++) {
        s = next[(size_t)s * 256 + in[e]];
        if (out_id[s] | dict[s]) {
            for (int t = out_id[s] ? s : dict[s]; t; t = dict[t]) {
                sum += match_hash((uint64_t)(e - out_len[t] + 1), (uint32_t)out_id[t]);
                cnt++;
            }
        }
    }
    if (checksum) *checksum = sum;
    return cnt;
}

/* The same count over a SLICE of a larger buffer, so that a whole 1-4 GiB shard can be checked on several host threads:
 * the automaton starts cold at in[0]; only hits that END at byte >= first_end and START before start_limit are counted,
 * hashed at position base + start.  With in = shard + a - (L-1) (L = longest pattern), first_end = L-1 (clamped at
 * the shard's head) and N covering up to byte b, the slices' (count, checksum) sums equal one serial pass over the
 * shard: a hit ending at e >= a starts at >= a - (L-1), which is where the cold automaton started.  start_limit is
 * the GPU scan's n_owned (matches are keyed by START offset, master_kernel.cu:37-74; the bytes behind n_owned are
 * read-only halo). */
int64_t ac_scan_count_range(const ac_model *a, const unsigned char *in, int64_t N, int64_t first_end, int64_t start_limit,
                            uint64_t base, uint64_t *checksum) {
    const int32_t *next = a->next, *out_id = a->out_id, *out_len = a->out_len, *dict = a->dict;
    int s = 0;
    int64_t cnt = 0;
    uint64_t sum = 0;
    for (int64_t e = 0; e < N; e++) {
        s = next[(size_t)s * 256 + in[e]];
        if ((out_id[s] | dict[s]) && e >= first_end) {
            for (int t = out_id[s] ? s : dict[s]; t; t = dict[t]) {
                const int64_t st = e - out_len[t] + 1;
                if (st >= start_limit) continue;
                sum += match_hash(base + (uint64_t)st, (uint32_t)out_id[t]);
                cnt++;
            }
        }
    }
    if (checksum) *checksum = sum;
    return cnt;
}

/* Collect (start, id); order is end-major (AC order), callers compare as a set. */
int64_t ac_scan_collect(const ac_model *a, const unsigned char *in, int64_t N, orc_matches *out) {
    int s = 0;
    for (int64_t e = 0; e < N; e++) {
        s = a->next[(size_t)s * 256 + in[e]];
        for (int t = a->out_id[s] ? s : a->dict[s]; t; t = a->dict[t]) {
            if (out->n == out->cap) {
                out->cap = out->cap ? out->cap * 2 : 4096;
                out->pos = (int64_t *)realloc(out->pos, (size_t)out->cap * sizeof(int64_t));
                out->id = (int32_t *)realloc(out->id, (size_t)out->cap * sizeof(int32_t));
            }
            out->pos[out->n] = e - a->out_len[t] + 1;
            out->id[out->n] = a->out_id[t];
            out->n++;
        }
    }
    return out->n;
}

/* checksum of an orc_matches list with the same formula */
uint64_t orc_matches_checksum(const orc_matches *o) {
    uint64_t sum = 0;
    for (int64_t k = 0; k < o->n; k++) sum += match_hash((uint64_t)o->pos[k], (uint32_t)o->id[k]);
    return sum;
}

void ac_free(ac_model *a) {
    if (!a) return;
    free(a->next); free(a->out_id); free(a->out_len); free(a->dict); free(a);
}

/*
 * seam_main.cc -- a driver with the SHAPE of the reference's main() (regex_GPU_PHF/main.cc:45-352) that reaches the
 * GPU only through the reference's own three seam functions (include/pfac_seam.h, libpfac_seam.so):
 *
 *     gphf_seam <pattern file name> <streamnum> <PHF width> <input file name>      -> GPU_match_result.txt
 *
 * It keeps what main.cc does around the seam -- P = 4 x streamnum pattern chunks (create_table_reorder.c:207,217),
 * one thread_data per chunk filled as main.cc:193-204 fills it, every chunk scans the WHOLE input into a dense
 * input_size x max_pat_len array, then the position-major merge (main.cc:304-324) and the fprintf loop
 * (main.cc:341-349) -- so tests/test_gpu_parity.py can check that a reference-shaped program linked against the seam
 * writes the golden GPU_match_result.txt.  The table builder is this repository's (pfac_table_build_file_part = the
 * reference's chunk tables, tests/test_partition.py); the product CLI is gphf.c, not this.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include <vector>

#include "pfac.h"
#include "pfac_seam.h"

int main(int argc, char *argv[]) {
    if (argc != 5) { fprintf(stderr, "usage: %s <pattern file name> <streamnum> <PHF width> <input file name>\n", argv[0]); return 1; }
    const int streamnum = atoi(argv[2]), width = atoi(argv[3]);
    int GPU_N = 0;
    if (streamnum < 1 || pfac_device_count(&GPU_N) || GPU_N < 1) { fprintf(stderr, "no GPU / bad streamnum\n"); return 1; }
    const int P = 4 * streamnum;                                 // chunks of the sorted pattern list
    std::vector<pfac_table *> tab(P, nullptr);
    char err[256];
    int max_pat_len = 0;
    for (int p = 0; p < P; p++) {
        if (pfac_table_build_file_part(argv[1], width, p, P, &tab[p], err, sizeof err)) { fprintf(stderr, "table: %s\n", err); return 1; }
        if (tab[p]->max_pat_len > max_pat_len) max_pat_len = tab[p]->max_pat_len;
    }
    FILE *fpin = fopen(argv[4], "rb");
    if (!fpin) { perror("Open input file failed."); return 1; }
    fseek(fpin, 0, SEEK_END);
    const long fsize = ftell(fpin);
    const int input_size = fsize > 0 ? (int)(fsize - 1) : 0;     // main.cc:137-138: the last byte is dropped
    rewind(fpin);
    std::vector<unsigned char> input_string((size_t)input_size + 1);
    if (fread(input_string.data(), 1, (size_t)input_size, fpin) != (size_t)input_size) { fprintf(stderr, "short read\n"); return 1; }
    fclose(fpin);

    std::vector<std::vector<unsigned int>> match_result(P);
    for (int p = 0; p < P; p++) {                                // one chunk after the other (the reference: one OpenMP thread each)
        if (hipSetDevice(p % GPU_N) != hipSuccess) { fprintf(stderr, "Set device %d error\n", p % GPU_N); return 1; }
        match_result[p].resize((size_t)input_size * (size_t)max_pat_len + 1);
        thread_data d;                                           // main.cc:193-204
        d.input_string = input_string.data();
        d.input_size = input_size;
        d.state_num = tab[p]->state_num;
        d.final_state_num = tab[p]->num_final;
        d.match_result = match_result[p].data();
        d.HTSize = tab[p]->ht_size;
        d.width = width;
        d.s0Table = tab[p]->s0;
        d.max_pat_len = max_pat_len;
        d.r = tab[p]->r;
        d.HT = tab[p]->HT;
        d.val = tab[p]->val;
        unsigned char *d_input_string; int *d_r, *d_hash_table, *d_val_table, *d_s0Table; unsigned int *d_match_result;
        GPU_Malloc_Memory(d, &d_input_string, &d_r, &d_hash_table, &d_match_result, &d_val_table, &d_s0Table);
        GPU_TraceTable(d, (hipStream_t)nullptr, d_input_string, d_r, d_hash_table, d_match_result, d_val_table, d_s0Table);
        GPU_Free_memory(&d_input_string, &d_r, &d_hash_table, &d_match_result, &d_val_table, &d_s0Table);
    }
    // merge, position-major: chunk p's ids go behind what position i already holds (main.cc:304-324) ...
    std::vector<int> all((size_t)input_size * (size_t)max_pat_len + 1, -1);
    for (int i = 0; i < input_size; i++) {
        int k = 0;
        for (int p = 0; p < P; p++)
            for (int j = 0; j < max_pat_len; j++) {
                const unsigned int st = match_result[p][(size_t)i * max_pat_len + j];
                if (st == 0xFFFFFFFFu) break;
                if (k < max_pat_len) all[(size_t)i * max_pat_len + k++] = tab[p]->idmap[st];
            }
    }
    // ... and the text (main.cc:335-350)
    FILE *fpout = fopen("GPU_match_result.txt", "w");
    if (!fpout) { perror("Open output file failed.\n"); return 1; }
    for (int i = 0; i < input_size; i++)
        for (int j = 0; j < max_pat_len; j++) {
            const int id = all[(size_t)i * max_pat_len + j];
            if (id != -1) fprintf(fpout, "At position %4d, match pattern %d\n", i, id);
        }
    fclose(fpout);
    for (int p = 0; p < P; p++) pfac_table_free(tab[p]);
    return 0;
}

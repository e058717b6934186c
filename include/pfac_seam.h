/*
 * include/pfac_seam.h -- the reference's host<->device seam, by its own names.
 *
 * regex_GPU_PHF/main.cc:19-37 declares struct thread_data and three free functions with C++ linkage
 * (defined in master_kernel.cu:188-257, 277-455, 457-524).  libpfac_seam.so (phfpfac_amd/csrc/pfac_seam.cc) defines
 * the same three functions, same parameter lists (the cudaStream_t, which the reference never uses --
 * master_kernel.cu:277,406,425 -- is accepted both as a void* and as a hipStream_t: two overloads), on top of the
 * C-ABI of pfac.h: a maintainer keeps main.cc, drops master_kernel.cu
 * and links this library instead.  C++ only (struct by value, mangled names), exactly like the reference.
 *
 * Behaviour kept: the device is the caller's CURRENT device (main.cc calls cudaSetDevice before each of them,
 * main.cc:183,229,258); every failure prints and exit(1)s (master_kernel.cu:240-244 and all the others); the result is
 * the reference's dense layout -- max_pat_len slots per input position, 0xFFFFFFFF = empty (master_kernel.cu:104-115,
 * 236) -- so main.cc's merge and emit loops (main.cc:304-350) run unchanged.
 * What the six d_* pointers hold is this library's business (one opaque handle, copied into all six); treat them as
 * the reference does: pass them back, never dereference them.
 */
#ifndef PFAC_SEAM_H
#define PFAC_SEAM_H

struct thread_data {            /* main.cc:19-32, field for field */
    unsigned char *input_string;
    int input_size;
    int state_num;
    int final_state_num;
    unsigned int *match_result;
    int HTSize;
    int width;
    int *s0Table;
    int max_pat_len;
    int *r;
    int *HT;
    int *val;
};

typedef void *pfac_seam_stream;  /* where main.cc passes a cudaStream_t */
struct ihipStream_t;             /* hipStream_t == ihipStream_t * (hip/hip_runtime_api.h): what a hipified main.cc:36 names */

int GPU_Malloc_Memory(thread_data dataset, unsigned char **d_input_string, int **d_r, int **d_hash_table,
                      unsigned int **d_match_result, int **d_val_table, int **d_s0Table);             /* main.cc:35 */
int GPU_TraceTable(thread_data dataset, pfac_seam_stream stream, unsigned char *d_input_string, int *d_r,
                   int *d_hash_table, unsigned int *d_match_result, int *d_val_table, int *d_s0Table);  /* main.cc:36 */
/* The same function for a main.cc whose cudaStream_t became hipStream_t (its prototype at main.cc:36 then mangles to
 * ...P12ihipStream_t...): such a main.cc links against this library with NO typedef and no edit of the prototype. */
int GPU_TraceTable(thread_data dataset, ihipStream_t *stream, unsigned char *d_input_string, int *d_r,
                   int *d_hash_table, unsigned int *d_match_result, int *d_val_table, int *d_s0Table);
int GPU_Free_memory(unsigned char **d_input_string, int **d_r, int **d_hash_table, unsigned int **d_match_result,
                    int **d_val_table, int **d_s0Table);                                               /* main.cc:37 */

#endif /* PFAC_SEAM_H */

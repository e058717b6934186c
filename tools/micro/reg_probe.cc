// GPU box diagnostic: cost of making page-cache pages DMA-able (populate / register), per 32 MiB piece.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static char *g_m; static size_t P = 32 << 20;
struct job { size_t lo, hi; };
static void *pop(void *a) { job *j = (job *)a; for (size_t o = j->lo; o < j->hi; o += P) madvise(g_m + o, P, MADV_POPULATE_READ); return 0; }
int main(int argc, char **argv) {
    int fd = open(argv[1], O_RDONLY); struct stat sb; fstat(fd, &sb);
    size_t len = (size_t)sb.st_size & ~(P - 1);
    char *d; CK(hipMalloc((void **)&d, P)); hipStream_t st; CK(hipStreamCreate(&st));
    for (int mode = 0; mode < 4; mode++) {
        g_m = (char *)mmap(0, len, PROT_READ, MAP_SHARED, fd, 0);
        size_t n = len < ((size_t)2 << 30) ? len : ((size_t)2 << 30);
        double t0 = now(), tp = 0, tr = 0, tc = 0, tu = 0;
        if (mode == 1) { double a = now(); for (size_t o = 0; o < n; o += P) madvise(g_m + o, P, MADV_POPULATE_READ); tp = now() - a; }
        if (mode == 2) { double a = now(); pthread_t th[8]; job jb[8]; for (int i = 0; i < 8; i++) { jb[i] = {n / 8 * i, n / 8 * (i + 1)}; pthread_create(&th[i], 0, pop, &jb[i]); } for (int i = 0; i < 8; i++) pthread_join(th[i], 0); tp = now() - a; }
        if (mode == 3) { double a = now(); CK(hipHostRegister(g_m, n, hipHostRegisterPortable)); tr = now() - a; }
        for (size_t o = 0; o < n; o += P) {
            double a = now();
            if (mode != 3) CK(hipHostRegister(g_m + o, P, hipHostRegisterPortable));
            double b = now();
            CK(hipMemcpyAsync(d, g_m + o, P, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
            double c = now();
            if (mode != 3) CK(hipHostUnregister(g_m + o));
            double e = now();
            tr += b - a; tc += c - b; tu += e - c;
        }
        if (mode == 3) { double a = now(); CK(hipHostUnregister(g_m)); tu = now() - a; }
        const char *nm[] = {"cold (no populate), register per 32 MiB", "populate 1 thread, then register per piece", "populate 8 threads, then register per piece", "ONE register of the whole range (cold), copies in pieces"};
        printf("%-62s: populate %.1f GB/s, register %.1f GB/s, copy (sync each) %.1f GB/s, unregister %.1f GB/s, all %.1f GB/s\n", nm[mode],
               tp ? n / tp / 1e9 : 0, n / tr / 1e9, n / tc / 1e9, n / tu / 1e9, n / (now() - t0) / 1e9);
        munmap(g_m, len);
    }
    return 0;
}

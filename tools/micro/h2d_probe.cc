// GPU box diagnostic (not part of the product): what the host link gives this process.
//   h2d_probe <file>     pinned H2D rate (one / two streams), the same with 12 threads pread()ing next to it, NUMA facts,
//                        and whether a page-cache mapping of <file> can be registered for DMA in place (zero-copy ingest)
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
#include <atomic>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static std::atomic<int> g_stop{0};
static int g_fd; static size_t g_fsize;
static void *reader(void *a) {
    size_t id = (size_t)a; char *buf; CK(hipHostMalloc((void **)&buf, 32 << 20, hipHostMallocPortable));
    size_t off = id * (256ull << 20), n = 0;
    while (!g_stop) { if (off + (32 << 20) > g_fsize) off = 0; ssize_t r = pread(g_fd, buf, 32 << 20, off); if (r <= 0) break; off += r; n += r; }
    return (void *)n;
}
int main(int argc, char **argv) {
    const size_t CH = 32 << 20; const int NB = 4, REP = 128;
    char *h[NB]; char *d[NB]; hipStream_t st[NB];
    for (int i = 0; i < NB; i++) { CK(hipHostMalloc((void **)&h[i], CH, hipHostMallocPortable)); memset(h[i], i + 1, CH); CK(hipMalloc((void **)&d[i], CH)); CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); }
    for (int ns = 1; ns <= 4; ns *= 2) {
        for (int w = 0; w < 2; w++) {
            double t0 = now();
            for (int r = 0; r < REP; r++) CK(hipMemcpyAsync(d[r % NB], h[r % NB], CH, hipMemcpyHostToDevice, st[r % ns]));
            CK(hipDeviceSynchronize());
            double dt = now() - t0;
            if (w) printf("H2D pinned, %d stream(s): %.1f GB/s\n", ns, REP * (double)CH / dt / 1e9);
        }
    }
    if (argc > 1) {
        g_fd = open(argv[1], O_RDONLY); struct stat sb; fstat(g_fd, &sb); g_fsize = sb.st_size;
        for (int nt = 4; nt <= 16; nt += 4) {
            g_stop = 0; pthread_t th[16];
            for (size_t i = 0; i < (size_t)nt; i++) pthread_create(&th[i], 0, reader, (void *)i);
            double t0 = now();
            for (int r = 0; r < 4 * REP; r++) CK(hipMemcpyAsync(d[r % NB], h[r % NB], CH, hipMemcpyHostToDevice, st[r % 2]));
            CK(hipDeviceSynchronize());
            double dt = now() - t0; g_stop = 1; size_t tot = 0;
            for (int i = 0; i < nt; i++) { void *n; pthread_join(th[i], &n); tot += (size_t)n; }
            printf("with %2d pread threads: H2D %.1f GB/s, preads %.1f GB/s\n", nt, 4 * REP * (double)CH / dt / 1e9, tot / dt / 1e9);
        }
        // zero-copy: register a page-cache mapping
        size_t len = g_fsize < (1ull << 30) ? g_fsize & ~((size_t)(2 << 20) - 1) : (1ull << 30);
        void *m = mmap(0, len, PROT_READ, MAP_SHARED | MAP_POPULATE, g_fd, 0);
        printf("mmap %zu bytes: %p\n", len, m);
        for (size_t piece : {(size_t)32 << 20, (size_t)256 << 20}) {
            double t0 = now(); hipError_t e = hipSuccess; size_t done = 0;
            for (size_t o = 0; o + piece <= len && e == hipSuccess; o += piece) { e = hipHostRegister((char *)m + o, piece, hipHostRegisterDefault | hipHostRegisterReadOnly); if (e == hipSuccess) done += piece; }
            double dt = now() - t0;
            printf("hipHostRegister(page cache mapping, %zu MiB pieces): %s, %.2f GB/s\n", piece >> 20, hipGetErrorString(e), done / dt / 1e9);
            if (e != hipSuccess) { (void)hipGetLastError(); e = hipHostRegister(m, piece, hipHostRegisterDefault); printf("  without ReadOnly flag: %s\n", hipGetErrorString(e)); if (e == hipSuccess) { done = piece; } }
            if (done) {
                t0 = now();
                for (size_t o = 0; o + CH <= done; o += CH) CK(hipMemcpyAsync(d[(o / CH) % NB], (char *)m + o, CH, hipMemcpyHostToDevice, st[(o / CH) % 2]));
                CK(hipDeviceSynchronize());
                printf("  H2D straight from the registered page cache: %.1f GB/s\n", done / (now() - t0) / 1e9);
                char *chk = (char *)malloc(CH); CK(hipMemcpy(chk, d[0], CH, hipMemcpyDeviceToHost));
                size_t o0 = ((done / CH - 1) / NB * NB) * CH;   // last chunk that landed in d[0]
                printf("  bytes equal: %d\n", memcmp(chk, (char *)m + o0, CH) == 0);
                t0 = now();
                for (size_t o = 0; o < done; o += piece) (void)hipHostUnregister((char *)m + o);
                printf("  unregister: %.2f GB/s\n", done / (now() - t0) / 1e9);
            }
        }
    }
    system("cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\\n' ' '; echo; lscpu | grep -iE 'numa|socket|model name' ; cat /proc/self/status | grep -i allowed_list");
    return 0;
}

// Microbenchmark (diagnostic, not product): what a persistent streaming-read kernel shaped like the PFAC scan can
// pull from HBM on MI355X -- waves per CU, tiles in flight per wave, LDS copy on/off.
//   hipcc --offload-arch=gfx950 -O3 -o ab/membw tools/micro/membw.hip && ab/membw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int DEPTH, bool LDSCOPY, int TILE_LOADS>
__global__ __launch_bounds__(1024) void stream_kernel(const unsigned char *in, unsigned long long n_tiles, unsigned *out, unsigned *ticket, int dynamic) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    constexpr int TB = TILE_LOADS * 1024;
    unsigned char *tile = smem + wave * TB;
    u32x4 w[DEPTH][TILE_LOADS];
    unsigned acc = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * nw;
    unsigned long long t = (unsigned long long)blockIdx.x * nw + wave;
    auto next_tile = [&](unsigned long long cur) -> unsigned long long {
        if (!dynamic) return cur + stride;
        unsigned g = 0;
        if (lane == 0) g = atomicAdd(ticket, 1u);
        return (unsigned long long)__builtin_amdgcn_readfirstlane(g);
    };
    if (dynamic) t = next_tile(0);
    unsigned long long tq[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
        tq[d] = t;
        if (t < n_tiles) {
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(in + t * TB), 0, TB, 0x00020000);
#pragma unroll
            for (int j = 0; j < TILE_LOADS; j++) w[d][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, j * 1024 + lane * 16, 0, 0);
        }
        t = next_tile(t);
    }
    for (;;) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            if (tq[d] >= n_tiles) goto done;
            if (LDSCOPY) {
#pragma unroll
                for (int j = 0; j < TILE_LOADS; j++) *reinterpret_cast<u32x4 *>(tile + j * 1024 + lane * 16) = w[d][j];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < TILE_LOADS / 2; j++) {
                    const u32x4 a = *reinterpret_cast<const u32x4 *>(tile + j * 2048 + lane * 32);
                    const u32x4 b = *reinterpret_cast<const u32x4 *>(tile + j * 2048 + lane * 32 + 16);
                    acc ^= a[0] ^ a[1] ^ a[2] ^ a[3] ^ b[0] ^ b[1] ^ b[2] ^ b[3];
                }
            } else {
#pragma unroll
                for (int j = 0; j < TILE_LOADS; j++) acc ^= w[d][j][0] ^ w[d][j][1] ^ w[d][j][2] ^ w[d][j][3];
            }
            tq[d] = t;
            if (t < n_tiles) {
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(in + t * TB), 0, TB, 0x00020000);
#pragma unroll
                for (int j = 0; j < TILE_LOADS; j++) w[d][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, j * 1024 + lane * 16, 0, 0);
            }
            t = next_tile(t);
        }
    }
done:
    if (acc == 0x12345678u) out[0] = acc;
}

template <int DEPTH, bool LDSCOPY, int TILE_LOADS>
void run(const char *name, const unsigned char *d_in, size_t n, unsigned *d_out, unsigned *d_ticket, int waves, int blocks_per_cu, int dynamic) {
    const int TB = TILE_LOADS * 1024;
    const unsigned long long n_tiles = n / TB;
    auto k = stream_kernel<DEPTH, LDSCOPY, TILE_LOADS>;
    size_t lds = (size_t)waves * TB;
    if (blocks_per_cu == 1 && lds < 82 * 1024) lds = 82 * 1024;
    CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0;
    const int reps = 12;
    for (int i = 0; i < reps + 3; i++) {
        CHECK(hipMemsetAsync(d_ticket, 0, 4, 0));
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(256 * blocks_per_cu), dim3(64 * waves), lds, 0, d_in, n_tiles, d_out, d_ticket, dynamic);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 3) { sum += ms; if (ms < best) best = ms; }
    }
    printf("%-44s waves/blk %2d blk/CU %d dyn %d : avg %.3f ms %.0f GB/s   best %.3f ms %.0f GB/s\n", name, waves, blocks_per_cu, dynamic,
           sum / reps, n / (sum / reps) / 1e6, best, n / best / 1e6);
}

int main() {
    const size_t n = 1ull << 30;
    unsigned char *d_in; unsigned *d_out, *d_ticket;
    CHECK(hipMalloc((void **)&d_in, n + 65536)); CHECK(hipMalloc((void **)&d_out, 64)); CHECK(hipMalloc((void **)&d_ticket, 64));
    CHECK(hipMemset(d_in, 1, n + 65536));
    run<1, false, 4>("depth1 noLDS tile4K", d_in, n, d_out, d_ticket, 16, 1, 0);
    run<2, false, 4>("depth2 noLDS tile4K", d_in, n, d_out, d_ticket, 16, 1, 0);
    run<4, false, 4>("depth4 noLDS tile4K", d_in, n, d_out, d_ticket, 16, 1, 0);
    run<1, true, 4>("depth1 LDScopy tile4K", d_in, n, d_out, d_ticket, 16, 1, 0);
    run<2, true, 4>("depth2 LDScopy tile4K", d_in, n, d_out, d_ticket, 16, 1, 0);
    run<1, true, 4>("depth1 LDScopy tile4K", d_in, n, d_out, d_ticket, 15, 1, 0);
    run<1, true, 4>("depth1 LDScopy tile4K", d_in, n, d_out, d_ticket, 12, 1, 0);
    run<1, true, 4>("depth1 LDScopy tile4K", d_in, n, d_out, d_ticket, 8, 1, 0);
    run<1, true, 4>("depth1 LDScopy tile4K 2blk", d_in, n, d_out, d_ticket, 16, 2, 0);
    run<2, true, 4>("depth2 LDScopy tile4K 2blk", d_in, n, d_out, d_ticket, 16, 2, 0);
    run<1, false, 4>("depth1 noLDS tile4K 2blk", d_in, n, d_out, d_ticket, 16, 2, 0);
    run<1, true, 2>("depth1 LDScopy tile2K 2blk", d_in, n, d_out, d_ticket, 16, 2, 0);
    run<2, true, 2>("depth2 LDScopy tile2K 2blk", d_in, n, d_out, d_ticket, 16, 2, 0);
    run<1, true, 8>("depth1 LDScopy tile8K", d_in, n, d_out, d_ticket, 16, 1, 0);
    run<1, true, 4>("depth1 LDScopy tile4K dynamic", d_in, n, d_out, d_ticket, 16, 1, 1);
    run<2, true, 4>("depth2 LDScopy tile4K dynamic", d_in, n, d_out, d_ticket, 16, 1, 1);
    return 0;
}

// LDS-DMA operand feed rate per CU under the lane -> address maps a GEMM k-tile stage can use (no MFMA, no LDS reads): is the
// ~80 GB/s per CU that bounds the 128 x 128 kernels a property of the L1 path, or of the 8-lanes-per-128-byte-row
// request shape (TCP_PENDING_STALL: the second 64-byte half of a line requested while the first is in flight)?
// Build: hipcc -O3 --offload-arch=gfx950 -o dma_feed dma_feed.hip ; run: ./dma_feed
// The kernel is gemm_a3_kernel's ring without the arithmetic: block = 8 waves, 3 A slots + 2 B slots of 16 KB, A two k-tiles ahead,
// B one; per k-tile a wave issues 2 pieces (1 KB each) per operand, waits for all but its 2 youngest, one barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const char* src, char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

// PAT 0: piece = 8 rows x 128 B (lane -> row l/8, chunk l%8)            [the product kernels]
// PAT 1: piece = 16 rows x 64 B; the two halves of a row go to DIFFERENT waves (pieces 2r, 2r+1)
// PAT 2: piece = 16 rows x 64 B; a wave's two pieces are the two halves of the same 16 rows (consecutive instructions)
// PAT 3: k-tile of 128 elements: piece = 4 rows x 256 B (64 rows per 16 KB stage)
template <int PAT> __device__ __forceinline__ void lane_src(int q, int lane, int& row, int& col) {
    if (PAT == 0) { row = 8 * q + (lane >> 3); col = (lane & 7) * 16; }
    else if (PAT == 1) { row = 16 * (q >> 1) + (lane >> 2); col = (q & 1) * 64 + (lane & 3) * 16; }
    else if (PAT == 2) { row = 16 * (q & 7) + (lane >> 2); col = (q >> 3) * 64 + (lane & 3) * 16; }
    else { row = 4 * q + (lane >> 4); col = (lane & 15) * 16; }
}

template <int PAT, int LDSKB>
__global__ __launch_bounds__(512, 2) void feed_kernel(const char* A, const char* B, long long lda, long long ldb, int tiles_m, int tiles_n, int nk, int* sink) {
    constexpr int SLOT = 16384, KROWB = PAT == 3 ? 256 : 128, ROWS = SLOT / KROWB;
    __shared__ __attribute__((aligned(1024))) char lds[LDSKB * 1024];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int total = tiles_m * tiles_n;
    int ra[2], ca[2];
    lane_src<PAT>(w, lane, ra[0], ca[0]);
    lane_src<PAT>(w + 8, lane, ra[1], ca[1]);
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        // 8 m-tiles x all n-tiles per group, as the product kernels order their tiles
        const int tpg = 8 * tiles_n, grp = tile / tpg, fm = grp * 8, gsz = min(8, tiles_m - fm), tin = tile - grp * tpg;
        const int mt = fm + tin % gsz, nt = tin / gsz;
        const char* a0 = A + (long long)mt * ROWS * lda;
        const char* b0 = B + (long long)nt * ROWS * ldb;
        auto issue_a = [&](int kt, int slot) {
            const int k = (kt < nk ? kt : nk - 1) * KROWB;
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(a0 + ra[j] * lda + k + ca[j], lds + slot * SLOT + (w + 8 * j) * 1024);
        };
        auto issue_b = [&](int kt, int slot) {
            const int k = (kt < nk ? kt : nk - 1) * KROWB;
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(b0 + ra[j] * ldb + k + ca[j], lds + (3 + slot) * SLOT + (w + 8 * j) * 1024);
        };
        issue_a(0, 0); issue_b(0, 0); issue_a(1, 1);
        int sa = 0, sb = 0;
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            issue_b(kt + 1, sb ^ 1);
            issue_a(kt + 2, sa == 0 ? 2 : sa - 1);
            sa = sa == 2 ? 0 : sa + 1;
            sb ^= 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) sink[0] = lds[lane];
}

template <int PAT, int LDSKB>
double run(const char* A, const char* B, long long lda, long long ldb, int tiles_m, int tiles_n, int nk, int grid, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int r = 0; r < reps + 2; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((feed_kernel<PAT, LDSKB>), dim3(grid), dim3(512), 0, 0, A, B, lda, ldb, tiles_m, tiles_n, nk, (int*)nullptr);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2] * 1e3;
}

int main() {
    // operands: A [M][K] and B [N][K] bf16.  Cases: an encoder projection (A from the Infinity Cache / L2), a conv-layer-like long A.
    struct Case { const char* name; int M, N, K; };
    const Case cases[] = {{"FFN1-like  M 6272 N 3072 K 768 ", 6272, 3072, 768}, {"FFN2-like  M 6272 N 768  K 3072", 6272, 768, 3072},
                          {"conv1-like M 200704 N 512 K 1536", 200704, 512, 1536}};
    for (const Case& c : cases) {
        char *A, *B;
        const size_t ab = (size_t)c.M * c.K * 2, bb = (size_t)c.N * c.K * 2;
        CK(hipMalloc(&A, ab)); CK(hipMalloc(&B, bb));
        CK(hipMemset(A, 1, ab)); CK(hipMemset(B, 1, bb));
        const long long lda = (long long)c.K * 2, ldb = lda;
        printf("== %s (A %.1f MB, B %.1f MB)\n", c.name, ab / 1e6, bb / 1e6);
        for (int bpc = 1; bpc <= 2; ++bpc) {
            const int grid_cap = 256 * bpc;
            auto report = [&](const char* pat, double us, int rows_per_tile, int krowb) {
                const int tm = c.M / rows_per_tile, tn = c.N / rows_per_tile, nk = c.K * 2 / krowb;
                const double bytes = (double)tm * tn * nk * 32768.0;
                printf("  %d block(s)/CU  %-44s %8.1f us  %6.1f GB/s per CU  (%.2f TB/s chip)\n", bpc, pat, us, bytes / us / 1e3 / 256, bytes / us / 1e6);
            };
#define RUN(PAT_, name_, rows_, krowb_) do { \
                const int tm = c.M / rows_, tn = c.N / rows_, nk = c.K * 2 / krowb_; \
                const int grid = std::min(grid_cap, tm * tn); \
                const double us = bpc == 1 ? run<PAT_, 160>(A, B, lda, ldb, tm, tn, nk, grid, 10) : run<PAT_, 80>(A, B, lda, ldb, tm, tn, nk, grid, 10); \
                report(name_, us, rows_, krowb_); } while (0)
            RUN(0, "8 rows x 128 B per piece (product)", 128, 128);
            RUN(1, "16 rows x 64 B, halves on different waves", 128, 128);
            RUN(2, "16 rows x 64 B, halves = a wave's 2 pieces", 128, 128);
            RUN(3, "4 rows x 256 B (k-tile 128, 64-row stage)", 64, 256);
#undef RUN
        }
        CK(hipFree(A)); CK(hipFree(B));
    }
    return 0;
}

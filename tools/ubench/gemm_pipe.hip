// Standalone experiment (not part of liba8hip): bf16 NT GEMM, 256x256 tile, 8 waves of 128x64, with the fragment reads of
// the NEXT quadrant issued ahead of the MFMAs of the current one and ONE barrier per k-tile placed mid-tile:
//
//   per k-tile t (LDS buffer b = t & 1), per wave, entering with A0, B0 (tile t) already in registers
//     read B1(t)                 | MFMA C00 += A0 B0
//     read A1(t)                 | MFMA C01 += A0 B1
//     wait: A1 in registers, this wave's DMA pieces of tile t+1 landed;   s_barrier
//        -> nobody reads buffer b any more (all four operand quarters are in registers), tile t+1 is visible in buffer b^1
//     DMA tile t+2 -> buffer b   | MFMA C11 += A1 B1
//     read A0(t+1), B0(t+1) from b^1 (into the registers B1 just vacated: the two B register sets swap roles every tile)
//                                | MFMA C10 += A1 B0
//
// so the DMA of a tile has a whole k-tile to land, no wave waits on an LDS read it has just issued, and the k-tile
// boundary has no barrier.  0.375 KB of fragment reads per MFMA (32x64 wave tiles: 0.75), half the staged bytes per FLOP.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench/gemm_pipe.hip -o /tmp/gemm_pipe && /tmp/gemm_pipe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define DEV __device__ __forceinline__
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int BM = 256, BN = 256, BK = 64, KROW = 128;          // KROW: bytes of K per row per k-tile
constexpr int TA = BM * KROW, TB = BN * KROW, STAGE = TA + TB;    // 64 KB per stage, two stages

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

DEV void glds16(const char* src, char* dst_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst_wave_uniform, 16, 0, 0);
}
DEV bf16x8 frag(const char* tile, int rb, int ks, int lane) {     // 16 rows x 32 k: lane&15 = row, lane>>4 = 16-byte chunk of the k-step
    const int row = rb + (lane & 15);
    return *reinterpret_cast<const bf16x8*>(tile + row * KROW + (((ks * 4 + (lane >> 4)) ^ (row & 7)) << 4));
}
DEV int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// EPI 0: C = bf16(acc).  EPI 1 (conv forward): Z = bf16(acc), C = bf16(act(acc)) with a GELU-cost activation.
// Persistent: block b takes tiles b, b + gridDim.x, ...; the next tile's first two k-tiles are DMA'd before the current
// tile's stores, so only the epilogue's own instructions are exposed.  lda: row stride of A in elements (conv rows overlap).
DEV float act(float v) {                                   // ~GELU cost (x * sigmoid(1.702 x)), benchmark only
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554f * v));
}
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_pipe(const bf16* __restrict__ A, long long lda, const bf16* __restrict__ B,
                                                    bf16* __restrict__ C, bf16* __restrict__ Z, int M, int N, int K, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), wr = w >> 2, wc = w & 3;
    const int total = tiles_m * tiles_n;
    const int nk = K / BK;
    constexpr int GROUP_M = 8;
    auto tile_origin = [&](int v, int& m0, int& n0) {
        const int tile = xcd_remap(v, total);
        const int tpg = GROUP_M * tiles_n, grp = tile / tpg, first_m = grp * GROUP_M;
        const int gsz = min(GROUP_M, tiles_m - first_m), tin = tile - grp * tpg;
        m0 = (first_m + tin % gsz) * BM;
        n0 = (tin / gsz) * BN;
    };

    // LDS-DMA: 32 one-KB pieces (8 rows) per operand per stage, 4 per wave (pieces w, w+8, w+16, w+24: rows 64 apart, so one
    // per-lane pointer per operand and scalar offsets); the XOR swizzle sits on the source address
    const char* a_src;
    const char* b_src;
    const int srow = 8 * w + (lane >> 3), sch = (lane & 7) ^ (srow & 7);
    auto set_src = [&](int m0, int n0) {
        a_src = reinterpret_cast<const char*>(A + (long long)(m0 + srow) * lda) + sch * 16;
        b_src = reinterpret_cast<const char*>(B + (long long)(n0 + srow) * K) + sch * 16;
    };
    const long long ajstep = 64ll * lda * 2, bjstep = 64ll * K * 2;          // bytes between pieces j and j+1 of a wave
    auto issue = [&](int kt) {
        char* ta = lds + (kt & 1) * STAGE;
        char* tb = ta + TA;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16(a_src + j * ajstep, ta + (w + 8 * j) * 1024);
            glds16(b_src + j * bjstep, tb + (w + 8 * j) * 1024);
        }
        a_src += KROW;
        b_src += KROW;
    };

    f32x4 acc[8][4];
    bf16x8 a0[2][4], a1[2][4], b0[2][2], b1[2][2];
    const int ra = wr * 128, cb = wc * 64;
    auto rd_a = [&](bf16x8 (&f)[2][4], const char* ta, int half) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) f[ks][i] = frag(ta, ra + 64 * half + 16 * i, ks, lane);
    };
    auto rd_b = [&](bf16x8 (&f)[2][2], const char* tb, int half) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) f[ks][i] = frag(tb, cb + 32 * half + 16 * i, ks, lane);
    };
    // swapped operands: accumulator register r <-> 4 consecutive n, lane & 15 <-> m
    auto mma = [&](const bf16x8 (&fa)[2][4], const bf16x8 (&fb)[2][2], int ah, int bh) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[4 * ah + i][2 * bh + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[ks][i], acc[4 * ah + i][2 * bh + j], 0, 0, 0);
    };

    // one k-tile; on entry a0 and bx hold the first A / B quarters of tile kt, by is free.  On exit a0 and by hold the first
    // quarters of tile kt+1 (read while the last quadrant's MFMAs run): the two B register sets swap roles every tile.
    auto step = [&](int kt, bf16x8 (&bx)[2][2], bf16x8 (&by)[2][2]) {
        const char* ta = lds + (kt & 1) * STAGE;
        const char* tb = ta + TA;
        const char* na = lds + ((kt + 1) & 1) * STAGE;
        const char* nb = na + TA;
        rd_b(by, tb, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, bx, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        rd_a(a1, ta, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, by, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) issue(kt + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, by, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            rd_a(a0, na, 0);
            rd_b(by, nb, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, bx, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    int m0, n0;
    tile_origin(blockIdx.x, m0, n0);
    set_src(m0, n0);
    issue(0);
    if (nk > 1) issue(1);
    for (int v = blockIdx.x; v < total; v += gridDim.x) {
        // k-tile 0 of this tile (and k-tile 1) were issued before the previous tile's stores
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        rd_a(a0, lds, 0);
        rd_b(b0, lds + TA, 0);
        for (int kt = 0; kt < nk; kt += 2) {
            step(kt, b0, b1);
            if (kt + 1 < nk) step(kt + 1, b1, b0);
        }
        // every fragment of this tile is in registers and every wave is past the last mid-tile barrier: the ring is free
        const int cm0 = m0, cn0 = n0;
        if (v + (int)gridDim.x < total) {
            tile_origin(v + gridDim.x, m0, n0);
            set_src(m0, n0);
            issue(0);
            if (nk > 1) issue(1);
        }
        __builtin_amdgcn_sched_barrier(0);
        // epilogue (benchmark-grade): 8-byte stores straight from the accumulators
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = cm0 + ra + 16 * i + (lane & 15), n = cn0 + cb + 16 * j + 4 * (lane >> 4);
                const f32x4 x = acc[i][j];
                if (EPI == 1) {
                    bf16x4 z = {(bf16)x[0], (bf16)x[1], (bf16)x[2], (bf16)x[3]};
                    *reinterpret_cast<bf16x4*>(Z + (long long)m * N + n) = z;
                    bf16x4 o = {(bf16)act(x[0]), (bf16)act(x[1]), (bf16)act(x[2]), (bf16)act(x[3])};
                    *reinterpret_cast<bf16x4*>(C + (long long)m * N + n) = o;
                } else {
                    bf16x4 o = {(bf16)x[0], (bf16)x[1], (bf16)x[2], (bf16)x[3]};
                    *reinterpret_cast<bf16x4*>(C + (long long)m * N + n) = o;
                }
            }
    }
}

__global__ void ref_gemm(const bf16* A, const bf16* B, float* C, int M, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)A[(long long)m * K + k] * (float)B[(long long)n * K + k];
    C[(long long)m * N + n] = s;
}
__global__ void fill(bf16* p, long long n, uint32_t seed, int integer) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 0x9E3779B1u + seed;
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
        p[i] = integer ? (bf16)(float)((int)(h % 5) - 2) : (bf16)(((float)(h >> 8) * (1.0f / 8388608.0f)) - 1.0f);
    }
}

template <int EPI>
static void launch(const bf16* A, long long lda, const bf16* B, bf16* C, bf16* Z, int M, int N, int K, int grid) {
    const int tm = M / BM, tn = N / BN;
    if (grid <= 0 || grid > tm * tn) grid = tm * tn;
    hipLaunchKernelGGL(gemm_pipe<EPI>, dim3(grid), dim3(512), 0, 0, A, lda, B, C, Z, M, N, K, tm, tn);
}

template <int EPI>
static void timeit(const char* what, const bf16* A, long long lda, const bf16* B, bf16* C, bf16* Z, int M, int N, int K, int grid) {
    for (int i = 0; i < 3; ++i) launch<EPI>(A, lda, B, C, Z, M, N, K, grid);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rnd = 0; rnd < 3; ++rnd) {
        CK(hipEventRecord(e0));
        const int it = 10;
        for (int i = 0; i < it; ++i) launch<EPI>(A, lda, B, C, Z, M, N, K, grid);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / it < best) best = ms / it;
    }
    printf("%-44s %dx%dx%d grid %5d: %.1f us  %.0f TFLOP/s\n", what, M, N, K, grid > 0 ? grid : (M / BM) * (N / BN), best * 1e3, 2.0 * M * N * K / (best * 1e-3) / 1e12);
}

int main() {
    {   // exactness on integer-valued operands (all partial sums exact in f32), through the persistent loop (12 tiles on 5 blocks)
        const int M = 1024, N = 768, K = 448;
        bf16 *A, *B, *C; float* R;
        CK(hipMalloc(&A, sizeof(bf16) * M * K)); CK(hipMalloc(&B, sizeof(bf16) * N * K)); CK(hipMalloc(&C, sizeof(bf16) * M * N)); CK(hipMalloc(&R, sizeof(float) * M * N));
        hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, A, (long long)M * K, 1u, 1);
        hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, B, (long long)N * K, 2u, 1);
        hipLaunchKernelGGL(ref_gemm, dim3((N + 255) / 256, M), dim3(256), 0, 0, A, B, R, M, N, K);
        std::vector<bf16> hc((size_t)M * N); std::vector<float> hr((size_t)M * N);
        for (int grid : {0, 5, 1}) {
            CK(hipMemset(C, 0xFF, sizeof(bf16) * M * N));
            launch<0>(A, K, B, C, nullptr, M, N, K, grid);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(hc.data(), C, sizeof(bf16) * M * N, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), R, sizeof(float) * M * N, hipMemcpyDeviceToHost));
            long long bad = 0;
            for (size_t i = 0; i < hc.size(); ++i) if ((float)hc[i] != (float)(bf16)hr[i]) { if (bad < 5) printf("mismatch at %zu: %f vs %f\n", i, (float)hc[i], hr[i]); ++bad; }
            printf("exactness %dx%dx%d grid %d: %lld mismatches\n", M, N, K, grid, bad);
            if (bad) return 1;
        }
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(R));
    }
    const int shapes[3][3] = {{4096, 4096, 8192}, {8192, 8192, 8192}, {8192, 4096, 4096}};
    for (int s = 0; s < 3; ++s) {
        const int M = shapes[s][0], N = shapes[s][1], K = shapes[s][2];
        bf16 *A, *B, *C;
        CK(hipMalloc(&A, sizeof(bf16) * (size_t)M * K)); CK(hipMalloc(&B, sizeof(bf16) * (size_t)N * K)); CK(hipMalloc(&C, sizeof(bf16) * (size_t)M * N));
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, A, (long long)M * K, 3u, 0);
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, B, (long long)N * K, 4u, 0);
        timeit<0>("square, one block per tile", A, K, B, C, nullptr, M, N, K, 0);
        timeit<0>("square, persistent (256 blocks)", A, K, B, C, nullptr, M, N, K, 256);
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
    }
    {   // conv layer 1 forward of the pretrain step: rows are overlapping windows (stride 1024 elements, 1536 long)
        const int M = 781 * 256, N = 512, K = 1536; const long long lda = 1024;
        bf16 *A, *B, *C, *Z;
        const size_t na = (size_t)(M - 1) * lda + K;
        CK(hipMalloc(&A, sizeof(bf16) * na)); CK(hipMalloc(&B, sizeof(bf16) * (size_t)N * K)); CK(hipMalloc(&C, sizeof(bf16) * (size_t)M * N)); CK(hipMalloc(&Z, sizeof(bf16) * (size_t)M * N));
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, A, (long long)na, 5u, 0);
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, B, (long long)N * K, 6u, 0);
        timeit<0>("conv1 shape, plain, one block per tile", A, lda, B, C, Z, M, N, K, 0);
        timeit<0>("conv1 shape, plain, persistent 256", A, lda, B, C, Z, M, N, K, 256);
        timeit<1>("conv1 shape, act + z, one block per tile", A, lda, B, C, Z, M, N, K, 0);
        timeit<1>("conv1 shape, act + z, persistent 256", A, lda, B, C, Z, M, N, K, 256);
        timeit<1>("conv1 shape, act + z, persistent 512", A, lda, B, C, Z, M, N, K, 512);
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(Z));
    }
    return 0;
}

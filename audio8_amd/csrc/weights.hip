// a8hip: per-step weight preparation -- for every 2-D weight (Linear, fused QKV, strided-conv kernels) write the
// compute-dtype working copy W2[r][c] AND its transpose W2^T[c][r] from the f32 master in one pass, so that forward
// (x W^T), data gradient (dy W = dy (W^T)^T) and the conv data gradient all run as K-contiguous "NT" GEMMs.
// Conv kernels [co][cin][kw] are re-laid-out on the fly to the GEMM's [co][kk*cin + ci].
// One C call prepares the whole model: the host passes a descriptor array; launches are issued back to back.
#include "a8_common.h"
#include "a8hip_internal.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void wprep_kernel(const float* __restrict__ src, T* __restrict__ dst, T* __restrict__ dstT,
                                                    int rows, int cols, int kw) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int cin = cols / kw;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + 8 * j, c = c0 + tx;
        float v = 0.f;
        if (r < rows && c < cols) {
            const int kk = c / cin, ci = c - kk * cin;                // logical column c = kk*cin + ci
            v = src[(long long)r * cols + (long long)ci * kw + kk];
            if (dst) dst[(long long)r * cols + c] = from_f<T>(v);
        }
        tile[ty + 8 * j][tx] = v;
    }
    __syncthreads();
    if (dstT) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + ty + 8 * j, r = r0 + tx;
            if (r < rows && c < cols) dstT[(long long)c * rows + r] = from_f<T>(tile[tx][ty + 8 * j]);
        }
    }
}

// single-launch variant: blockIdx.x walks a prefix table of 64x64 tiles over all matrices.  Reads are 256-byte row
// segments of the f32 master, the row-major copy leaves as 128-byte segments, and the transposed copy is written as
// 16-byte (8 x bf16) pieces of 128-byte transposed rows out of a padded LDS tile (the 32x32 / 2-byte-store version ran
// at 2.4 TB/s).
constexpr int WT = 64;
template <typename T>
__global__ __launch_bounds__(256) void wprep_multi_kernel(const a8_wprep_desc* __restrict__ descs, const int* __restrict__ tile_start, int n) {
    __shared__ float tile[WT][WT + 1];
    // binary search: largest i with tile_start[i] <= blockIdx.x
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tile_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const a8_wprep_desc d = descs[lo];
    const int tl = blockIdx.x - tile_start[lo];
    const int tcols = (d.cols + WT - 1) / WT;
    const int c0 = (tl % tcols) * WT, r0 = (tl / tcols) * WT;
    const int cin = d.cols / d.kw;
    T* dst = reinterpret_cast<T*>(d.dst);
    T* dstT = reinterpret_cast<T*>(d.dstT);
    const long long ldd = d.ld_dst ? d.ld_dst : d.cols, ldt = d.ld_dstT ? d.ld_dstT : d.rows;
    const bool plain = d.kw == 1 && d.tap < 0 && (d.cols & 3) == 0;
    {   // load (+ row-major copy): thread -> rows ty + 16 j, columns 4 tx .. 4 tx + 3
        const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rl = ty + 16 * j, r = r0 + rl, c = c0 + 4 * tx;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (r < d.rows) {
                if (plain && c + 3 < d.cols) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(d.src + (long long)r * d.cols + c);
                    v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int cc = c + e;
                        if (cc < d.cols) {
                            if (d.tap >= 0) {
                                v[e] = d.src[((long long)r * d.cols + cc) * d.kw + d.tap];
                            } else {
                                const int kk = cc / cin, ci = cc - kk * cin;        // logical column = kk*cin + ci
                                v[e] = d.src[(long long)r * d.cols + (long long)ci * d.kw + kk];
                            }
                        }
                    }
                }
                if (dst) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e < d.cols) dst[(long long)r * ldd + c + e] = from_f<T>(v[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[rl][4 * tx + e] = v[e];
        }
    }
    __syncthreads();
    if (dstT) {   // transposed copy: thread -> transposed rows (columns) ty + 32 j, source rows 8 tx .. 8 tx + 7
        const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cl = ty + 32 * j, c = c0 + cl, r = r0 + 8 * tx;
            if (c >= d.cols) continue;
            T* o = dstT + (long long)c * ldt + r;
            if (r + 7 < d.rows && ((ldt * (long long)sizeof(T)) & 15) == 0 && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
                typedef T T8 __attribute__((ext_vector_type(8)));
                T8 q;
#pragma unroll
                for (int e = 0; e < 8; ++e) q[e] = from_f<T>(tile[8 * tx + e][cl]);
                *reinterpret_cast<T8*>(o) = q;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (r + e < d.rows) o[e] = from_f<T>(tile[8 * tx + e][cl]);
            }
        }
    }
}

// grad[o][c][kk] += dw2[o][kk][c]; dw2[o][kk][c] = 0.  One thread per (o, c): reads are contiguous along c, the k writes of a thread are adjacent.
__global__ __launch_bounds__(256) void conv_wgrad_fold_kernel(float* __restrict__ dw2, float* __restrict__ grad, int co, int cin, int k) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= (long long)co * cin) return;
    const int o = (int)(i / cin), c = (int)(i - (long long)o * cin);
    for (int kk = 0; kk < k; ++kk) {
        float* src = dw2 + ((long long)o * k + kk) * cin + c;
        grad[i * k + kk] += *src;
        *src = 0.f;
    }
}

}  // namespace

extern "C" int a8_prepare_weights(int dtype, const a8_wprep_desc* descs, int n, void* stream) {
    A8_REQUIRE(dtype == A8_F32 || dtype == A8_BF16, "a8_prepare_weights: bad dtype");
    A8_REQUIRE(n >= 0 && (n == 0 || descs), "a8_prepare_weights: bad descriptor array");
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < n; ++i) {
        const a8_wprep_desc& d = descs[i];
        A8_REQUIRE(d.rows > 0 && d.cols > 0 && d.kw >= 1 && d.cols % d.kw == 0 && d.src, "a8_prepare_weights: bad descriptor %d", i);
        A8_REQUIRE(d.tap < 0 && d.ld_dst == 0 && d.ld_dstT == 0, "a8_prepare_weights: tap / leading dimensions need the device-table form");
        const dim3 grid(cdiv(d.cols, 32), cdiv(d.rows, 32));
        if (dtype == A8_BF16) hipLaunchKernelGGL(wprep_kernel<bf16>, grid, dim3(256), 0, s, d.src, (bf16*)d.dst, (bf16*)d.dstT, d.rows, d.cols, d.kw);
        else hipLaunchKernelGGL(wprep_kernel<float>, grid, dim3(256), 0, s, d.src, (float*)d.dst, (float*)d.dstT, d.rows, d.cols, d.kw);
    }
    return a8_check_launch("a8_prepare_weights");
}

// descs/tile_start live in DEVICE memory (built once by the host); tile_start[i] = first 64x64-tile index of matrix i,
// total_tiles = tile_start[n].  One launch prepares every weight of the model.
extern "C" int a8_prepare_weights_device(int dtype, const a8_wprep_desc* descs_dev, const int* tile_start_dev, int n, int total_tiles, void* stream) {
    A8_REQUIRE(dtype == A8_F32 || dtype == A8_BF16, "a8_prepare_weights_device: bad dtype");
    if (n == 0 || total_tiles == 0) return A8_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == A8_BF16) hipLaunchKernelGGL(wprep_multi_kernel<bf16>, dim3(total_tiles), dim3(256), 0, s, descs_dev, tile_start_dev, n);
    else hipLaunchKernelGGL(wprep_multi_kernel<float>, dim3(total_tiles), dim3(256), 0, s, descs_dev, tile_start_dev, n);
    return a8_check_launch("a8_prepare_weights_device");
}

extern "C" int a8_zero(void* p, int64_t bytes, void* stream) {
    A8_REQUIRE(bytes >= 0 && (p != nullptr || bytes == 0), "a8_zero: bad arguments");
    if (bytes == 0) return A8_OK;
    const hipError_t e = hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream);
    if (e != hipSuccess) return a8_set_error(A8_ERR_LAUNCH, "a8_zero: %s", hipGetErrorString(e));
    return A8_OK;
}

extern "C" int a8_conv_wgrad_fold(float* dw2, float* grad, int co, int cin, int k, void* stream) {
    A8_REQUIRE(dw2 && grad && co >= 1 && cin >= 1 && k >= 1, "a8_conv_wgrad_fold: bad arguments");
    hipLaunchKernelGGL(conv_wgrad_fold_kernel, dim3(cdiv((long long)co * cin, 256)), dim3(256), 0, (hipStream_t)stream, dw2, grad, co, cin, k);
    return a8_check_launch("a8_conv_wgrad_fold");
}

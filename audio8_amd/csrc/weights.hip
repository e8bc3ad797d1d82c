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

// single-launch variant: blockIdx.x walks a prefix table of 32x32 tiles over all matrices
template <typename T>
__global__ __launch_bounds__(256) void wprep_multi_kernel(const a8_wprep_desc* __restrict__ descs, const int* __restrict__ tile_start, int n) {
    __shared__ float tile[32][33];
    // binary search: largest i with tile_start[i] <= blockIdx.x
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tile_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const a8_wprep_desc d = descs[lo];
    const int tl = blockIdx.x - tile_start[lo];
    const int tcols = (d.cols + 31) >> 5;
    const int c0 = (tl % tcols) * 32, r0 = (tl / tcols) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int cin = d.cols / d.kw;
    T* dst = reinterpret_cast<T*>(d.dst);
    T* dstT = reinterpret_cast<T*>(d.dstT);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + 8 * j, c = c0 + tx;
        float v = 0.f;
        if (r < d.rows && c < d.cols) {
            if (d.tap >= 0) {
                v = d.src[((long long)r * d.cols + c) * d.kw + d.tap];
            } else {
                const int kk = c / cin, ci = c - kk * cin;
                v = d.src[(long long)r * d.cols + (long long)ci * d.kw + kk];
            }
            if (dst) dst[(long long)r * (d.ld_dst ? d.ld_dst : d.cols) + c] = from_f<T>(v);
        }
        tile[ty + 8 * j][tx] = v;
    }
    __syncthreads();
    if (dstT) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + ty + 8 * j, r = r0 + tx;
            if (r < d.rows && c < d.cols) dstT[(long long)c * (d.ld_dstT ? d.ld_dstT : d.rows) + r] = from_f<T>(tile[tx][ty + 8 * j]);
        }
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

// descs/tile_start live in DEVICE memory (built once by the host); tile_start[i] = first 32x32-tile index of matrix i,
// total_tiles = tile_start[n].  One launch prepares every weight of the model.
extern "C" int a8_prepare_weights_device(int dtype, const a8_wprep_desc* descs_dev, const int* tile_start_dev, int n, int total_tiles, void* stream) {
    A8_REQUIRE(dtype == A8_F32 || dtype == A8_BF16, "a8_prepare_weights_device: bad dtype");
    if (n == 0 || total_tiles == 0) return A8_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == A8_BF16) hipLaunchKernelGGL(wprep_multi_kernel<bf16>, dim3(total_tiles), dim3(256), 0, s, descs_dev, tile_start_dev, n);
    else hipLaunchKernelGGL(wprep_multi_kernel<float>, dim3(total_tiles), dim3(256), 0, s, descs_dev, tile_start_dev, n);
    return a8_check_launch("a8_prepare_weights_device");
}

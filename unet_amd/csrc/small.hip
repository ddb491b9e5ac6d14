// Thin-channel kernels: the first encoder convolution (1 -> N, stride 2), the last decoder
// transposed convolution (N -> 1, stride 2), their weight gradients, and the one-channel
// 3^d stencil of the top residual unit.  All are bandwidth-bound VALU kernels; the
// one-channel tensors are fp32 and share PyTorch's contiguous layout.
#include "common.h"
#include "kparams.h"

namespace {

// ---- 1 -> N stride-2 convolution (k=3, pad=1), fine one-channel -> coarse N-channel ------
template <class TR>
__global__ __launch_bounds__(256) void c1_s2_fwd_kernel(const C1Params p) {
    constexpr int ESZ = TR::ESZ;
    extern __shared__ __attribute__((aligned(16))) float sm[];  // weights [N][KK], then partials [2][ppb][N]
    const int KZ = p.ndim == 3 ? 3 : 1;
    const int KK = KZ * 9;
    const int G = p.N / 8;
    const int img = blockIdx.y;
    float* wl = sm;
    float* sst = sm + p.N * KK;
    for (int i = threadIdx.x; i < p.N * KK; i += 256) {
        const int n = i / KK, k = i - n * KK;
        wl[i] = (n < p.n_w1) ? p.w[n * KK + k] : p.w2[(n - p.n_w1) * KK + k];
    }
    __syncthreads();
    const int cpix = p.cd[0] * p.cd[1] * p.cd[2];
    const int grp = threadIdx.x % G, pl = threadIdx.x / G;
    const int ppb = 256 / G;
    const int pix = blockIdx.x * ppb + pl;
    const bool valid = pl < ppb && pix < cpix;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (valid) {
        const int cx = pix % p.cd[2]; const int r = pix / p.cd[2];
        const int cy = r % p.cd[1]; const int cz = r / p.cd[1];
        const float* xin = p.x1 + (size_t)img * p.fd[0] * p.fd[1] * p.fd[2];
        for (int kz = 0; kz < KZ; ++kz) {
            const int fz = (p.ndim == 3) ? 2 * cz - 1 + kz : 0;
            if ((unsigned)fz >= (unsigned)p.fd[0]) continue;
            for (int ky = 0; ky < 3; ++ky) {
                const int fy = 2 * cy - 1 + ky;
                if ((unsigned)fy >= (unsigned)p.fd[1]) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int fx = 2 * cx - 1 + kx;
                    if ((unsigned)fx >= (unsigned)p.fd[2]) continue;
                    const float xv = xin[((size_t)fz * p.fd[1] + fy) * p.fd[2] + fx];
                    const int k = (kz * 3 + ky) * 3 + kx;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += xv * wl[(grp * 8 + e) * KK + k];
                }
            }
        }
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = grp * 8 + e;
            o[e] = acc[e] + ((p.bias && ch >= p.bias_c0) ? p.bias[ch - p.bias_c0] : 0.f);
        }
        const size_t gp = (size_t)img * cpix + pix;
        store_n<TR, 8>(p.y_out + (gp * p.y_pitch + p.y_coff + grp * 8) * ESZ, o);
    }
    if (p.stats) {
        // fixed-order reduction over the workgroup's pixels, then fp64 accumulators
        if (pl < ppb) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = valid ? acc[e] : 0.f;
                sst[pl * p.N + grp * 8 + e] = v;
                sst[(ppb + pl) * p.N + grp * 8 + e] = v * v;
            }
        }
        __syncthreads();
        const int slot = blockIdx.x % MI_NSLOT;
        for (int i = threadIdx.x; i < p.nstat * 2; i += 256) {
            const int ch = i >> 1, k = i & 1;
            float a = 0.f;
            for (int r = 0; r < ppb; ++r) a += sst[(k * ppb + r) * p.N + ch];
            atomicAdd(&p.stats[(((size_t)img * MI_NSLOT + slot) * p.stats_c) * 2 + i], (double)a);
        }
    }
}

// ---- N -> 1 stride-2 transposed convolution (k=3, pad=1, output_padding=1) ---------------
template <class TR>
__global__ __launch_bounds__(256) void c1_s2_T_kernel(const C1Params p) {
    constexpr int ESZ = TR::ESZ;
    extern __shared__ __attribute__((aligned(16))) float sm[];  // weights transposed [KK][N]
    __shared__ float red[2][4];
    const int KZ = p.ndim == 3 ? 3 : 1;
    const int KK = KZ * 9;
    const int img = blockIdx.y;
    for (int i = threadIdx.x; i < p.N * KK; i += 256) {
        const int n = i / KK, k = i - n * KK;
        sm[k * p.N + n] = p.w[i];
    }
    __syncthreads();
    const int fpix = p.fd[0] * p.fd[1] * p.fd[2];
    const int cpix = p.cd[0] * p.cd[1] * p.cd[2];
    const int pix = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    const bool valid = pix < fpix;
    if (valid) {
        const int fx = pix % p.fd[2]; const int r = pix / p.fd[2];
        const int fy = r % p.fd[1]; const int fz = r / p.fd[1];
        // per dim: even f -> (k=1, c=f/2); odd f -> (k=0, c=(f+1)/2) and (k=2, c=(f-1)/2)
        int kzs[2], czs[2], nz = 0, kys[2], cys[2], ny = 0, kxs[2], cxs[2], nx = 0;
        if (p.ndim == 3) {
            if (fz & 1) {
                if ((fz + 1) / 2 < p.cd[0]) { kzs[nz] = 0; czs[nz++] = (fz + 1) / 2; }
                kzs[nz] = 2; czs[nz++] = (fz - 1) / 2;
            } else { kzs[nz] = 1; czs[nz++] = fz / 2; }
        } else { kzs[0] = 0; czs[0] = 0; nz = 1; }
        if (fy & 1) {
            if ((fy + 1) / 2 < p.cd[1]) { kys[ny] = 0; cys[ny++] = (fy + 1) / 2; }
            kys[ny] = 2; cys[ny++] = (fy - 1) / 2;
        } else { kys[ny] = 1; cys[ny++] = fy / 2; }
        if (fx & 1) {
            if ((fx + 1) / 2 < p.cd[2]) { kxs[nx] = 0; cxs[nx++] = (fx + 1) / 2; }
            kxs[nx] = 2; cxs[nx++] = (fx - 1) / 2;
        } else { kxs[nx] = 1; cxs[nx++] = fx / 2; }
        for (int a = 0; a < nz; ++a)
            for (int b = 0; b < ny; ++b)
                for (int c = 0; c < nx; ++c) {
                    const int k = (kzs[a] * 3 + kys[b]) * 3 + kxs[c];
                    const size_t gp = (size_t)img * cpix + ((size_t)czs[a] * p.cd[1] + cys[b]) * p.cd[2] + cxs[c];
                    const char* src = p.y + (gp * p.y_pitch + p.y_coff) * ESZ;
                    const float* wk = sm + k * p.N;
                    for (int n0 = 0; n0 < p.N; n0 += 8) {
                        float v[8];
                        load_n<TR, 8>(src + n0 * ESZ, v);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc += v[e] * wk[n0 + e];
                    }
                }
        p.x1_out[(size_t)img * fpix + pix] = acc + (p.bias ? p.bias[0] : 0.f);
    }
    if (p.stats) {
        float a = valid ? acc : 0.f, b = a * a;
        a = wave_sum(a); b = wave_sum(b);
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red[0][w] = a; red[1][w] = b; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int slot = blockIdx.x % MI_NSLOT;
            atomicAdd(&p.stats[((size_t)img * MI_NSLOT + slot) * 2], (double)(((red[0][0] + red[0][1]) + red[0][2]) + red[0][3]));
            atomicAdd(&p.stats[((size_t)img * MI_NSLOT + slot) * 2 + 1], (double)(((red[1][0] + red[1][1]) + red[1][2]) + red[1][3]));
        }
    }
}

// ---- weight gradient of both: partial[slab][k][n] = sum_c Y[c][n] * x1[2c-1+k] ;
//      partial_bias[slab][n] = sum_c Y[c][n].  blockIdx.y = kz plane in 3-D. ----------------
template <class TR>
__global__ __launch_bounds__(256) void c1_s2_wgrad_kernel(const C1Params p) {
    constexpr int ESZ = TR::ESZ;
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [4 waves][N][10]
    const int KZ = p.ndim == 3 ? 3 : 1;
    const int KK = KZ * 9;
    const int kz = blockIdx.y;
    const int G = p.N / 8;
    for (int i = threadIdx.x; i < 4 * p.N * 10; i += 256) sm[i] = 0.f;
    __syncthreads();
    const int cpix = p.cd[0] * p.cd[1] * p.cd[2];
    const int fpix = p.fd[0] * p.fd[1] * p.fd[2];
    const long long items = (long long)p.nb * cpix * G;
    float acc[8][9], accb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        accb[e] = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[e][k] = 0.f;
    }
    const int grp = threadIdx.x % G, pl = threadIdx.x / G;
    const int ppb = 256 / G;  // (pixel, channel group) pairs per sweep; the group of a thread never changes
    const long long npx = items / G;
    for (long long gp = (long long)blockIdx.x * ppb + pl; pl < ppb && gp < npx; gp += (long long)gridDim.x * ppb) {
        const int img = (int)(gp / cpix); const int pix = (int)(gp - (long long)img * cpix);
        const int cx = pix % p.cd[2]; const int r = pix / p.cd[2];
        const int cy = r % p.cd[1]; const int cz = r / p.cd[1];
        float v[8];
        load_n<TR, 8>(p.y + ((size_t)gp * p.y_pitch + p.y_coff + grp * 8) * ESZ, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) accb[e] += v[e];
        const int fz = (p.ndim == 3) ? 2 * cz - 1 + kz : 0;
        if ((unsigned)fz >= (unsigned)p.fd[0]) continue;
        const float* xin = p.x1 + (size_t)img * fpix + (size_t)fz * p.fd[1] * p.fd[2];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int fy = 2 * cy - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int fx = 2 * cx - 1 + kx;
                float xv = 0.f;
                if ((unsigned)fy < (unsigned)p.fd[1] && (unsigned)fx < (unsigned)p.fd[2]) xv = xin[(size_t)fy * p.fd[2] + fx];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e][ky * 3 + kx] += v[e] * xv;
            }
        }
    }
    // lanes that share a channel group: xor-reduce over lane bits >= log2(G) when G divides the wave
    const bool pow2 = (64 % G) == 0;
    if (pow2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                float a = acc[e][k];
                for (int o = G; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
                acc[e][k] = a;
            }
            float b = accb[e];
            for (int o = G; o < 64; o <<= 1) b += __shfl_xor(b, o, 64);
            accb[e] = b;
        }
    }
    // one LDS image per wave; when the channel groups tile the wave each entry has a single writer and the
    // result is order-independent, otherwise (N/8 not a power of two) LDS float atomics are used
    float* mine = sm + (threadIdx.x >> 6) * p.N * 10;
    if (pow2) {
        if ((threadIdx.x & 63) < G) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int k = 0; k < 9; ++k) mine[(grp * 8 + e) * 10 + k] = acc[e][k];
                mine[(grp * 8 + e) * 10 + 9] = accb[e];
            }
        }
    } else if (pl < ppb) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int k = 0; k < 9; ++k) atomicAdd(&mine[(grp * 8 + e) * 10 + k], acc[e][k]);
            atomicAdd(&mine[(grp * 8 + e) * 10 + 9], accb[e]);
        }
    }
    __syncthreads();
    const size_t slab = blockIdx.x;
    const int wstride = p.N * 10;
    for (int i = threadIdx.x; i < p.N * 9; i += 256) {
        const int n = i / 9, k = i - n * 9;
        const int o = n * 10 + k;
        p.partial[(slab * KK + kz * 9 + k) * p.N + n] = ((sm[o] + sm[wstride + o]) + sm[2 * wstride + o]) + sm[3 * wstride + o];
    }
    if (kz == 0)
        for (int n = threadIdx.x; n < p.N; n += 256) {
            const int o = n * 10 + 9;
            p.partial_bias[slab * p.N + n] = ((sm[o] + sm[wstride + o]) + sm[2 * wstride + o]) + sm[3 * wstride + o];
        }
}

// ---- one-channel 3^d stencil: out = sum_k w[k] * in[p + k - 1] + bias + addend ------------
__global__ __launch_bounds__(256) void stencil_kernel(const StencilParams p) {
    const int KZ = p.ndim == 3 ? 3 : 1;
    const int KK = KZ * 9;
    const int img = blockIdx.y;
    const int npix = p.d[0] * p.d[1] * p.d[2];
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= npix) return;
    const int x = pix % p.d[2]; const int r = pix / p.d[2];
    const int y = r % p.d[1]; const int z = r / p.d[1];
    const float* in = p.in + (size_t)img * npix;
    float acc = p.bias ? p.bias[0] : 0.f;
    for (int kz = 0; kz < KZ; ++kz) {
        const int zz = (p.ndim == 3) ? z - 1 + kz : 0;
        if ((unsigned)zz >= (unsigned)p.d[0]) continue;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y - 1 + ky;
            if ((unsigned)yy >= (unsigned)p.d[1]) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = x - 1 + kx;
                if ((unsigned)xx >= (unsigned)p.d[2]) continue;
                const int k = (kz * 3 + ky) * 3 + kx;
                acc += p.w[p.flip ? KK - 1 - k : k] * in[((size_t)zz * p.d[1] + yy) * p.d[2] + xx];
            }
        }
    }
    if (p.addend) acc += p.addend[(size_t)img * npix + pix];
    p.out[(size_t)img * npix + pix] = acc;
}

// partial[slab][k] = sum_p in[p + k - 1] * dout[p];  partial_bias[slab] = sum_p dout[p]
__global__ __launch_bounds__(256) void stencil_wgrad_kernel(const StencilParams p) {
    __shared__ float red[10][4];
    const int KZ = p.ndim == 3 ? 3 : 1;
    const int KK = KZ * 9;
    const int kz = blockIdx.y;
    const int npix = p.d[0] * p.d[1] * p.d[2];
    const long long total = (long long)p.nb * npix;
    float acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = 0.f;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
        const int img = (int)(it / npix); const int pix = (int)(it - (long long)img * npix);
        const int x = pix % p.d[2]; const int r = pix / p.d[2];
        const int y = r % p.d[1]; const int z = r / p.d[1];
        const float g = p.dout[it];
        acc[9] += g;
        const int zz = (p.ndim == 3) ? z - 1 + kz : 0;
        if ((unsigned)zz >= (unsigned)p.d[0]) continue;
        const float* in = p.in + (size_t)img * npix + (size_t)zz * p.d[1] * p.d[2];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = x - 1 + kx;
                float v = 0.f;
                if ((unsigned)yy < (unsigned)p.d[1] && (unsigned)xx < (unsigned)p.d[2]) v = in[(size_t)yy * p.d[2] + xx];
                acc[ky * 3 + kx] += v * g;
            }
        }
    }
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const float a = wave_sum(acc[k]);
        if ((threadIdx.x & 63) == 0) red[k][w] = a;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        const float a = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        if (threadIdx.x < 9) p.partial[(size_t)blockIdx.x * KK + kz * 9 + threadIdx.x] = a;
        else if (kz == 0) p.partial_bias[blockIdx.x] = a;
    }
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string(what) + ": " + hipGetErrorString(e)); return -2; }
    return 0;
}

}  // namespace

int launch_c1_s2_fwd(int dtype, const C1Params& p, hipStream_t st) {
    MI_CHECK(p.N % 8 == 0 && p.N >= 8 && p.N <= 256, "N must be a multiple of 8, at most 256");
    const int KK = p.ndim == 3 ? 27 : 9;
    const int cpix = p.cd[0] * p.cd[1] * p.cd[2];
    const int ppb = 256 / (p.N / 8);
    const size_t lds = (size_t)(p.N * KK + 2 * ppb * p.N) * sizeof(float);
    dim3 grid((cpix + ppb - 1) / ppb, p.nb);
    if (dtype) hipLaunchKernelGGL(c1_s2_fwd_kernel<BF16>, grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL(c1_s2_fwd_kernel<F32>, grid, dim3(256), lds, st, p);
    return check_launch("c1_s2_fwd");
}

int launch_c1_s2_T(int dtype, const C1Params& p, hipStream_t st) {
    MI_CHECK(p.N % 8 == 0 && p.N >= 8 && p.N <= 256, "N must be a multiple of 8, at most 256");
    const int KK = p.ndim == 3 ? 27 : 9;
    const int fpix = p.fd[0] * p.fd[1] * p.fd[2];
    const size_t lds = (size_t)p.N * KK * sizeof(float);
    dim3 grid((fpix + 255) / 256, p.nb);
    if (dtype) hipLaunchKernelGGL(c1_s2_T_kernel<BF16>, grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL(c1_s2_T_kernel<F32>, grid, dim3(256), lds, st, p);
    return check_launch("c1_s2_T");
}

// partial: [nsplit][KK][N] floats, partial_bias: [nsplit][N] floats
int launch_c1_s2_wgrad(int dtype, const C1Params& p, hipStream_t st) {
    MI_CHECK(p.N % 8 == 0 && p.N >= 8 && p.N <= 512, "N must be a multiple of 8, at most 512");
    const size_t lds = (size_t)4 * p.N * 10 * sizeof(float);
    dim3 grid(p.nsplit, p.ndim == 3 ? 3 : 1);
    if (dtype) hipLaunchKernelGGL(c1_s2_wgrad_kernel<BF16>, grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL(c1_s2_wgrad_kernel<F32>, grid, dim3(256), lds, st, p);
    return check_launch("c1_s2_wgrad");
}

int launch_stencil(const StencilParams& p, hipStream_t st) {
    const int npix = p.d[0] * p.d[1] * p.d[2];
    hipLaunchKernelGGL(stencil_kernel, dim3((npix + 255) / 256, p.nb), dim3(256), 0, st, p);
    return check_launch("stencil");
}

int launch_stencil_wgrad(const StencilParams& p, hipStream_t st) {
    hipLaunchKernelGGL(stencil_wgrad_kernel, dim3(p.nsplit, p.ndim == 3 ? 3 : 1), dim3(256), 0, st, p);
    return check_launch("stencil_wgrad");
}

// Thin-channel kernels: the first encoder convolution (1 -> N, stride 2), the last decoder
// transposed convolution (N -> 1, stride 2), their weight gradients, and the one-channel
// 3^d stencil of the top residual unit.  All are bandwidth-bound VALU kernels; the
// one-channel tensors are fp32 and share PyTorch's contiguous layout.
#include <type_traits>
#include "common.h"
#include "kparams.h"

namespace {

constexpr int C1_WGRAD_CG = 8;   // channels per thread of c1_s2_wgrad_kernel
constexpr int C1W_RU = 4;        // rows in flight per thread of c1_s2_wgrad_kernel
constexpr int C1F_YG = 1;        // groups of four coarse rows per workgroup of c1_s2_fwd_kernel
constexpr int C1T_YP = 4;        // patches along y per workgroup of c1_s2_T_kernel

// ---- 1 -> N stride-2 convolution (k=3, pad=1), fine one-channel -> coarse N-channel ------
// Thread = (coarse column, group of 8 output channels); a workgroup covers ppb = 256/(N/8) columns x ROWS rows.
// The 72 weights of a kz plane sit in registers, the (2*ROWS+1) x 3 input values of the thread's column are
// loaded up front (independent loads), and each thread keeps running sums for the norm statistics.
template <class TR, int KZ>
__global__ __launch_bounds__(256) void c1_s2_fwd_kernel(const C1Params p) {
    constexpr int ESZ = TR::ESZ;
    constexpr int KK = KZ * 9;
    constexpr int ROWS = 4;
    extern __shared__ __attribute__((aligned(16))) float sm[];  // weights [KK][N], then partials [2][ppb][N]
    const int N = p.N;
    const int G = N / 8;
    const int nthr = blockDim.x;                // 256, or fewer when a row of the coarse grid has fewer than 256 / G columns
    const int ppb = nthr / G;
    float* wl = sm;
    float* sst = sm + N * KK;
    const int grp = threadIdx.x % G, pl = threadIdx.x / G;
    const int cz = blockIdx.z % p.cd[0], img = blockIdx.z / p.cd[0];
    const int cx = blockIdx.x * ppb + pl;
    const bool lane_ok = pl < ppb && cx < p.cd[2];
    const size_t fpix = (size_t)p.fd[0] * p.fd[1] * p.fd[2];
    const size_t cpix = (size_t)p.cd[0] * p.cd[1] * p.cd[2];
    const float* xin = p.x1 + (size_t)img * fpix;
    // the input values of one (row group, kz plane): (2*ROWS+1) x 3 independent loads
    auto load_rows = [&](int cy0, int fz, float (*xr)[3]) {
#pragma unroll
        for (int j = 0; j < 2 * ROWS + 1; ++j) {
            const int fy = 2 * cy0 - 1 + j;
            const bool rok = lane_ok && (unsigned)fy < (unsigned)p.fd[1];
            const float* row = xin + ((size_t)fz * p.fd[1] + (rok ? fy : 0)) * p.fd[2];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int fx = 2 * cx - 1 + kx;
                xr[j][kx] = (rok && (unsigned)fx < (unsigned)p.fd[2]) ? row[fx] : 0.f;
            }
        }
    };
    // 2-D: the first row group's inputs are requested before the weights are staged, so the two round trips overlap
    float xr0[2 * ROWS + 1][3];
    if (KZ == 1) load_rows(blockIdx.y * C1F_YG * ROWS, 0, xr0);
    for (int i = threadIdx.x; i < N * KK; i += nthr) {
        const int n = i / KK, k = i - n * KK;
        wl[k * N + n] = (n < p.n_w1) ? p.w[n * KK + k] : p.w2[(n - p.n_w1) * KK + k];
    }
    __syncthreads();
    float bv[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = grp * 8 + e;
        bv[e] = (p.bias && ch >= p.bias_c0 && pl < ppb) ? p.bias[ch - p.bias_c0] : 0.f;
        s1[e] = 0.f; s2[e] = 0.f;
    }
    // a workgroup walks C1F_YG consecutive groups of ROWS coarse rows with the weight image staged once
    for (int yg = 0; yg < C1F_YG; ++yg) {
    const int cy0 = (blockIdx.y * C1F_YG + yg) * ROWS;
    if (cy0 >= p.cd[1]) break;
    float acc[ROWS][8];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[r][e] = 0.f;
#pragma unroll
    for (int kz = 0; kz < KZ; ++kz) {
        const int fz = (KZ == 3) ? 2 * cz - 1 + kz : 0;
        if ((unsigned)fz >= (unsigned)p.fd[0]) continue;   // uniform over the workgroup
        float xr[2 * ROWS + 1][3];
        if (KZ == 1 && yg == 0) {
#pragma unroll
            for (int j = 0; j < 2 * ROWS + 1; ++j)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) xr[j][kx] = xr0[j][kx];
        } else {
            load_rows(cy0, fz, xr);
        }
        // weights of one tap at a time from LDS (two 16-byte reads): keeps the register count low enough
        // for six waves per SIMD, which is what hides the latency of the gather loads
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 a = *reinterpret_cast<const float4*>(wl + (kz * 9 + ky * 3 + kx) * N + grp * 8);
                const float4 b = *reinterpret_cast<const float4*>(wl + (kz * 9 + ky * 3 + kx) * N + grp * 8 + 4);
                const float w8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const float xv = xr[2 * r + ky][kx];
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[r][e] += xv * w8[e];
                }
            }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int cy = cy0 + r;
        if (lane_ok && cy < p.cd[1]) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = acc[r][e] + bv[e];
                s1[e] += acc[r][e];
                s2[e] += acc[r][e] * acc[r][e];
            }
            const size_t gp = (size_t)img * cpix + ((size_t)cz * p.cd[1] + cy) * p.cd[2] + cx;
            char* dst = (p.y2_out && grp * 8 >= p.n_o1) ? p.y2_out + (gp * p.y2o_pitch + p.y2o_coff + grp * 8 - p.n_o1) * ESZ
                                                       : p.y_out + (gp * p.y_pitch + p.y_coff + grp * 8) * ESZ;
            store_n<TR, 8>(dst, o);
        }
    }
    }   // row groups
    if (p.stats) {
        // fixed-order reduction over the workgroup's columns, then fp64 accumulators
        const int slot = (blockIdx.x + blockIdx.y) % MI_NSLOT;
        if (G == 4) {
            // four channel groups per column: a 16-lane row holds four columns, two DPP row shifts add them up in
            // lanes 12..15; the 16 rows of the workgroup then meet in LDS ([2][16 rows][N])
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1[e]), 0x114, 0xf, 0xf, true));
                s1[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1[e]), 0x118, 0xf, 0xf, true));
                s2[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2[e]), 0x114, 0xf, 0xf, true));
                s2[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2[e]), 0x118, 0xf, 0xf, true));
            }
            const int row = threadIdx.x >> 4;
            if ((threadIdx.x & 15) >= 12) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sst[row * N + grp * 8 + e] = s1[e];
                    sst[(16 + row) * N + grp * 8 + e] = s2[e];
                }
            }
            __syncthreads();
            const int nrow = nthr >> 4;
            for (int i = threadIdx.x; i < p.nstat * 2; i += nthr) {
                const int ch = i >> 1, k = i & 1;
                float a = 0.f;
                for (int r = 0; r < nrow; ++r) a += sst[(k * 16 + r) * N + ch];
                atomicAdd(&p.stats[(((size_t)img * MI_NSLOT + slot) * p.stats_c) * 2 + i], (double)a);
            }
        } else {
            if (pl < ppb) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sst[pl * N + grp * 8 + e] = s1[e];
                    sst[(ppb + pl) * N + grp * 8 + e] = s2[e];
                }
            }
            __syncthreads();
            for (int i = threadIdx.x; i < p.nstat * 2; i += nthr) {
                const int ch = i >> 1, k = i & 1;
                float a = 0.f;
                for (int r = 0; r < ppb; ++r) a += sst[(k * ppb + r) * N + ch];
                atomicAdd(&p.stats[(((size_t)img * MI_NSLOT + slot) * p.stats_c) * 2 + i], (double)a);
            }
        }
    }
}

// ---- the same in 3-D on the matrix cores (bf16 storage, N = 16 or 32) -------------------------------------------------
// 27 taps x N channels per output pixel are 864 FMAs per thread in the kernel above, which is bound by issuing them.
// Here the 27 taps are the contraction of one MFMA step (padded to 32): a wave takes 16 consecutive coarse columns of
// one row, gathers each lane's eight taps from an LDS tile of the input (3 planes x 17 rows of the workgroup's
// 8 coarse rows) and multiplies by the weight fragments it keeps in registers.  The input and the weights stay fp32
// in effect: each is split into a bf16 head and a bf16 remainder (x = hi + lo to 16 mantissa bits) and the three
// products hi*hi, hi*lo, lo*hi are accumulated in fp32, so the result differs from the fp32 FMA chain by ~2^-16
// relative, far below the bf16 rounding of the stored output.
constexpr int C1M_TY = 8;
__device__ __forceinline__ void split_bf16(float v, unsigned& hi, unsigned& lo) {
    const unsigned short h = f32_to_bf16(v);
    hi = h;
    lo = f32_to_bf16(v - bf16_to_f32(h));
}
template <int NB, int TX16>      // 16-channel blocks (N = 16 * NB); 16-column blocks of a workgroup's tile
__global__ __launch_bounds__(256) void c1_s2_fwd3d_mfma_kernel(const C1Params p) {
    constexpr int tx16 = TX16;
    constexpr int N = 16 * NB, FY = 2 * C1M_TY + 1;
    extern __shared__ __attribute__((aligned(16))) float sm[];      // input tile [3][FY][FX], then statistics [4 waves][N][2]
    constexpr int FX = 32 * TX16 + 1;
    float* sst = sm + 3 * FY * FX;
    unsigned* tile = reinterpret_cast<unsigned*>(sm);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, g = lane >> 4;
    const int cz = blockIdx.z % p.cd[0], img = blockIdx.z / p.cd[0];
    const int cy0 = blockIdx.y * C1M_TY, cx0 = blockIdx.x * 16 * tx16;
    const size_t fpix = (size_t)p.fd[0] * p.fd[1] * p.fd[2];
    const size_t cpix = (size_t)p.cd[0] * p.cd[1] * p.cd[2];
    const float* xin = p.x1 + (size_t)img * fpix;
    // stage the input tile (zero outside the image): one (plane, row) per wave and step, consecutive lanes along x; all
    // loads of a thread are requested before the first LDS store (a rolled loop waits for every load in turn: 14
    // dependent round trips, two thirds of this kernel's time when it was written that way)
    {
        const int fz0 = 2 * cz - 1, fy0 = 2 * cy0 - 1, fx0 = 2 * cx0 - 1;
        constexpr int NR = (3 * FY + 3) / 4, NX = (FX + 63) / 64;       // rows per wave, 64-lane steps along x
        float tv[NR][NX];
#pragma unroll
        for (int ri = 0; ri < NR; ++ri) {
            const int row = wave + 4 * ri;
            const int z = row / FY, y = row - z * FY;
            const int fz = fz0 + z, fy = fy0 + y;
            const bool rok = row < 3 * FY && (unsigned)fz < (unsigned)p.fd[0] && (unsigned)fy < (unsigned)p.fd[1];
            const float* src = xin + ((size_t)(rok ? fz : 0) * p.fd[1] + (rok ? fy : 0)) * p.fd[2];
#pragma unroll
            for (int xi = 0; xi < NX; ++xi) {
                const int x = lane + 64 * xi, fx = fx0 + x;
                tv[ri][xi] = (rok && x < FX && (unsigned)fx < (unsigned)p.fd[2]) ? src[fx] : 0.f;
            }
        }
#pragma unroll
        for (int ri = 0; ri < NR; ++ri) {
            const int row = wave + 4 * ri;
#pragma unroll
            for (int xi = 0; xi < NX; ++xi) {
                const int x = lane + 64 * xi;
                if (row < 3 * FY && x < FX) {       // split once per input value: bf16 head in the low half, remainder in the high half
                    unsigned h, l;
                    split_bf16(tv[ri][xi], h, l);
                    tile[row * FX + x] = h | (l << 16);
                }
            }
        }
    }
    // weight fragments (row = channel lr of block nb, contraction slots 8 g .. 8 g + 7 = taps, 27 .. 31 zero) and the
    // tile offsets of the lane's eight taps
    uint4 ahi[NB], alo[NB];
    int toff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * g + j;
        const int kz = k / 9, r = k - kz * 9, ky = r / 3, kx = r - ky * 3;
        toff[j] = k < 27 ? (kz * FY + ky) * FX + kx : 0;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = nb * 16 + lr;
        unsigned h[8], l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * g + j;
            const float w = k < 27 ? ((n < p.n_w1) ? p.w[n * 27 + k] : p.w2[(n - p.n_w1) * 27 + k]) : 0.f;
            split_bf16(w, h[j], l[j]);
        }
        ahi[nb] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
        alo[nb] = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
    }
    float bv[NB][4], s1[NB][4], s2[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = nb * 16 + 4 * g + j;
            bv[nb][j] = (p.bias && ch >= p.bias_c0) ? p.bias[ch - p.bias_c0] : 0.f;
            s1[nb][j] = 0.f; s2[nb][j] = 0.f;
        }
    __syncthreads();
    const int nunit = C1M_TY * tx16;
    for (int u = wave; u < nunit; u += 4) {
        const int ry = u / tx16, xb = u - ry * tx16;
        const int cy = cy0 + ry, cx = cx0 + 16 * xb + lr;
        if (cy >= p.cd[1] || cx0 + 16 * xb >= p.cd[2]) continue;      // uniform over the wave
        const int base = (2 * ry) * FX + 2 * (16 * xb + lr);
        unsigned t8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t8[j] = tile[base + toff[j]];
        // heads of two taps into one register, remainders into another
        uint4 bhi, blo;
        bhi.x = __builtin_amdgcn_perm(t8[1], t8[0], 0x05040100u); blo.x = __builtin_amdgcn_perm(t8[1], t8[0], 0x07060302u);
        bhi.y = __builtin_amdgcn_perm(t8[3], t8[2], 0x05040100u); blo.y = __builtin_amdgcn_perm(t8[3], t8[2], 0x07060302u);
        bhi.z = __builtin_amdgcn_perm(t8[5], t8[4], 0x05040100u); blo.z = __builtin_amdgcn_perm(t8[5], t8[4], 0x07060302u);
        bhi.w = __builtin_amdgcn_perm(t8[7], t8[6], 0x05040100u); blo.w = __builtin_amdgcn_perm(t8[7], t8[6], 0x07060302u);
        const bool valid = cx < p.cd[2];
        const size_t gp = (size_t)img * cpix + ((size_t)cz * p.cd[1] + cy) * p.cd[2] + cx;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            v4f acc = v4f{0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, alo[nb]), __builtin_bit_cast(v8bf, bhi), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, ahi[nb]), __builtin_bit_cast(v8bf, blo), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, ahi[nb]), __builtin_bit_cast(v8bf, bhi), acc, 0, 0, 0);
            if (valid) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = acc[j] + bv[nb][j];
                    s1[nb][j] += acc[j];
                    s2[nb][j] += acc[j] * acc[j];
                }
                const int ch = nb * 16 + 4 * g;
                char* dst = (p.y2_out && ch >= p.n_o1) ? p.y2_out + (gp * p.y2o_pitch + p.y2o_coff + ch - p.n_o1) * 2
                                                        : p.y_out + (gp * p.y_pitch + p.y_coff + ch) * 2;
                store_n<BF16, 4>(dst, o);
            }
        }
    }
    if (p.stats) {
        // fixed-order reduction: the 16 pixel lanes of a row by butterfly, the four waves through LDS, then fp64 accumulators
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = s1[nb][j], b = s2[nb][j];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                if (lr == 0) {
                    const int ch = nb * 16 + 4 * g + j;
                    sst[(wave * N + ch) * 2] = a;
                    sst[(wave * N + ch) * 2 + 1] = b;
                }
            }
        __syncthreads();
        const int slot = (blockIdx.x + blockIdx.y) % MI_NSLOT;
        for (int i = tid; i < p.nstat * 2 && i < N * 2; i += 256) {
            const float v = ((sst[i] + sst[N * 2 + i]) + sst[2 * N * 2 + i]) + sst[3 * N * 2 + i];
            atomicAdd(&p.stats[(((size_t)img * MI_NSLOT + slot) * p.stats_c) * 2 + i], (double)v);
        }
    }
}

// ---- the same for 2-D and N = 32 (the first encoder layer with its residual branch, the data gradient of the last
// transposed convolution), as a strip walk.  Measured on the kernel above with its parts switched off: without the
// stores, the FMAs and the statistics it still takes 34 of its 58 us - the per-workgroup set-up (weights to LDS, bias,
// statistics epilogue with 64 fp64 atomics) and 27 branch-guarded gather loads in front of 4 rows of work.  Here a
// workgroup owns 64 columns x NG groups of ROWS coarse rows: set-up and epilogue once per strip, the input rows of
// group g + 1 are requested before group g is multiplied (consecutive groups share one input row: 2 ROWS new rows
// each), and the loads are unconditional (only fine row -1 and fine column -1 fall outside: the address is clamped and
// the value replaced by zero).
template <class TR, int ROWS, int NG, int N>     // N = 32 or 64 output channels: 4 or 8 lanes (channel groups of eight) per column
__global__ __launch_bounds__(256) void c1_s2_fwd32_kernel(const C1Params p) {
    constexpr int ESZ = TR::ESZ;
    constexpr int NR = 2 * ROWS + 1;
    constexpr int NGRP = N / 8, COLS = 256 / NGRP;
    static_assert(N == 32 || N == 64, "four or eight channel groups per column");
    __shared__ __attribute__((aligned(16))) float wl[9 * N];
    __shared__ float sst[2 * 16 * N];
    const int grp = threadIdx.x % NGRP, pl = threadIdx.x / NGRP;
    const int img = blockIdx.z;
    const int cx = blockIdx.x * COLS + pl;
    const bool lane_ok = cx < p.cd[2];
    const int H = p.fd[1], W = p.fd[2];
    const size_t cpix = (size_t)p.cd[1] * p.cd[2];
    const int cxs = lane_ok ? cx : 0;              // lanes past the last column read column 0 and store nothing
    const bool left = cxs == 0;
    const float* const xc = p.x1 + (size_t)img * H * W + 2 * cxs;   // fine column 2 cx of row 0
    // the loads carry no condition (clamped row, clamped left neighbour), the zeroing happens where the values are
    // first used: a guarded load is waited for on the spot, which would undo the prefetch
    const int xl = left ? 0 : -1;
    auto load_raw = [&](int fy, float* v) {
        const int fyc = min(max(fy, 0), H - 1);
        const float* r = xc + (size_t)fyc * W;
        v[0] = r[xl]; v[1] = r[0]; v[2] = r[1];
    };
    auto mask_row = [&](int fy, float* v) {
        const bool rok = (unsigned)fy < (unsigned)H;
        v[0] = (rok && !left) ? v[0] : 0.f; v[1] = rok ? v[1] : 0.f; v[2] = rok ? v[2] : 0.f;
    };
    const int cy_begin = blockIdx.y * ROWS * NG;
    float xr[NR][3], xn[2 * ROWS][3];
#pragma unroll
    for (int j = 0; j < NR; ++j) load_raw(2 * cy_begin - 1 + j, xr[j]);
    for (int i = threadIdx.x; i < N * 9; i += 256) {
        const int n = i / 9, k = i - n * 9;
        wl[k * N + n] = (n < p.n_w1) ? p.w[n * 9 + k] : p.w2[(n - p.n_w1) * 9 + k];
    }
    float bv[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = grp * 8 + e;
        bv[e] = (p.bias && ch >= p.bias_c0) ? p.bias[ch - p.bias_c0] : 0.f;
        s1[e] = 0.f; s2[e] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) mask_row(2 * cy_begin - 1 + j, xr[j]);
    __syncthreads();
    char* const dst0 = (p.y2_out && grp * 8 >= p.n_o1) ? p.y2_out + ((size_t)p.y2o_coff + grp * 8 - p.n_o1) * ESZ : p.y_out + ((size_t)p.y_coff + grp * 8) * ESZ;
    const int dpitch = ((p.y2_out && grp * 8 >= p.n_o1) ? p.y2o_pitch : p.y_pitch) * ESZ;
#pragma unroll 1
    for (int gi = 0; gi < NG; ++gi) {
        const int cy0 = cy_begin + gi * ROWS;
        if (cy0 >= p.cd[1]) break;
        asm volatile("" ::: "memory");   // the weights are read from LDS tap by tap in every group: 72 registers hoisted out of the loop cost more
        if (gi + 1 < NG) {
#pragma unroll
            for (int j = 0; j < 2 * ROWS; ++j) load_raw(2 * cy0 + 2 * ROWS + j, xn[j]);
        }
        float acc[ROWS][8];
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[r][e] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 a = *reinterpret_cast<const float4*>(wl + (ky * 3 + kx) * N + grp * 8);
                const float4 b = *reinterpret_cast<const float4*>(wl + (ky * 3 + kx) * N + grp * 8 + 4);
                const float w8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const float xv = xr[2 * r + ky][kx];
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[r][e] += xv * w8[e];
                }
                if (kx == 2) __builtin_amdgcn_sched_barrier(0);   // one kernel row of weights in registers at a time
            }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int cy = cy0 + r;
            if (lane_ok && cy < p.cd[1]) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    o[e] = acc[r][e] + bv[e];
                    s1[e] += acc[r][e];
                    s2[e] += acc[r][e] * acc[r][e];
                }
                const size_t gp = (size_t)img * cpix + (size_t)cy * p.cd[2] + cx;
                store_n<TR, 8>(dst0 + gp * dpitch, o);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) xr[0][k] = xr[2 * ROWS][k];
#pragma unroll
        for (int j = 0; j < 2 * ROWS; ++j) {
            mask_row(2 * cy0 + 2 * ROWS + j, xn[j]);
#pragma unroll
            for (int k = 0; k < 3; ++k) xr[1 + j][k] = xn[j][k];
        }
    }
    if (p.stats) {
        // NGRP channel groups per column: a 16-lane row holds 16 / NGRP columns, DPP row shifts (by 4 and 8 lanes, or by
        // 8) add them up in its last NGRP lanes; the 16 rows of the workgroup then meet in LDS ([2][16 rows][N]), fixed
        // order, then fp64 accumulators
        const int slot = (blockIdx.x + blockIdx.y) % MI_NSLOT;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (NGRP == 4) s1[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1[e]), 0x114, 0xf, 0xf, true));
            s1[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1[e]), 0x118, 0xf, 0xf, true));
            if (NGRP == 4) s2[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2[e]), 0x114, 0xf, 0xf, true));
            s2[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2[e]), 0x118, 0xf, 0xf, true));
        }
        const int row = threadIdx.x >> 4;
        if ((threadIdx.x & 15) >= 16 - NGRP) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sst[row * N + grp * 8 + e] = s1[e];
                sst[(16 + row) * N + grp * 8 + e] = s2[e];
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < p.nstat * 2; i += 256) {
            const int ch = i >> 1, k = i & 1;
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) a += sst[(k * 16 + r) * N + ch];
            atomicAdd(&p.stats[(((size_t)img * MI_NSLOT + slot) * p.stats_c) * 2 + i], (double)a);
        }
    }
}

// ---- N -> 1 stride-2 transposed convolution (k=3, pad=1, output_padding=1) ---------------
// Phase 1: every thread forms the KK tap products d[k] = <Y[c][:], w[:][k]> of one coarse pixel of the
// workgroup's patch and parks them in LDS.  Phase 2: every fine pixel sums the 1, 2, 4 (or 8) products that
// land on it: per dimension an even fine index takes tap 1 of its own coarse pixel, an odd one takes tap 2
// of its own and tap 0 of the next coarse pixel.  The patch is exactly one pixel per thread; a workgroup
// writes the fine pixels of all patch pixels but the last row / column / plane (whose "next" neighbour lies
// in the next patch), so consecutive patches overlap by one pixel and no second pass is needed.
template <class TR, int KZ>
__global__ __launch_bounds__(256) void c1_s2_T_kernel(const C1Params p) {
    constexpr int ESZ = TR::ESZ;
    constexpr int KK = KZ * 9;
    constexpr int PZ = KZ == 3 ? 4 : 1, PY = 8, PX = KZ == 3 ? 8 : 32, NP = PZ * PY * PX;   // NP = 256
    constexpr int OZ = KZ == 3 ? PZ - 1 : 1, OY = PY - 1, OX = PX - 1;                       // pixels a workgroup owns
    extern __shared__ __attribute__((aligned(16))) float sm[];  // weights [KK][N], then products [KK][NP]
    __shared__ float red[2][4];
    const int N = p.N;
    float* wl = sm;
    float* D = sm + KK * N;
    for (int i = threadIdx.x; i < N * KK; i += 256) {
        const int n = i / KK, k = i - n * KK;
        wl[k * N + n] = p.w[i];
    }
    __syncthreads();
    constexpr int TB = (KK + 15) / 16;
    const int NC = (N + 31) & ~31;
    unsigned short* wb = reinterpret_cast<unsigned short*>(D + KK * NP);   // bf16: [TB * 16 tap rows][NC], NC = N rounded up to 32
    if constexpr (TR::ESZ == 2) {
        for (int i = threadIdx.x; i < TB * 16 * NC; i += 256) {
            const int n = i % NC, kk = i / NC;
            wb[i] = (kk < KK && n < N) ? f32_to_bf16(wl[kk * N + n]) : (unsigned short)0;
        }
        __syncthreads();
    }
    const int tzn = (p.cd[0] + OZ - 1) / OZ;
    const int img = blockIdx.z / tzn;
    const int cz0 = (blockIdx.z % tzn) * OZ, cx0 = blockIdx.x * OX;
    const size_t cpix = (size_t)p.cd[0] * p.cd[1] * p.cd[2];
    const size_t fpix = (size_t)p.fd[0] * p.fd[1] * p.fd[2];
    const int tx = threadIdx.x % PX, ty = (threadIdx.x / PX) % PY, tz = threadIdx.x / (PX * PY);
    const float bias = p.bias ? p.bias[0] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    // a workgroup walks C1T_YP consecutive patches along y with the weight images prepared once
    for (int yi = 0; yi < C1T_YP; ++yi) {
    const int cy0 = (blockIdx.y * C1T_YP + yi) * OY;
    if (cy0 >= p.cd[1]) break;
    if (yi) __syncthreads();                      // the previous patch's products have been consumed
    const int cz = cz0 + tz, cy = cy0 + ty, cx = cx0 + tx;
    const bool inside = cz < p.cd[0] && cy < p.cd[1] && cx < p.cd[2];
    if constexpr (TR::ESZ == 2) {
        // bf16: the tap products are one small GEMM per 16 pixels, D[tap][pixel] = W[tap][channel] * Y[channel][pixel],
        // on the matrix cores: A = weights (16 tap rows x 32 channels from LDS, bf16, rows >= KK zero), B = the
        // pixel's channels straight from memory (lane (pixel, g) holds channels 8g..8g+7 of a 32-channel chunk)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, g = lane >> 4;
        // the four 16-pixel groups of the wave are loaded together and without a condition (a pixel outside the tensor
        // reads pixel 0 of the image and is zeroed afterwards): a load under a condition is waited for where it stands,
        // which made four dependent round trips out of one
        size_t gpx[4];
        bool okp[4];
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const int pl = wave * 64 + mf * 16 + lr;                // pixel of the patch this lane feeds
            const int qx = pl % PX, qy = (pl / PX) % PY, qz = pl / (PX * PY);
            const int pz = cz0 + qz, py = cy0 + qy, px = cx0 + qx;
            okp[mf] = pz < p.cd[0] && py < p.cd[1] && px < p.cd[2];
            gpx[mf] = (size_t)img * cpix + (okp[mf] ? ((size_t)pz * p.cd[1] + py) * p.cd[2] + px : 0);
        }
        v4f acc[4][TB];
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) acc[mf][tb] = v4f{0.f, 0.f, 0.f, 0.f};
        for (int n0 = 0; n0 < N; n0 += 32) {
            const int ch = n0 + 8 * g;
            const bool chok = ch < N;
            const int chs = chok ? ch : 0;
            const bool second = p.y2 && chs >= p.n_y1;
            const char* const sbase = second ? p.y2 + (size_t)(p.y2_coff + chs - p.n_y1) * ESZ : p.y + (size_t)(p.y_coff + chs) * ESZ;
            const size_t spitch = (size_t)(second ? p.y2_pitch : p.y_pitch) * ESZ;
            uint4 b[4];
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) b[mf] = *reinterpret_cast<const uint4*>(sbase + gpx[mf] * spitch);
#pragma unroll
            for (int mf = 0; mf < 4; ++mf)
                if (!(okp[mf] && chok)) b[mf] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) {
                const uint4 a = *reinterpret_cast<const uint4*>(wb + (size_t)(tb * 16 + lr) * NC + chs);
#pragma unroll
                for (int mf = 0; mf < 4; ++mf)
                    acc[mf][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b[mf]), acc[mf][tb], 0, 0, 0);
            }
        }
        // lane (pixel lr, g) holds taps 4g..4g+3 of each 16-tap block
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const int pl = wave * 64 + mf * 16 + lr;
#pragma unroll
            for (int tb = 0; tb < TB; ++tb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = tb * 16 + 4 * g + j;
                    if (k < KK) D[k * NP + pl] = acc[mf][tb][j];
                }
        }
    } else {
        float d[KK];
#pragma unroll
        for (int k = 0; k < KK; ++k) d[k] = 0.f;
        if (inside) {
            const size_t gp = (size_t)img * cpix + ((size_t)cz * p.cd[1] + cy) * p.cd[2] + cx;
            const char* src = p.y + (gp * p.y_pitch + p.y_coff) * ESZ;
            const char* src2 = p.y2 ? p.y2 + (gp * p.y2_pitch + p.y2_coff) * ESZ : nullptr;
            for (int n0 = 0; n0 < N; n0 += 8) {
                float v[8];
                load_n<TR, 8>((src2 && n0 >= p.n_y1) ? src2 + (n0 - p.n_y1) * ESZ : src + n0 * ESZ, v);
#pragma unroll
                for (int k = 0; k < KK; ++k) {
                    const float4 a = *reinterpret_cast<const float4*>(wl + k * N + n0);
                    const float4 b = *reinterpret_cast<const float4*>(wl + k * N + n0 + 4);
                    d[k] += v[0] * a.x + v[1] * a.y + v[2] * a.z + v[3] * a.w + v[4] * b.x + v[5] * b.y + v[6] * b.z + v[7] * b.w;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < KK; ++k) D[k * NP + threadIdx.x] = d[k];
    }
    __syncthreads();
    const bool owner = inside && tx < OX && ty < OY && (KZ == 1 || tz < OZ);
    if (owner) {
        float* out = p.x1_out + (size_t)img * fpix;
#pragma unroll
        for (int pz = 0; pz < (KZ == 3 ? 2 : 1); ++pz)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                float o[2];
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    float acc = 0.f;
#pragma unroll
                    for (int az = 0; az <= ((KZ == 3) ? pz : 0); ++az)
#pragma unroll
                        for (int ay = 0; ay <= py; ++ay)
#pragma unroll
                            for (int ax = 0; ax <= px; ++ax) {
                                // parity 0: (tap 1, own pixel); parity 1: a = 0 -> (tap 2, own), a = 1 -> (tap 0, next)
                                const int kz = (KZ == 3) ? (pz == 0 ? 1 : (az == 0 ? 2 : 0)) : 0;
                                const int ky = py == 0 ? 1 : (ay == 0 ? 2 : 0);
                                const int kx = px == 0 ? 1 : (ax == 0 ? 2 : 0);
                                const int k = (kz * 3 + ky) * 3 + kx;
                                acc += D[k * NP + ((tz + az) * PY + ty + ay) * PX + tx + ax];
                            }
                    o[px] = acc;
                    s1 += acc; s2 += acc * acc;
                }
                const size_t fo = ((size_t)(2 * cz + pz) * p.fd[1] + 2 * cy + py) * p.fd[2] + 2 * cx;
                *reinterpret_cast<float2*>(out + fo) = make_float2(o[0] + bias, o[1] + bias);
            }
    }
    }   // patches
    if (p.stats) {
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red[0][w] = s1; red[1][w] = s2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int slot = (blockIdx.x + blockIdx.y) % MI_NSLOT;
            atomicAdd(&p.stats[((size_t)img * MI_NSLOT + slot) * 2], (double)(((red[0][0] + red[0][1]) + red[0][2]) + red[0][3]));
            atomicAdd(&p.stats[((size_t)img * MI_NSLOT + slot) * 2 + 1], (double)(((red[1][0] + red[1][1]) + red[1][2]) + red[1][3]));
        }
    }
}

// ---- weight gradient of both: partial[slab][k][n] = sum_c Y[c][n] * x1[2c-1+k] ;
//      partial_bias[slab][n] = sum_c Y[c][n].  blockIdx.y = kz plane in 3-D.
// Thread = (coarse column, group of 8 channels); blockIdx.x = (column block, row block), the row block strides
// over all (image, z, y) rows so that no per-lane division is needed.  Channels >= n_y1 come from y2.
// QUAD (N = 32: the four channel groups of a pixel are the four lanes of a quad): lane ky of the quad loads the three
// image values of stencil row ky and the quad exchanges them with DPP broadcasts, so a thread issues 3 instead
// of 9 scattered 4-byte loads per row: these loads, not the bytes, bound the kernel.
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), K * 0x55, 0xf, 0xf, false));
}
template <class TR, bool QUAD>
__global__ __launch_bounds__(256) void c1_s2_wgrad_kernel(const C1Params p) {
    constexpr int ESZ = TR::ESZ;
    constexpr int CG = C1_WGRAD_CG;   // channels per thread: few registers, many waves per SIMD
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [4 waves (16 rows: QUAD)][N][10]
    const int KZ = p.ndim == 3 ? 3 : 1;
    const int KK = KZ * 9;
    const int kz = blockIdx.y;
    const int G = p.N / CG;
    for (int i = threadIdx.x; i < (QUAD ? 16 : 4) * p.N * 10; i += 256) sm[i] = 0.f;
    __syncthreads();
    const size_t cpix = (size_t)p.cd[0] * p.cd[1] * p.cd[2];
    const size_t fpix = (size_t)p.fd[0] * p.fd[1] * p.fd[2];
    float acc[CG][9], accb[CG];
#pragma unroll
    for (int e = 0; e < CG; ++e) {
        accb[e] = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[e][k] = 0.f;
    }
    const int grp = threadIdx.x % G, pl = threadIdx.x / G;
    const int ppb = 256 / G;
    const int xblocks = (p.cd[2] + ppb - 1) / ppb;
    const int xb = blockIdx.x % xblocks, rb = blockIdx.x / xblocks, nrb = gridDim.x / xblocks;
    const int cx = xb * ppb + pl;
    const bool lane_ok = pl < ppb && cx < p.cd[2];
    const bool second = grp * CG >= p.n_y1;
    const char* ysrc = second ? p.y2 : p.y;
    const int ypitch = second ? p.y2_pitch : p.y_pitch;
    const int ych = second ? p.y2_coff + grp * CG - p.n_y1 : p.y_coff + grp * CG;
    const int nrows = p.nb * p.cd[0] * p.cd[1];
    // A workgroup walks its rows C1W_RU at a time: all loads of a step (one channel vector and nine image values
    // per row) are issued before the first use, so a thread has C1W_RU rows in flight instead of one and the
    // walk takes a quarter of the dependent round trips.
    constexpr int RV = CG * ESZ / 16;          // 16-byte vectors per channel group
    if (rb < nrb && lane_ok)   // workgroups beyond the last full row block (grid not a multiple of xblocks) only write zeros
    for (int row0 = rb; row0 < nrows; row0 += nrb * C1W_RU) {
        uint4 yr[C1W_RU][RV];
        float xv[C1W_RU][QUAD ? 3 : 9];
        bool rowok[C1W_RU], zok[C1W_RU];
#pragma unroll
        for (int u = 0; u < C1W_RU; ++u) {
            const int row = row0 + u * nrb;
            rowok[u] = row < nrows; zok[u] = false;
            if (!rowok[u]) continue;
            const int cy = row % p.cd[1]; const int r2 = row / p.cd[1];
            const int cz = r2 % p.cd[0]; const int img = r2 / p.cd[0];
            const char* ysrc_p = ysrc + (((size_t)row * p.cd[2] + cx) * ypitch + ych) * ESZ;
#pragma unroll
            for (int q = 0; q < RV; ++q) yr[u][q] = *reinterpret_cast<const uint4*>(ysrc_p + q * 16);
            const int fz = (p.ndim == 3) ? 2 * cz - 1 + kz : 0;
            zok[u] = (unsigned)fz < (unsigned)p.fd[0];
            if (!zok[u]) continue;
            const float* xin = p.x1 + (size_t)img * fpix + (size_t)fz * p.fd[1] * p.fd[2];
            if (QUAD) {
                const int fy = 2 * cy - 1 + grp;
                const bool rok = grp < 3 && (unsigned)fy < (unsigned)p.fd[1];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int fx = 2 * cx - 1 + kx;
                    xv[u][kx] = (rok && (unsigned)fx < (unsigned)p.fd[2]) ? xin[(size_t)fy * p.fd[2] + fx] : 0.f;
                }
            } else {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int fy = 2 * cy - 1 + ky;
                    const bool rok = (unsigned)fy < (unsigned)p.fd[1];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int fx = 2 * cx - 1 + kx;
                        xv[u][ky * 3 + kx] = (rok && (unsigned)fx < (unsigned)p.fd[2]) ? xin[(size_t)fy * p.fd[2] + fx] : 0.f;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < C1W_RU; ++u) {
            if (!rowok[u]) continue;
            float v[CG];
#pragma unroll
            for (int q = 0; q < RV; ++q) unpack_vec<TR>(yr[u][q], v + q * TR::VEC);
#pragma unroll
            for (int e = 0; e < CG; ++e) accb[e] += v[e];
            if (!zok[u]) continue;
            float x9[9];
            if (QUAD) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    x9[kx] = quad_bcast<0>(xv[u][kx]); x9[3 + kx] = quad_bcast<1>(xv[u][kx]); x9[6 + kx] = quad_bcast<2>(xv[u][kx]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 9; ++k) x9[k] = xv[u][k];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int e = 0; e < CG; ++e) acc[e][k] += v[e] * x9[k];
        }
    }
    (void)cpix;
    // lanes that share a channel group: xor-reduce over lane bits >= log2(G) when G divides the wave
    const bool pow2 = (64 % G) == 0;
    constexpr int NIMG = QUAD ? 16 : 4;      // LDS images summed at the end: one per 16-lane row (QUAD) or per wave
    if (QUAD) {
        // four channel groups per pixel: the partner lanes inside a 16-lane row are 4 and 8 away, which two
        // DPP row shifts add up in lanes 12..15; every row then writes its own LDS image
#pragma unroll
        for (int e = 0; e < CG; ++e) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                float a = acc[e][k];
                a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x114, 0xf, 0xf, true));
                a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x118, 0xf, 0xf, true));
                acc[e][k] = a;
            }
            float b = accb[e];
            b += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b), 0x114, 0xf, 0xf, true));
            b += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b), 0x118, 0xf, 0xf, true));
            accb[e] = b;
        }
    } else if (pow2) {
#pragma unroll
        for (int e = 0; e < CG; ++e) {
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
    float* mine = sm + (QUAD ? (threadIdx.x >> 4) : (threadIdx.x >> 6)) * p.N * 10;
    if (QUAD) {
        if ((threadIdx.x & 15) >= 12) {
#pragma unroll
            for (int e = 0; e < CG; ++e) {
#pragma unroll
                for (int k = 0; k < 9; ++k) mine[(grp * CG + e) * 10 + k] = acc[e][k];
                mine[(grp * CG + e) * 10 + 9] = accb[e];
            }
        }
    } else if (pow2) {
        if ((threadIdx.x & 63) < G) {
#pragma unroll
            for (int e = 0; e < CG; ++e) {
#pragma unroll
                for (int k = 0; k < 9; ++k) mine[(grp * CG + e) * 10 + k] = acc[e][k];
                mine[(grp * CG + e) * 10 + 9] = accb[e];
            }
        }
    } else if (pl < ppb) {
#pragma unroll
        for (int e = 0; e < CG; ++e) {
#pragma unroll
            for (int k = 0; k < 9; ++k) atomicAdd(&mine[(grp * CG + e) * 10 + k], acc[e][k]);
            atomicAdd(&mine[(grp * CG + e) * 10 + 9], accb[e]);
        }
    }
    __syncthreads();
    const size_t slab = blockIdx.x;
    const int wstride = p.N * 10;
    for (int i = threadIdx.x; i < p.N * 9; i += 256) {
        const int n = i / 9, k = i - n * 9;
        const int o = n * 10 + k;
        float a = sm[o];
#pragma unroll
        for (int im = 1; im < NIMG; ++im) a += sm[im * wstride + o];
        p.partial[(slab * KK + kz * 9 + k) * p.N + n] = a;
    }
    if (kz == 0)
        for (int n = threadIdx.x; n < p.N; n += 256) {
            const int o = n * 10 + 9;
            float a = sm[o];
#pragma unroll
            for (int im = 1; im < NIMG; ++im) a += sm[im * wstride + o];
            p.partial_bias[slab * p.N + n] = a;
        }
}

// ---- the same weight gradient on the matrix cores (bf16 storage; 3-D: N = 16 or 32, 2-D (KZ = 1): N = 16, 32 or 64) ------
// partial[slab][k][n] = sum_p Y[p][n] * x[p -> tap k] is a matrix product with the pixels as contraction: a wave takes
// 32 consecutive coarse pixels of a row per MFMA step, rows = channels (Y loaded as 16-byte vectors and transposed
// through LDS: two-byte loads of single channels cost a third of the kernel), columns = the 27 taps of all three kz
// planes, a column of ones (the bias gradient) and four empty ones.  A workgroup's tile is four coarse rows of up to 64
// pixels, one row per wave; its image values (3 planes x 9 rows) are staged in LDS once and gathered per tap from there.  The image values are split into a bf16 head and
// remainder (two products, fp32 accuracy as in c1_s2_fwd3d_mfma_kernel); Y is bf16 as stored.  The vector-ALU kernel
// above runs once per kz plane, reads Y three times and spends 864 FMAs per pixel and channel group.
template <int NB, int TX32, int KZ>       // 16-channel blocks (N = 16 NB); 32-pixel units along x in a workgroup's tile (1 or 2); kernel planes (3, or 1 in 2-D)
__global__ __launch_bounds__(256) void c1_s2_wgrad3d_mfma_kernel(const C1Params p) {
    constexpr int KK = 9 * KZ, NTB = (KK + 1 + 15) / 16;       // taps; 16-column blocks holding the taps and the column of ones
    constexpr int N = 16 * NB, TY = 4, FY = 2 * TY + 1, FX = 64 * TX32 + 1;
    constexpr int PXP = 32 * TX32 + 8;               // pixels per channel row of the transposed Y (144 or 80 bytes: 16-byte aligned)
    constexpr int TILE_B = KZ * FY * FX * 4, YSH_B = 4 * N * PXP * 2, RED_B = 4 * 16 * NTB * (N + 1) * 4;
    __shared__ __align__(16) unsigned char smem[(TILE_B + 15) / 16 * 16 + YSH_B > RED_B ? (TILE_B + 15) / 16 * 16 + YSH_B : RED_B];
    unsigned* tile = reinterpret_cast<unsigned*>(smem);                       // image values of a tile: bf16 head | remainder << 16
    unsigned short* ysh = reinterpret_cast<unsigned short*>(smem + (TILE_B + 15) / 16 * 16);   // [wave][channel][pixel]
    float (*red)[16 * NTB][N + 1] = reinterpret_cast<float (*)[16 * NTB][N + 1]>(smem);   // after the last tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, g = lane >> 4;
    const size_t fpix = (size_t)p.fd[0] * p.fd[1] * p.fd[2];
    const int ntx = (p.cd[2] + 32 * TX32 - 1) / (32 * TX32), nty = (p.cd[1] + TY - 1) / TY;
    const long long ntile = (long long)p.nb * p.cd[0] * nty * ntx;
    // the lane's tap per column block: tile offset of (kz, ky, kx), or the ones column (KK), or nothing
    int toff[NTB], tkind[NTB];
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) {
        const int k = tb * 16 + lr;
        tkind[tb] = k < KK ? 0 : k == KK ? 1 : 2;
        const int kz = k / 9, r = k - kz * 9, ky = r / 3, kx = r - ky * 3;
        toff[tb] = k < KK ? (kz * FY + ky) * FX + kx : 0;
    }
    v4f acc[NB][NTB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) acc[nb][tb] = v4f{0.f, 0.f, 0.f, 0.f};

    // A tile's global loads are requested one tile ahead (registers), so that their latency runs beside the previous
    // tile's LDS gathers and MFMAs; every load of a thread is in flight before its first LDS store.
    constexpr int NR = (KZ * FY + 3) / 4, NX = (FX + 63) / 64, NV = 32 * TX32 * N / 8 / 64;
    float tv[NR][NX];
    uint4 yraw[NV];
    auto request = [&](long long t) {
        const int txi = (int)(t % ntx); long long r1 = t / ntx;
        const int tyi = (int)(r1 % nty); r1 /= nty;
        const int cz = (int)(r1 % p.cd[0]), img = (int)(r1 / p.cd[0]);
        const int cy0 = tyi * TY, cxt = txi * 32 * TX32;
        const float* xin = p.x1 + (size_t)img * fpix;
        const int fz0 = KZ == 3 ? 2 * cz - 1 : 0, fy0 = 2 * cy0 - 1, fx0 = 2 * cxt - 1;
#pragma unroll
        for (int ri = 0; ri < NR; ++ri) {
            const int row = wave + 4 * ri;
            const int z = row / FY, y = row - z * FY;
            const int fz = fz0 + z, fy = fy0 + y;
            const bool rok = row < KZ * FY && (unsigned)fz < (unsigned)p.fd[0] && (unsigned)fy < (unsigned)p.fd[1];
            const float* src = xin + ((size_t)(rok ? fz : 0) * p.fd[1] + (rok ? fy : 0)) * p.fd[2];
#pragma unroll
            for (int xi = 0; xi < NX; ++xi) {
                const int x = lane + 64 * xi, fx = fx0 + x;
                tv[ri][xi] = (rok && x < FX && (unsigned)fx < (unsigned)p.fd[2]) ? src[fx] : 0.f;   // zero outside the image
            }
        }
        // Y of this wave's row as 16-byte vectors (eight channels of a pixel)
        const int cy = cy0 + wave;
        const bool rowok = cy < p.cd[1];
        const size_t grow = ((size_t)img * p.cd[0] + cz) * p.cd[1] + cy;
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int v = lane + 64 * r, px = v / (N / 8), ch = (v % (N / 8)) * 8;
            const bool second = ch >= p.n_y1;
            const unsigned short* ysrc = reinterpret_cast<const unsigned short*>(second ? p.y2 : p.y);
            const int ypitch = second ? p.y2_pitch : p.y_pitch;
            const int ych = second ? p.y2_coff + ch - p.n_y1 : p.y_coff + ch;
            yraw[r] = (rowok && cxt + px < p.cd[2]) ? *reinterpret_cast<const uint4*>(ysrc + (grow * p.cd[2] + cxt + px) * ypitch + ych)
                                                    : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    if ((long long)blockIdx.x < ntile) request(blockIdx.x);
    for (long long t = blockIdx.x; t < ntile; t += gridDim.x) {
        __syncthreads();                                  // the previous tile has been consumed
#pragma unroll
        for (int ri = 0; ri < NR; ++ri) {
            const int row = wave + 4 * ri;
#pragma unroll
            for (int xi = 0; xi < NX; ++xi) {
                const int x = lane + 64 * xi;
                if (row < KZ * FY && x < FX) {
                    unsigned h, l;
                    split_bf16(tv[ri][xi], h, l);
                    tile[row * FX + x] = h | (l << 16);
                }
            }
        }
        // Y scattered into LDS as [channel][pixel]: the MFMA's row operand is eight pixels of one channel per lane
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int v = lane + 64 * r, px = v / (N / 8), ch = (v % (N / 8)) * 8;
            unsigned short* dst = ysh + ((size_t)wave * N + ch) * PXP + px;
            dst[0 * PXP] = (unsigned short)yraw[r].x; dst[1 * PXP] = (unsigned short)(yraw[r].x >> 16);
            dst[2 * PXP] = (unsigned short)yraw[r].y; dst[3 * PXP] = (unsigned short)(yraw[r].y >> 16);
            dst[4 * PXP] = (unsigned short)yraw[r].z; dst[5 * PXP] = (unsigned short)(yraw[r].z >> 16);
            dst[6 * PXP] = (unsigned short)yraw[r].w; dst[7 * PXP] = (unsigned short)(yraw[r].w >> 16);
        }
        if (t + gridDim.x < ntile) request(t + gridDim.x);
        __syncthreads();
#pragma unroll
        for (int xb = 0; xb < TX32; ++xb) {
            uint4 afr[NB], bhi[NTB], blo[NTB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                afr[nb] = *reinterpret_cast<const uint4*>(ysh + ((size_t)wave * N + nb * 16 + lr) * PXP + xb * 32 + 8 * g);
            const int base = (2 * wave) * FX + 2 * (xb * 32 + 8 * g);
#pragma unroll
            for (int tb = 0; tb < NTB; ++tb) {
                unsigned t8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    t8[j] = tile[base + toff[tb] + 2 * j];
                    if (tkind[tb] == 1) t8[j] = 0x3f80u;          // the column of ones (Y is zero where the pixel is not)
                    if (tkind[tb] == 2) t8[j] = 0u;
                }
                bhi[tb].x = __builtin_amdgcn_perm(t8[1], t8[0], 0x05040100u); blo[tb].x = __builtin_amdgcn_perm(t8[1], t8[0], 0x07060302u);
                bhi[tb].y = __builtin_amdgcn_perm(t8[3], t8[2], 0x05040100u); blo[tb].y = __builtin_amdgcn_perm(t8[3], t8[2], 0x07060302u);
                bhi[tb].z = __builtin_amdgcn_perm(t8[5], t8[4], 0x05040100u); blo[tb].z = __builtin_amdgcn_perm(t8[5], t8[4], 0x07060302u);
                bhi[tb].w = __builtin_amdgcn_perm(t8[7], t8[6], 0x05040100u); blo[tb].w = __builtin_amdgcn_perm(t8[7], t8[6], 0x07060302u);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int tb = 0; tb < NTB; ++tb) {
                    acc[nb][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, afr[nb]), __builtin_bit_cast(v8bf, blo[tb]), acc[nb][tb], 0, 0, 0);
                    acc[nb][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, afr[nb]), __builtin_bit_cast(v8bf, bhi[tb]), acc[nb][tb], 0, 0, 0);
                }
        }
    }
    // accumulator (nb, tb)[j] of lane (lr, g) is channel nb * 16 + 4 g + j, tap tb * 16 + lr; the four waves meet in LDS
    __syncthreads();                                      // red[] lies over the tile
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[wave][tb * 16 + lr][nb * 16 + 4 * g + j] = acc[nb][tb][j];
    __syncthreads();
    const size_t slab = blockIdx.x;
    for (int i = tid; i < (KK + 1) * N; i += 256) {
        const int k = i / N, n = i - k * N;
        const float a = ((red[0][k][n] + red[1][k][n]) + red[2][k][n]) + red[3][k][n];
        if (k < KK) p.partial[(slab * KK + k) * p.N + n] = a;
        else p.partial_bias[slab * p.N + n] = a;
    }
}

// ---- one-channel 3^d stencil: out = sum_k w[k] * in[p + k - 1] + bias + addend ------------
// Thread = four consecutive pixels of a row (one aligned 16-byte load per stencil row plus the two
// neighbours); needs a row length that is a multiple of 4.
template <int KZ>
__global__ __launch_bounds__(256) void stencil_kernel(const StencilParams p) {
    constexpr int KK = KZ * 9;
    const int qrow = p.d[2] >> 2;
    const long long nq = (long long)p.d[0] * p.d[1] * qrow;
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const int img = blockIdx.y;
    const size_t npix = (size_t)p.d[0] * p.d[1] * p.d[2];
    const int x0 = (int)(q % qrow) * 4; const int r = (int)(q / qrow);
    const int y = r % p.d[1]; const int z = r / p.d[1];
    const float* in = p.in + (size_t)img * npix;
    float w[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) w[k] = p.w[p.flip ? KK - 1 - k : k];
    const float b = p.bias ? p.bias[0] : 0.f;
    float acc[4] = {b, b, b, b};
#pragma unroll
    for (int kz = 0; kz < KZ; ++kz) {
        const int zz = (KZ == 3) ? z - 1 + kz : 0;
        if ((unsigned)zz >= (unsigned)p.d[0]) continue;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y - 1 + ky;
            if ((unsigned)yy >= (unsigned)p.d[1]) continue;
            const float* row = in + ((size_t)zz * p.d[1] + yy) * p.d[2];
            const float4 c = *reinterpret_cast<const float4*>(row + x0);
            const float v[6] = {x0 > 0 ? row[x0 - 1] : 0.f, c.x, c.y, c.z, c.w, x0 + 4 < p.d[2] ? row[x0 + 4] : 0.f};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += w[(kz * 3 + ky) * 3 + kx] * v[j + kx];
        }
    }
    const size_t o = (size_t)img * npix + ((size_t)z * p.d[1] + y) * p.d[2] + x0;
    if (p.addend) {
        const float4 a = *reinterpret_cast<const float4*>(p.addend + o);
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
    }
    *reinterpret_cast<float4*>(p.out + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// 3-D variant through LDS: a workgroup owns S3_TZ x S3_TY x S3_TX outputs (four consecutive x per thread) and stages
// their input block with a one-pixel halo (in x: one aligned 16-byte group on either side) once; the 27 taps then read
// LDS.  The per-thread form above fetches every input row nine times through L2 (three planes of a 96^2 image exceed
// L1): 27 x the tensor; here a block is fetched ~2.3 x.
constexpr int S3_TZ = 4, S3_TY = 8, S3_TX = 32;
__global__ __launch_bounds__(256) void stencil3d_tile_kernel(const StencilParams p) {
    constexpr int PZ = S3_TZ + 2, PY = S3_TY + 2, PQ = S3_TX / 4 + 2;      // planes, rows, 16-byte groups per row
    __shared__ __attribute__((aligned(16))) float tile[PZ][PY][PQ * 4];
    const int ntx = (p.d[2] + S3_TX - 1) / S3_TX;
    const int x0 = (blockIdx.x % ntx) * S3_TX, y0 = (blockIdx.x / ntx) * S3_TY, z0 = blockIdx.y * S3_TZ;
    const int img = blockIdx.z;
    const size_t npix = (size_t)p.d[0] * p.d[1] * p.d[2];
    const float* in = p.in + (size_t)img * npix;
    {
        // all of a thread's tile loads are requested before its first LDS store (one round trip instead of three)
        constexpr int NIT = (PZ * PY * PQ + 255) / 256;
        float4 tv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = threadIdx.x + 256 * it;
            const int q = i % PQ, py = (i / PQ) % PY, pz = i / (PQ * PY);
            const int z = z0 - 1 + pz, y = y0 - 1 + py, x = x0 - 4 + 4 * q;
            tv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < PZ * PY * PQ && (unsigned)z < (unsigned)p.d[0] && (unsigned)y < (unsigned)p.d[1] && (unsigned)x < (unsigned)p.d[2])
                tv[it] = *reinterpret_cast<const float4*>(in + ((size_t)z * p.d[1] + y) * p.d[2] + x);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = threadIdx.x + 256 * it;
            const int q = i % PQ, py = (i / PQ) % PY, pz = i / (PQ * PY);
            if (i < PZ * PY * PQ) *reinterpret_cast<float4*>(&tile[pz][py][4 * q]) = tv[it];
        }
    }
    float w[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) w[k] = p.w[p.flip ? 26 - k : k];
    const float b = p.bias ? p.bias[0] : 0.f;
    const int tq = threadIdx.x & 7, ty = (threadIdx.x >> 3) & 7, tz = threadIdx.x >> 6;
    const int z = z0 + tz, y = y0 + ty, x = x0 + 4 * tq;
    const bool inside = z < p.d[0] && y < p.d[1] && x < p.d[2];
    const size_t o = (size_t)img * npix + (inside ? ((size_t)z * p.d[1] + y) * p.d[2] + x : 0);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.addend) a = *reinterpret_cast<const float4*>(p.addend + o);     // requested with the block, used at the end
    __syncthreads();
    float acc[4] = {b, b, b, b};
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float* row = &tile[tz + kz][ty + ky][4 * tq + 4];
            const float4 c = *reinterpret_cast<const float4*>(row);
            const float v[6] = {row[-1], c.x, c.y, c.z, c.w, row[4]};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += w[(kz * 3 + ky) * 3 + kx] * v[j + kx];
        }
    if (inside) *reinterpret_cast<float4*>(p.out + o) = make_float4(acc[0] + a.x, acc[1] + a.y, acc[2] + a.z, acc[3] + a.w);
}

// 2-D variant: a thread owns four consecutive pixels of ST_R consecutive rows.  All ST_R + 2 input rows are
// requested before the first use (one 16-byte load plus the two neighbours per row), so each input row is
// fetched (ST_R + 2) / ST_R times instead of three times and a thread has its whole working set in flight.
constexpr int ST_R = 8;
// NRM: the input is a raw one-channel convolution output that is normalised, dropped and activated as it is loaded
// (same arithmetic as norm1_fwd_kernel: a 16-byte group of four pixels is one dropout group; the two halo pixels of
// a row belong to the neighbouring groups); the thread's own pixels are the identity addend and go to act_out.
template <bool NRM>
__global__ __launch_bounds__(256) void stencil2d_strip_kernel(const StencilParams p) {
    const int H = p.d[1], W = p.d[2];
    const int qrow = W >> 2;
    const int nstrip = (H + ST_R - 1) / ST_R;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= qrow * nstrip) return;
    const int strip = t / qrow;
    const int x0 = (t - strip * qrow) * 4, y0 = strip * ST_R;
    const size_t ibase = (size_t)blockIdx.y * H * W;
    const float* in = p.in + ibase;
    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = p.w[p.flip ? 8 - k : k];
    const float b = p.bias ? p.bias[0] : 0.f;
    float mean = 0.f, rstd = 1.f, alpha = 0.f, scale = 1.f;
    DropKeys dk = {0u, 0u, 0u, 0u};
    bool drop = false;
    if (NRM) {
        c1_moments(p.nrm_stats, (int)blockIdx.y, H * W, p.nrm_eps, mean, rstd);
        alpha = *p.nrm_alpha;
        dk = dropout_keys(p.nrm_seed, p.nrm_layer, blockIdx.y, p.nrm_thresh);
        scale = p.nrm_scale;
        drop = p.nrm_thresh != 0;
    }
    auto activate = [&](float x, bool keep) {
        float d = (x - mean) * rstd;
        if (drop) d = keep ? d * scale : 0.f;
        return d > 0.f ? d : alpha * d;
    };
    float v[ST_R + 2][6];
#pragma unroll
    for (int r = 0; r < ST_R + 2; ++r) {
        const int yy = y0 - 1 + r;
        if ((unsigned)yy < (unsigned)H) {
            const float* row = in + (size_t)yy * W;
            const float4 c = *reinterpret_cast<const float4*>(row + x0);
            v[r][0] = x0 > 0 ? row[x0 - 1] : 0.f;
            v[r][1] = c.x; v[r][2] = c.y; v[r][3] = c.z; v[r][4] = c.w;
            v[r][5] = x0 + 4 < W ? row[x0 + 4] : 0.f;
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) v[r][j] = 0.f;
        }
    }
    if (NRM) {
#pragma unroll
        for (int r = 0; r < ST_R + 2; ++r) {
            const int yy = y0 - 1 + r;
            if ((unsigned)yy < (unsigned)H) {
                const unsigned gc = (unsigned)(yy * W + x0) >> 2;        // dropout group of the four centre pixels
                bool kc[4] = {true, true, true, true}, kl[4] = {true, true, true, true}, kr[4] = {true, true, true, true};
                if (drop) {
                    dropout_keep4(dk, gc, kc);
                    if (x0 > 0) dropout_keep4(dk, gc - 1u, kl);
                    if (x0 + 4 < W) dropout_keep4(dk, gc + 1u, kr);
                }
                if (x0 > 0) v[r][0] = activate(v[r][0], kl[3]);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[r][1 + j] = activate(v[r][1 + j], kc[j]);
                if (x0 + 4 < W) v[r][5] = activate(v[r][5], kr[0]);
            }
        }
        if (p.act_out) {
#pragma unroll
            for (int r = 0; r < ST_R; ++r)
                if (y0 + r < H)
                    *reinterpret_cast<float4*>(p.act_out + ibase + (size_t)(y0 + r) * W + x0) = make_float4(v[r + 1][1], v[r + 1][2], v[r + 1][3], v[r + 1][4]);
        }
    }
    float4 add[ST_R];
    if (!NRM && p.addend) {
#pragma unroll
        for (int r = 0; r < ST_R; ++r)
            if (y0 + r < H) add[r] = *reinterpret_cast<const float4*>(p.addend + ibase + (size_t)(y0 + r) * W + x0);
    }
#pragma unroll
    for (int r = 0; r < ST_R; ++r) {
        if (y0 + r >= H) break;
        float acc[4] = {b, b, b, b};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += w[ky * 3 + kx] * v[r + ky][j + kx];
        if (NRM) { acc[0] += v[r + 1][1]; acc[1] += v[r + 1][2]; acc[2] += v[r + 1][3]; acc[3] += v[r + 1][4]; }
        else if (p.addend) { acc[0] += add[r].x; acc[1] += add[r].y; acc[2] += add[r].z; acc[3] += add[r].w; }
        *reinterpret_cast<float4*>(p.out + ibase + (size_t)(y0 + r) * W + x0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// any row length: one pixel per thread
__global__ __launch_bounds__(256) void stencil_scalar_kernel(const StencilParams p) {
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
// QUAD = 4: four consecutive pixels per thread and iteration (row length a multiple of 4); QUAD = 1: any length.
// All KZ planes of taps in one pass: dout is read once, the nine input rows of a pixel group are independent loads.
template <int QUAD, int KZ, bool I64>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void stencil_wgrad_kernel(const StencilParams p) {   // four waves per SIMD: the 1024 workgroups of a launch are resident at once
    constexpr int KK = KZ * 9;
    __shared__ float red[KK + 1][4];
    const int qrow = p.d[2] / QUAD;
    const long long rows = (long long)p.nb * p.d[0] * p.d[1];
    const long long total = rows * qrow;
    const int lane = threadIdx.x & 63;
    float acc[KK + 1];
#pragma unroll
    for (int k = 0; k <= KK; ++k) acc[k] = 0.f;
    typedef typename std::conditional<I64, long long, unsigned>::type idx_t;
    for (idx_t it = (idx_t)blockIdx.x * 256 + threadIdx.x; it < (idx_t)total; it += (idx_t)gridDim.x * 256) {
        const idx_t urow = it / (idx_t)qrow;
        const int x0 = (int)(it - urow * (idx_t)qrow) * QUAD;
        const idx_t r2 = urow / (idx_t)p.d[1];
        const int y = (int)(urow - r2 * (idx_t)p.d[1]);
        const int z = (int)(r2 % (idx_t)p.d[0]);
        const long long row = (long long)urow;
        float g[QUAD];
        if constexpr (QUAD == 4) {
            const float4 t = *reinterpret_cast<const float4*>(p.dout + row * p.d[2] + x0);
            g[0] = t.x; g[1] = t.y; g[2] = t.z; g[3] = t.w;
        } else {
            g[0] = p.dout[row * p.d[2] + x0];
        }
#pragma unroll
        for (int j = 0; j < QUAD; ++j) acc[KK] += g[j];
        float v[KZ][3][QUAD + 2];
#pragma unroll
        for (int kz = 0; kz < KZ; ++kz) {
            const int zz = KZ == 3 ? z - 1 + kz : 0;
            const bool zok = (unsigned)zz < (unsigned)p.d[0];
            const float* in = p.in + (row + (long long)((zok ? zz : z) - z) * p.d[1] - y) * p.d[2];   // row 0 of plane zz of this image
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = y - 1 + ky;
                const bool ok = zok && (unsigned)yy < (unsigned)p.d[1];
                const float* rp = in + (size_t)(ok ? yy : y) * p.d[2];
                if constexpr (QUAD == 4) {
                    // the two neighbours along x are the next lanes' values (consecutive lanes are consecutive groups
                    // of a row; at a row's ends the neighbour is outside the image); only the wave's end lanes load them
                    const float4 c = ok ? *reinterpret_cast<const float4*>(rp + x0) : make_float4(0.f, 0.f, 0.f, 0.f);
                    v[kz][ky][1] = c.x; v[kz][ky][2] = c.y; v[kz][ky][3] = c.z; v[kz][ky][4] = c.w;
                    float left = __shfl_up(c.w, 1, 64), right = __shfl_down(c.x, 1, 64);
                    if (lane == 0 && ok && x0 > 0) left = rp[x0 - 1];
                    if (lane == 63 && ok && x0 + QUAD < p.d[2]) right = rp[x0 + QUAD];
                    v[kz][ky][0] = x0 > 0 ? left : 0.f;
                    v[kz][ky][QUAD + 1] = x0 + QUAD < p.d[2] ? right : 0.f;
                } else {
                    v[kz][ky][0] = (ok && x0 > 0) ? rp[x0 - 1] : 0.f;
                    v[kz][ky][1] = ok ? rp[x0] : 0.f;
                    v[kz][ky][QUAD + 1] = (ok && x0 + QUAD < p.d[2]) ? rp[x0 + QUAD] : 0.f;
                }
            }
        }
#pragma unroll
        for (int kz = 0; kz < KZ; ++kz)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int j = 0; j < QUAD; ++j) acc[(kz * 3 + ky) * 3 + kx] += v[kz][ky][j + kx] * g[j];
    }
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k <= KK; ++k) {
        const float a = wave_sum(acc[k]);
        if ((threadIdx.x & 63) == 0) red[k][w] = a;
    }
    __syncthreads();
    if (threadIdx.x <= KK) {
        const float a = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        if (threadIdx.x < KK) p.partial[(size_t)blockIdx.x * KK + threadIdx.x] = a;
        else p.partial_bias[blockIdx.x] = a;
    }
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string(what) + ": " + hipGetErrorString(e)); return -2; }
    return 0;
}

template <class K>
int small_lds(K kern, size_t bytes) {
    if (bytes > 160 * 1024) { mi_set_error("thin-channel kernel: LDS request exceeds 160 KiB"); return -1; }
    if (bytes > 48 * 1024) {
        hipError_t e = mi_func_set_lds(reinterpret_cast<const void*>(kern), (int)bytes);
        if (e != hipSuccess) { mi_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return -2; }
    }
    return 0;
}

template <class TR, int KZ>
int run_c1_fwd(const C1Params& p, hipStream_t st) {
    constexpr int KK = KZ * 9, ROWS = 4;
    // threads: N / 8 lanes per coarse column, 256 at most; a grid row with fewer columns than that (48^3 coarse grids
    // of 96^3 patches: 48 columns x 4 lanes) gets a smaller workgroup instead of a quarter of its lanes idling
    const int G = p.N / 8;
    int nthr = 256;
    if (p.cd[2] * G < 256 && (16 * G) % 64 == 0) { nthr = (p.cd[2] + 15) / 16 * 16 * G; if (nthr > 256) nthr = 256; }
    const int ppb = nthr / G;
    const size_t lds = (size_t)(p.N * KK + 2 * (ppb > 16 ? ppb : 16) * p.N) * sizeof(float);
    if (KZ == 1 && (p.N == 32 || p.N == 64) && (size_t)p.fd[1] * p.fd[2] < (1u << 30) && p.fd[2] >= 2 * p.cd[2] && p.fd[1] >= 1) {
        static const bool strip = getenv("MI_NO_C1_STRIP") == nullptr;
        constexpr int R2 = 4, G2 = 8;      // 32 coarse rows per workgroup
        const int cols = p.N == 32 ? 64 : 32;
        dim3 grid((p.cd[2] + cols - 1) / cols, (p.cd[1] + R2 * G2 - 1) / (R2 * G2), p.nb);
        // strips only where they fill the chip: 8 slices of 256 x 256 are 64 strips (0.775 ms per step with them, 0.765
        // with the one-row-group form and its 512 workgroups)
        if (strip && (long long)grid.x * grid.y * grid.z >= 256) {
            if (p.N == 32) hipLaunchKernelGGL((c1_s2_fwd32_kernel<TR, R2, G2, 32>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((c1_s2_fwd32_kernel<TR, R2, G2, 64>), grid, dim3(256), 0, st, p);
            return check_launch("c1_s2_fwd32");
        }
    }
    if (KZ == 3 && TR::ESZ == 2 && (p.N == 16 || p.N == 32) && (long long)p.fd[1] * p.fd[2] < (1ll << 30)) {
        static const bool mfma = getenv("MI_NO_C1_MFMA3D") == nullptr;
        if (mfma) {
            int tx16 = (p.cd[2] + 15) / 16;
            if (tx16 > 4) tx16 = 4;
            const size_t l2 = (size_t)(3 * (2 * C1M_TY + 1) * (32 * tx16 + 1) + 4 * p.N * 2) * sizeof(float);
            dim3 g2((p.cd[2] + 16 * tx16 - 1) / (16 * tx16), (p.cd[1] + C1M_TY - 1) / C1M_TY, p.nb * p.cd[0]);
#define C1M_CASE(NBV, TXV) if (p.N == 16 * NBV && tx16 == TXV) hipLaunchKernelGGL((c1_s2_fwd3d_mfma_kernel<NBV, TXV>), g2, dim3(256), l2, st, p);
            C1M_CASE(1, 1) C1M_CASE(1, 2) C1M_CASE(1, 3) C1M_CASE(1, 4) C1M_CASE(2, 1) C1M_CASE(2, 2) C1M_CASE(2, 3) C1M_CASE(2, 4)
#undef C1M_CASE
            return check_launch("c1_s2_fwd3d_mfma");
        }
    }
    if (int rc = small_lds(c1_s2_fwd_kernel<TR, KZ>, lds)) return rc;
    dim3 grid((p.cd[2] + ppb - 1) / ppb, (p.cd[1] + ROWS * C1F_YG - 1) / (ROWS * C1F_YG), p.nb * p.cd[0]);
    hipLaunchKernelGGL((c1_s2_fwd_kernel<TR, KZ>), grid, dim3(nthr), lds, st, p);
    return check_launch("c1_s2_fwd");
}

template <class TR, int KZ>
int run_c1_T(const C1Params& p, hipStream_t st) {
    constexpr int KK = KZ * 9;
    constexpr int PZ = KZ == 3 ? 4 : 1, PY = 8, PX = KZ == 3 ? 8 : 32;
    constexpr int OZ = KZ == 3 ? PZ - 1 : 1, OY = PY - 1, OX = PX - 1;
    // weights fp32 [KK][N], tap products [KK][256], and (bf16) the weight image for the matrix cores
    const size_t lds = (size_t)(KK * p.N + KK * 256) * sizeof(float) + (TR::ESZ == 2 ? (size_t)((KK + 15) / 16) * 16 * ((p.N + 31) & ~31) * 2 : 0);
    if (int rc = small_lds(c1_s2_T_kernel<TR, KZ>, lds)) return rc;
    const int nyp = (p.cd[1] + OY - 1) / OY;
    dim3 grid((p.cd[2] + OX - 1) / OX, (nyp + C1T_YP - 1) / C1T_YP, p.nb * ((p.cd[0] + OZ - 1) / OZ));
    hipLaunchKernelGGL((c1_s2_T_kernel<TR, KZ>), grid, dim3(256), lds, st, p);
    return check_launch("c1_s2_T");
}

}  // namespace

int launch_c1_s2_fwd(int dtype, const C1Params& p, hipStream_t st) {
    MI_CHECK(p.N % 8 == 0 && p.N >= 8 && p.N <= 256, "N must be a multiple of 8, at most 256");
    MI_CHECK(!p.y2_out || (p.n_o1 % 8 == 0 && p.n_o1 > 0), "second destination must start at a multiple of 8 channels");
    MI_CHECK((long long)p.nb * p.cd[0] <= 65535, "batch x depth exceeds the grid limit");
    if (p.ndim == 3) return dtype ? run_c1_fwd<BF16, 3>(p, st) : run_c1_fwd<F32, 3>(p, st);
    return dtype ? run_c1_fwd<BF16, 1>(p, st) : run_c1_fwd<F32, 1>(p, st);
}

int launch_c1_s2_T(int dtype, const C1Params& p, hipStream_t st) {
    MI_CHECK(p.N % 8 == 0 && p.N >= 8 && p.N <= 256, "N must be a multiple of 8, at most 256");
    MI_CHECK(!p.y2 || (p.n_y1 % 8 == 0 && p.n_y1 > 0), "second source must start at a multiple of 8 channels");
    MI_CHECK(p.fd[2] == 2 * p.cd[2] && p.fd[1] == 2 * p.cd[1] && (p.ndim == 2 || p.fd[0] == 2 * p.cd[0]),
             "fine dims must be twice the coarse dims");
    MI_CHECK((long long)p.nb * p.cd[0] <= 65535, "batch x depth exceeds the grid limit");
    if (p.ndim == 3) return dtype ? run_c1_T<BF16, 3>(p, st) : run_c1_T<F32, 3>(p, st);
    return dtype ? run_c1_T<BF16, 1>(p, st) : run_c1_T<F32, 1>(p, st);
}

// slabs (= workgroups along x) of the weight-gradient kernel: whole column blocks times row blocks
int c1_wgrad_slabs(int N, const int cd[3]) {
    const int ppb = 256 / (N / C1_WGRAD_CG);
    const int xblocks = (cd[2] + ppb - 1) / ppb;
    int rb = 768 / xblocks;         // three workgroups per CU fit the register file: all resident at once
    if (rb < 1) rb = 1;
    return xblocks * rb;
}

// whether launch_c1_s2_wgrad() takes the matrix-core kernel for these operands
bool c1_wgrad3d_mfma_applicable(int dtype, const C1Params& p) {
    static const bool mfma = getenv("MI_NO_C1_MFMA3D") == nullptr;
    static const bool mfma2d = getenv("MI_NO_C1_MFMA2D") == nullptr;
    return dtype && (p.ndim == 3 ? mfma && (p.N == 16 || p.N == 32) : mfma2d && (p.N == 16 || p.N == 32 || p.N == 64)) && p.n_y1 % 16 == 0 && p.y_pitch % 8 == 0 && p.y_coff % 8 == 0 &&
           (p.n_y1 >= p.N || (p.y2_pitch % 8 == 0 && p.y2_coff % 8 == 0));
}

// partial: [nsplit][KK][N] floats, partial_bias: [nsplit][N] floats
int launch_c1_s2_wgrad(int dtype, const C1Params& p, hipStream_t st) {
    MI_CHECK(p.N % 8 == 0 && p.N >= 8 && p.N <= 512 && p.N / C1_WGRAD_CG <= 256, "N must be a multiple of 8, at most 512");
    MI_CHECK(p.n_y1 % 8 == 0 && p.n_y1 > 0 && (p.n_y1 >= p.N || p.y2 != nullptr), "second source missing or misaligned");
    MI_CHECK(p.nsplit == c1_wgrad_slabs(p.N, p.cd), "nsplit must come from c1_wgrad_slabs()");
    if (c1_wgrad3d_mfma_applicable(dtype, p)) {
        const bool wide = p.cd[2] > 32;
#define C1WM_CASE(NBV, TXV, KZV) hipLaunchKernelGGL((c1_s2_wgrad3d_mfma_kernel<NBV, TXV, KZV>), dim3(p.nsplit), dim3(256), 0, st, p)
        if (p.ndim == 3) {
            if (p.N == 16) { if (wide) C1WM_CASE(1, 2, 3); else C1WM_CASE(1, 1, 3); }
            else { if (wide) C1WM_CASE(2, 2, 3); else C1WM_CASE(2, 1, 3); }
        } else {
            if (p.N == 16) { if (wide) C1WM_CASE(1, 2, 1); else C1WM_CASE(1, 1, 1); }
            else if (p.N == 32) { if (wide) C1WM_CASE(2, 2, 1); else C1WM_CASE(2, 1, 1); }
            else { if (wide) C1WM_CASE(4, 2, 1); else C1WM_CASE(4, 1, 1); }
        }
#undef C1WM_CASE
        return check_launch("c1_s2_wgrad3d_mfma");
    }
    const bool quad = p.N / C1_WGRAD_CG == 4;
    const size_t lds = (size_t)(quad ? 16 : 4) * p.N * 10 * sizeof(float);   // one image per wave, per 16-lane row in the quad variant
    dim3 grid(p.nsplit, p.ndim == 3 ? 3 : 1);
#define C1W_CASE(TR, Q)                                                                   \
    do {                                                                                  \
        if (int rc = small_lds(c1_s2_wgrad_kernel<TR, Q>, lds)) return rc;                \
        hipLaunchKernelGGL((c1_s2_wgrad_kernel<TR, Q>), grid, dim3(256), lds, st, p);     \
    } while (0)
    if (dtype) { if (quad) C1W_CASE(BF16, true); else C1W_CASE(BF16, false); }
    else { if (quad) C1W_CASE(F32, true); else C1W_CASE(F32, false); }
#undef C1W_CASE
    return check_launch("c1_s2_wgrad");
}

// the 2-D strip kernel serves p: whole 16-byte groups per row and at least four strips of rows
bool stencil_norm_on_load_ok(const StencilParams& p) {
    return p.ndim == 2 && p.d[0] == 1 && p.d[2] % 4 == 0 && p.d[1] >= 4 * ST_R && (long long)p.d[1] * p.d[2] < (1ll << 31);
}

int launch_stencil(const StencilParams& p, hipStream_t st) {
    const long long npix = (long long)p.d[0] * p.d[1] * p.d[2];
    MI_CHECK(!p.nrm_stats || stencil_norm_on_load_ok(p), "this stencil cannot normalise its input on load");
    if (p.d[2] % 4 == 0) {
        const unsigned gx = (unsigned)((npix / 4 + 255) / 256);
        static const bool tiled = getenv("MI_NO_STENCIL3D_TILE") == nullptr;
        const long long tiles_xy = (long long)((p.d[2] + S3_TX - 1) / S3_TX) * ((p.d[1] + S3_TY - 1) / S3_TY);
        if (p.ndim == 3 && tiled && tiles_xy < (1ll << 31) && (p.d[0] + S3_TZ - 1) / S3_TZ <= 65535 && p.nb <= 65535)
            hipLaunchKernelGGL(stencil3d_tile_kernel, dim3((unsigned)tiles_xy, (p.d[0] + S3_TZ - 1) / S3_TZ, p.nb), dim3(256), 0, st, p);
        else if (p.ndim == 3) hipLaunchKernelGGL(stencil_kernel<3>, dim3(gx, p.nb), dim3(256), 0, st, p);
        else if (p.d[1] >= 4 * ST_R) {
            const int threads = (p.d[2] >> 2) * ((p.d[1] + ST_R - 1) / ST_R);
            if (p.nrm_stats) hipLaunchKernelGGL(stencil2d_strip_kernel<true>, dim3((threads + 255) / 256, p.nb), dim3(256), 0, st, p);
            else hipLaunchKernelGGL(stencil2d_strip_kernel<false>, dim3((threads + 255) / 256, p.nb), dim3(256), 0, st, p);
        } else hipLaunchKernelGGL(stencil_kernel<1>, dim3(gx, p.nb), dim3(256), 0, st, p);
    } else {
        hipLaunchKernelGGL(stencil_scalar_kernel, dim3((unsigned)((npix + 255) / 256), p.nb), dim3(256), 0, st, p);
    }
    return check_launch("stencil");
}

int launch_stencil_wgrad(const StencilParams& p, hipStream_t st) {
    MI_CHECK((long long)p.nb * p.d[0] * p.d[1] * p.d[2] < (1ll << 31), "pixel count must stay below 2^31");
    const dim3 grid(p.nsplit);
    const bool quad = p.d[2] % 4 == 0;
    // index arithmetic: 32-bit in 2-D (25.9 -> 24.1 us at 32 x 512^2); in 3-D the 64-bit form is the faster one
    // (4 x 96^3: 29.5 us against 41-52 us: its long divisions stagger the waves' 28 loads, measured, not understood)
    if (p.ndim == 3) {
        if (quad) hipLaunchKernelGGL((stencil_wgrad_kernel<4, 3, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((stencil_wgrad_kernel<1, 3, true>), grid, dim3(256), 0, st, p);
    } else {
        if (quad) hipLaunchKernelGGL((stencil_wgrad_kernel<4, 1, false>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((stencil_wgrad_kernel<1, 1, false>), grid, dim3(256), 0, st, p);
    }
    return check_launch("stencil_wgrad");
}
